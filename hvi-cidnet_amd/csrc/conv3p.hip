// Dense 3x3 convolution (zero padding) of the bf16 MODE, without LDS: the nine taps as nine shifted 1x1 products on the BF16
// matrix cores, every wave independent (the structure of pwx.hip, third version).
//
//   Y[b][m][p] = sum over k = (ci, tap) of A[m][k] * X[b][ci][p + shift(tap)]   (+ R[b][m][p]),   zero outside the image
//
// conv3x.hip stages a (8+2) x (32+2) pixel tile through LDS to turn NCHW planes into channel-innermost B fragments; with
// one-level operands (both rounded to nearest bf16, ONE product per term -- ops.set_precision("bf16")) its matrix-core work is
// 40 us of a 238 us launch at 8 x 36 -> 36 x 400 x 600 and the load -> convert -> LDS -> barrier structure is what remains
// (DESIGN.md section 4.2 (d)).  Here, as in pwx.hip, column n of N-tile e is pixel 4 n + e of the wave's 64-pixel group of the
// FLATTENED plane, so one float4 load of channel ci at pixels 4n + shift .. 4n + 3 + shift supplies element k = (tap, ci) of
// all four N-tiles' B fragments: a lane loads the eight k of its lane group (eight float4, each with its own tap shift;
// unaligned 16-byte loads), converts pairs (one v_cvt_pk_bf16_f32 per two values) and multiplies from registers.  An input
// element is then fetched nine times -- from L1 / L2, not from HBM: 90 KB of vector-L1 traffic per 64 pixels at K = 324, 2.7 us
// per 256 pixels and CU against 4.9 us of HBM time -- and no barrier, no LDS capacity limit and no tile halo exist.
// Image borders: a lane knows, per pixel of its quad, which of the three row shifts and which of the three column shifts stay
// inside the image (two 12-bit masks); loads come from clamped (always valid) addresses and masked elements are zeroed.  A
// wave whose 64 pixels and their neighbours are all interior skips the masking (wave-uniform branch).
// Only one operand level exists here: with three (the parity mode) the split would be redone for each of the nine taps.
// No packed-fp32 / SDWA instructions (hvi-cidnet_amd/build.py).
#include "common.h"
#include "cidnet_hip.h"
#include <type_traits>

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kMaxMTW = 5;

struct P3Args {
  const float* X; long x_bs;
  const uint4* Af;                     // weights rounded to bf16 in fragment order: [kb][mt][64 lanes]
  float* Y; long y_bs;
  const float* R; long r_bs;
  int B, M, K9;                        // K9 = 9 * input channels
  int H, W; long HW;
  int KB, MT, tiles_per_sample;
};

__device__ __forceinline__ unsigned cvt_pair(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// ---- weights -> bf16 fragments: lane (r = lane & 15, g = lane >> 4) of fragment (kb, mt) holds A[16 mt + r][32 kb + 8 g .. + 7],
// k = 9 ci + tap (tap = 3 dy + dx of the FORWARD orientation; `flip` selects the data-gradient's rotated taps) ----
__device__ __forceinline__ void conv3p_prep_item(const float* __restrict__ Wt, long w_ms, long w_ks, int flip, int M, int cin,
                                                 uint4* __restrict__ Af, int KB, int MT, long idx) {
  const int lane = (int)(idx & 63);
  const long t = idx >> 6;
  if (t >= (long)KB * MT) return;
  const int mt = (int)(t % MT), kb = (int)(t / MT);
  const int m = mt * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j, ci = k / 9, tap = k - 9 * ci;
    v[j] = (m < M && k < 9 * cin) ? Wt[(long)m * w_ms + (long)ci * w_ks + (flip ? 8 - tap : tap)] : 0.f;
  }
  Af[t * 64 + lane] = uint4{cvt_pair(v[0], v[1]), cvt_pair(v[2], v[3]), cvt_pair(v[4], v[5]), cvt_pair(v[6], v[7])};
}

__global__ __launch_bounds__(kThreads) void conv3p_prep_kernel(const float* __restrict__ Wt, long w_ms, long w_ks, int flip, int M, int cin,
                                                               uint4* __restrict__ Af, int KB, int MT) {
  conv3p_prep_item(Wt, w_ms, w_ks, flip, M, cin, Af, KB, MT, (long)blockIdx.x * kThreads + threadIdx.x);
}

// many layers in ONE launch (see pwx_split_w_batch_kernel): row = {source, destination, M, K, w_ms, w_ks, first block, flip}
__global__ __launch_bounds__(kThreads) void conv3p_prep_batch_kernel(const long long* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * 8 + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* r = table + (long)lo * 8;
  const int M = (int)r[2], K = (int)r[3];
  conv3p_prep_item(reinterpret_cast<const float*>(r[0]), r[4], r[5], (int)r[7], M, K, reinterpret_cast<uint4*>(r[1]), (9 * K + 31) / 32,
                   (M + 15) / 16, ((long)blockIdx.x - r[6]) * kThreads + threadIdx.x);
}

template <int MTW, int WM, int CIN>
__global__ __launch_bounds__(kThreads, 2) void conv3p_kernel(P3Args a) {
  constexpr int NG = 4 / WM;                                     // pixel groups per block
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int b = blockIdx.x / a.tiles_per_sample, tile = blockIdx.x - b * a.tiles_per_sample;
  const long HW = a.HW;
  const int W = a.W, H = a.H;
  const long p0o = ((long)tile * NG + wn) * 64;                  // first pixel this wave owns
  const int mt0 = (blockIdx.y * WM + wm) * MTW;
  if (p0o >= HW || mt0 >= a.MT) return;                          // wave-uniform; no barrier anywhere in this kernel
  const long p0 = p0o + 64 <= HW ? p0o : HW - 64;                // ragged last group: pulled back (HW >= 64)
  const int pq = (int)(p0 + 4 * n);                              // this lane's pixel quad (flattened index)
  const bool stores = pq + 3 >= p0o;
  const bool whole = pq >= p0o;

  // which row shifts (dy - 1) and column shifts (dx - 1) keep pixel e of the quad inside the image: bit 4 d + e
  unsigned rowm = 0u, colm = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int y = (pq + e) / W, x = (pq + e) - y * W;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (y + d - 1 >= 0 && y + d - 1 < H) rowm |= 1u << (4 * d + e);
      if (x + d - 1 >= 0 && x + d - 1 < W) colm |= 1u << (4 * d + e);
    }
  }
  const bool interior = __builtin_amdgcn_ballot_w64(rowm != 0xFFFu || colm != 0xFFFu) == 0ull;   // wave-uniform

  f32x4 acc[MTW][4];
  if (a.R) {                                                      // the addend is the accumulators' start value
    const float* Rb = a.R + (long)b * a.r_bs + pq;
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min((mt0 + j) * 16 + 4 * g + r, a.M - 1);
        const f32x4 t = load4u(Rb + (long)m * HW);
        acc[j][0][r] = t[0]; acc[j][1][r] = t[1]; acc[j][2][r] = t[2]; acc[j][3][r] = t[3];
      }
  } else {
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int mts[MTW];
#pragma unroll
  for (int j = 0; j < MTW; ++j) mts[j] = min(mt0 + j, a.MT - 1);

  const float* Xb = a.X + (long)b * a.x_bs;
  const int K9 = a.K9;
  const int last = (int)HW - 4;                                  // CIN * HW < 2^31 (supported()): 32-bit element offsets
  uint4 A[MTW];
  auto load_a = [&](int kb) {
    const uint4* Ak = a.Af + (long)kb * a.MT * 64;
#pragma unroll
    for (int j = 0; j < MTW; ++j) A[j] = Ak[(long)mts[j] * 64 + lane];
  };
  // eight k of this lane group: k = 32 kb + 8 g + c -> (tap, ci); the quad at the tap's shift.  INTERIOR waves (every pixel of the
  // wave has all nine neighbours inside the image: wave-uniform) load and convert without any check.  Elsewhere a shifted quad
  // may start before the first or end after the last element of the plane (first / last image row only): those quads are
  // read element by element from clamped indices, all others as one 16-byte load; elements outside the image are zeroed.
  uint4 bf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bf[e] = uint4{0u, 0u, 0u, 0u};
  auto burst = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16x8 bl = __builtin_bit_cast(bf16x8, bf[e]);
#pragma unroll
      for (int j = 0; j < MTW; ++j) acc[j][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[j]), bl, acc[j][e], 0, 0, 0);
    }
  };
  auto run = [&](auto interior_tag) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
    auto load_raw = [&](f32x4 (&raw)[8], unsigned (&msk)[8], int kb) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int k = min(kb * 32 + g * 8 + c, K9 - 1);          // slots past K: finite data times a zero weight
        const int ci = k / 9, tap = k - 9 * ci;                  // channel-major k: the nine taps of a channel are neighbours
        const int dy = tap / 3, dx = tap - 3 * dy;               // in k, so its three rows are fetched from L2 once and re-read from L1
        const int o = pq + (dy - 1) * W + (dx - 1);
        const float* plane = Xb + (unsigned)ci * (unsigned)HW;
        if constexpr (INTERIOR) {
          raw[c] = load4u(plane + o);
        } else {
          const bool inr = o >= 0 && o <= last;
          raw[c] = load4u(plane + (inr ? o : 0));
          if (!inr) {                                            // first / last image row: rare, divergent
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[c][e] = plane[min(max(o + e, 0), last + 3)];
          }
          msk[c] = (rowm >> (4 * dy)) & (colm >> (4 * dx)) & 15u;
        }
      }
    };
    auto convert = [&](f32x4 (&raw)[8], const unsigned (&msk)[8]) {
      if constexpr (!INTERIOR) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) raw[c][e] = ((msk[c] >> e) & 1u) ? raw[c][e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf[e].x = cvt_pair(raw[0][e], raw[1][e]);
        bf[e].y = cvt_pair(raw[2][e], raw[3][e]);
        bf[e].z = cvt_pair(raw[4][e], raw[5][e]);
        bf[e].w = cvt_pair(raw[6][e], raw[7][e]);
      }
    };
    // one k-block: convert what was loaded, re-request into the same registers, MFMA burst, request the next weight
    // fragments (the scheduling fences keep the phases apart, as in pwx.hip)
    f32x4 r0[8];
    unsigned m0[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    load_raw(r0, m0, 0);
    load_a(0);
    const int KB = a.KB;
    for (int kb = 0; kb < KB; ++kb) {
      convert(r0, m0);
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < KB) load_raw(r0, m0, kb + 1);
      __builtin_amdgcn_sched_barrier(0);
      burst();
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < KB) load_a(kb + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (interior) run(std::true_type{});
  else run(std::false_type{});

  // ---- epilogue ----
  if (!stores) return;
#pragma unroll
  for (int j = 0; j < MTW; ++j) {
    const int mt = mt0 + j;
    if (mt >= a.MT) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 16 + 4 * g + r;
      if (m >= a.M) continue;
      float* yp = a.Y + (long)b * a.y_bs + (long)m * HW + pq;
      const float v0 = acc[j][0][r], v1 = acc[j][1][r], v2 = acc[j][2][r], v3 = acc[j][3][r];
      if (whole) {
        store4u(yp, f32x4{v0, v1, v2, v3});
      } else {
        const float vv[4] = {v0, v1, v2, v3};
        for (int e = 0; e < 4; ++e)
          if (pq + e >= p0o) yp[e] = vv[e];
      }
    }
  }
}

struct P3Plan {
  int KB, MT, WM, MTW, chunks, tiles_per_sample;
};

inline P3Plan p3_plan(int M, int cin, long HW) {
  P3Plan p;
  p.KB = (9 * cin + 31) / 32;
  p.MT = (M + 15) / 16;
  p.WM = p.MT <= kMaxMTW ? 1 : (p.MT <= 2 * kMaxMTW ? 2 : 4);
  p.chunks = (p.MT + p.WM * kMaxMTW - 1) / (p.WM * kMaxMTW);
  p.MTW = (p.MT + p.WM * p.chunks - 1) / (p.WM * p.chunks);
  const int block_px = (4 / p.WM) * 64;
  p.tiles_per_sample = (int)((HW + block_px - 1) / block_px);
  return p;
}

template <int MTW, int CIN>
void launch_p3(const P3Args& a, const P3Plan& p, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * p.tiles_per_sample), (unsigned)p.chunks);
  if (p.WM == 1) hipLaunchKernelGGL((conv3p_kernel<MTW, 1, CIN>), grid, dim3(kThreads), 0, s, a);
  else if (p.WM == 2) hipLaunchKernelGGL((conv3p_kernel<MTW, 2, CIN>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((conv3p_kernel<MTW, 4, CIN>), grid, dim3(kThreads), 0, s, a);
}

template <int CIN>
void launch_p3_cin(const P3Args& a, const P3Plan& p, hipStream_t s) {
  switch (p.MTW) {
    case 1: launch_p3<1, CIN>(a, p, s); break;
    case 2: launch_p3<2, CIN>(a, p, s); break;
    case 3: launch_p3<3, CIN>(a, p, s); break;
    case 4: launch_p3<4, CIN>(a, p, s); break;
    default: launch_p3<5, CIN>(a, p, s); break;
  }
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* CIDNet's dense layers: 36 / 72 / 144 input channels; planes of at least 64 pixels, 32-bit element offsets */
int cidnet_conv3x3_bf16_direct_supported(int M, int K, int H, int W) {
  return M >= 1 && (K == 36 || K == 72 || K == 144) && H >= 1 && W >= 4 && (long)H * W >= 64 && (long)K * H * W < (1L << 31) &&
                 (long)M * H * W < (1L << 31)
             ? 1 : 0;
}

long cidnet_conv3x3_bf16_direct_ws_floats(int M, int K) { return (long)((9 * K + 31) / 32) * ((M + 15) / 16) * 64 * 4; }

int cidnet_conv3x3_bf16_direct_prep(const float* Wt, long w_ms, long w_ks, int flip, float* ws, long ws_floats, int M, int K, void* stream) {
  CIDNET_CHECK_ARG(Wt && ws && M > 0 && K > 0);
  if (ws_floats < cidnet_conv3x3_bf16_direct_ws_floats(M, K)) return CIDNET_ERR_WS;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  const int KB = (9 * K + 31) / 32, MT = (M + 15) / 16;
  const long threads = (long)KB * MT * 64;
  hipLaunchKernelGGL(conv3p_prep_kernel, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, Wt,
                     w_ms, w_ks, flip, M, K, reinterpret_cast<uint4*>(ws), KB, MT);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_conv3x3_bf16_direct_prep_blocks(int M, int K) {
  return ((long)((9 * K + 31) / 32) * ((M + 15) / 16) * 64 + kThreads - 1) / kThreads;
}

int cidnet_conv3x3_bf16_direct_prep_batch(const long long* table, int n, long total_blocks, void* stream) {
  CIDNET_CHECK_ARG(table && n > 0 && total_blocks > 0);
  hipLaunchKernelGGL(conv3p_prep_batch_kernel, dim3((unsigned)total_blocks), dim3(kThreads), 0, (hipStream_t)stream, table, n);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_conv3x3_bf16_direct_pre(const float* X, long x_bs, const float* Wprep, const float* R, long r_bs, float* Y, long y_bs, int B,
                                   int M, int K, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(X && Wprep && Y && B > 0);
  if (!cidnet_conv3x3_bf16_direct_supported(M, K, H, W)) return CIDNET_ERR_SHAPE;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(Wprep) & 15) == 0);
  const long HW = (long)H * W;
  const P3Plan p = p3_plan(M, K, HW);
  P3Args a{X, x_bs, reinterpret_cast<const uint4*>(Wprep), Y, y_bs, R, r_bs, B, M, 9 * K, H, W, HW, p.KB, p.MT, p.tiles_per_sample};
  hipStream_t s = (hipStream_t)stream;
  if (K == 36) launch_p3_cin<36>(a, p, s);
  else if (K == 72) launch_p3_cin<72>(a, p, s);
  else launch_p3_cin<144>(a, p, s);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
