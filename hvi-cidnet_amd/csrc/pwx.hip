// Pointwise (1x1) convolution with fp32 operands on the BF16 matrix cores ("bf16x3" split products, see conv3x.hip for the
// arithmetic): for the 1x1 convs of the step that the fp32 MFMA rate bounds -- the coarse levels (100x150 and 50x75 planes
// with 72 .. 766 channels: 45 - 75 TFLOP/s on v_mfma_f32_16x16x4_f32, 1 - 2.5 TB/s).  Same contract as cidnet_pw_conv
// (pw.hip; net/LCA.py:13,15,17,51,57, net/transformer_utils.py:60):
//   Y[b] (M x HW) = A_b (M x K) * X[b] (K x HW) [+ R[b]],   A_b[m][k] = Wt[b*w_bs + m*w_ms + k*w_ks].
// Every global access moves 16 bytes:
//
//  * MFMA 16x16x32: lane (n = lane & 15, g = lane >> 4) holds A[m = n][8g .. 8g+7] and B[8g .. 8g+7][column n].
//    Column n of N-tile e (e = 0..3) is pixel 4 n + e of the wave's 64-pixel group, so ONE float4 load of channel k at
//    pixels 4n .. 4n+3 supplies element k of all four N-tiles' B fragments, and register r of the four accumulators of a
//    channel tile is the float4  Y[m = 4 g + r][4n .. 4n+3]  -- loads, residual loads and stores are 16 B per lane, 256 B
//    contiguous per channel row.
//  * A lane that loads the eight channels 8g .. 8g+7 of its quad therefore owns the complete B fragments of its four
//    N-tiles: a wave splits what it loaded (round to nearest, exact: two channels per v_cvt_pk) and multiplies from
//    registers -- no LDS, no barrier (third version, round 4; the history is at the kernel).  (M <= 16 -- a single
//    channel tile -- is left to pw.hip.)
//  * Weights: a small kernel splits them once per call into fragment order in a workspace (L2 resident); a wave requests
//    the fragments of its <= 5 channel tiles for the next k-block right after the MFMA burst that read the current ones.
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

struct PwxArgs {
  const void* X; long x_bs;            // fp32, or bf16 (template parameter XB) in the bf16 mode; strides in elements
  const uint4* Af; long a_bs;          // split weights in fragment order; a_bs = fragments-per-sample * 64 (0: shared)
  void* Y; long y_bs;                  // fp32 or bf16 (YB)
  const float* R; long r_bs;           // the residual stream stays fp32
  int B, M, K; long HW;
  int KB, MT;                          // k-blocks of 32, 16-row tiles of M
  int tiles_per_sample;                // block pixel tiles (4 / WM * 64 pixels) per sample
};

// exact three-way split of two fp32 values into packed bf16 pairs (lo half = a, hi half = b), round to nearest even
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const bf16x2 h0 = __builtin_convertvector(f32x2{a, b}, bf16x2);
  p0 = __builtin_bit_cast(unsigned, h0);
  const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xFFFF0000u);
  const bf16x2 h1 = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
  p1 = __builtin_bit_cast(unsigned, h1);
  const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xFFFF0000u);
  const bf16x2 h2 = __builtin_convertvector(f32x2{sa, sb}, bf16x2);
  p2 = __builtin_bit_cast(unsigned, h2);
}

// ---- weights -> three bf16 levels in fragment order: fragment (b, kb, mt, level) = 64 lanes x uint4 ----
__device__ __forceinline__ void pwx_split_w_item(const float* __restrict__ Wt, long w_bs, long w_ms, long w_ks, uint4* __restrict__ Af,
                                                 int M, int K, int KB, int MT, int nb, int cpg, long idx) {
  const int lane = (int)(idx & 63);
  const long t = idx >> 6;
  if (t >= (long)nb * KB * MT) return;
  const int mt = (int)(t % MT), kb = (int)((t / MT) % KB), b = (int)(t / ((long)MT * KB));
  // k-slots 8g .. 8g + cpg - 1 of a k-block carry channels, the rest of the lane group's eight slots are zero (cpg = 6:
  // 24-channel k-blocks for K = 36 / 72 / 144, see pwx_cpg)
  const int m = mt * 16 + (lane & 15), k0 = kb * (4 * cpg) + (lane >> 4) * cpg;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (m < M && j < cpg && k0 + j < K) ? Wt[(long)b * w_bs + (long)m * w_ms + (long)(k0 + j) * w_ks] : 0.f;
  uint4 o[3];
  split3_pair(v[0], v[1], o[0].x, o[1].x, o[2].x);
  split3_pair(v[2], v[3], o[0].y, o[1].y, o[2].y);
  split3_pair(v[4], v[5], o[0].z, o[1].z, o[2].z);
  split3_pair(v[6], v[7], o[0].w, o[1].w, o[2].w);
#pragma unroll
  for (int l = 0; l < 3; ++l) Af[(t * 3 + l) * 64 + lane] = o[l];
}

__global__ __launch_bounds__(kThreads) void pwx_split_w_kernel(const float* __restrict__ Wt, long w_bs, long w_ms, long w_ks,
                                                               uint4* __restrict__ Af, int M, int K, int KB, int MT, int nb, int cpg) {
  pwx_split_w_item(Wt, w_bs, w_ms, w_ks, Af, M, K, KB, MT, nb, cpg, (long)blockIdx.x * kThreads + threadIdx.x);
}

// Channels per lane group and k-block: 8 (32-channel k-blocks) unless 6 (24-channel k-blocks, two of a lane's eight k-slots
// zero) needs the same number of k-blocks -- K = 36, 72: 48 / 72 instead of 64 / 96 loaded and split channels, same MFMA
// count (-9 % at 72 x 72 on 200x300 planes).  K = 144 would take six k-blocks instead of five: slower where M is large.
#ifndef PWX_CPG6
#define PWX_CPG6 1
#endif
inline int pwx_cpg(int K) {
  return (PWX_CPG6 && (K + 23) / 24 == (K + 31) / 32) ? 6 : 8;
}

// Many weight tensors in ONE launch (the trainer refreshes every prepared operand of the step after the optimizer's update
// instead of ~100 five-microsecond launches spread over the step).  Row r of the table (8 x int64): source pointer,
// destination pointer, M, K, w_ms, w_ks, first block of the row, unused; a block finds its row by bisection.
constexpr int kPrepRow = 8;
__global__ __launch_bounds__(kThreads) void pwx_split_w_batch_kernel(const long long* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {                                              // last row whose first block is <= blockIdx.x (uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * kPrepRow + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* r = table + (long)lo * kPrepRow;
  const int M = (int)r[2], K = (int)r[3];
  const int cpg = (PWX_CPG6 && (K + 23) / 24 == (K + 31) / 32) ? 6 : 8;
  const int KB = (K + 4 * cpg - 1) / (4 * cpg), MT = (M + 15) / 16;
  pwx_split_w_item(reinterpret_cast<const float*>(r[0]), 0, r[4], r[5], reinterpret_cast<uint4*>(r[1]), M, K, KB, MT, 1, cpg,
                   ((long)blockIdx.x - r[6]) * kThreads + threadIdx.x);
}


#ifndef PWX_RINIT
#define PWX_RINIT 1          // accumulators start as the residual (its loads are the first in flight; no epilogue loads)
#endif
#ifndef PWX_DEEP_MAX_MTW
#define PWX_DEEP_MAX_MTW 0   // waves with at most this many tiles keep TWO k-blocks of activations in flight
#endif
#ifndef PWX_MAX_MTW
#define PWX_MAX_MTW 5
#endif

// WL / XL: bf16 levels of the weights / activations that enter the products (all pairs i + j <= 2, small terms first):
// (3, 3) the fp32-exact six products (parity mode); (1, 1) both operands rounded to nearest bf16, one product (the
// arithmetic of a bf16 autocast conv, fp32 accumulation and output), the activations converted instead of split;
// (3, 1) exact weights, rounded activations.
// XB / YB: the activations / the output are STORED as bf16 (bf16 mode only: XB needs XL == 1 -- a bf16 value has one level).
// A lane then loads 8 bytes per channel (four pixels) and packs the channel pairs of a pixel with one v_perm each; the
// output is rounded to nearest bf16 on store (8 bytes per row and lane).
template <int MTW, int WM, int CPG, int WL, int XL, bool XB = false, bool YB = false>
__global__ __launch_bounds__(kThreads, 2) void pwx_kernel(PwxArgs a) {
  static_assert(!XB || XL == 1, "bf16 activations have one level");
  using Raw = std::conditional_t<XB, uint2, f32x4>;              // four pixels of one channel as loaded
  constexpr int NG = 4 / WM;                                     // pixel groups per block
  constexpr int NP = CPG / 2;                                    // channel pairs = non-zero dwords of a B fragment
  constexpr bool DEEP = MTW <= PWX_DEEP_MAX_MTW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int b = blockIdx.x / a.tiles_per_sample, tile = blockIdx.x - b * a.tiles_per_sample;
  const long HW = a.HW;
  const long p0o = ((long)tile * NG + wn) * 64;                  // first pixel this wave owns
  const int mt0 = (blockIdx.y * WM + wm) * MTW;                  // this wave's tiles: mt0 .. mt0 + MTW - 1
  if (p0o >= HW || mt0 >= a.MT) return;                          // wave-uniform; no barrier anywhere in this kernel
  // the last, ragged group of a plane is pulled back to the plane's last 64 pixels (HW >= 64): every load is a whole
  // quad inside the plane, the pixels before p0o are computed twice and stored only by the group that owns them
  const long p0 = p0o + 64 <= HW ? p0o : HW - 64;
  const long pq = p0 + 4 * n;                                    // this lane's pixel quad
  const int K = a.K;
  const float* Xb = reinterpret_cast<const float*>(a.X) + (XB ? 0 : (long)b * a.x_bs);
  const bf16_t* Xh = reinterpret_cast<const bf16_t*>(a.X) + (XB ? (long)b * a.x_bs : 0);
  const uint4* Au = a.Af + (long)b * a.a_bs;                     // wave-uniform base; lanes differ by `lane` only
  const bool stores = pq + 3 >= p0o;                             // else: quad owned by the previous group
  const bool whole = pq >= p0o;                                  // else: a quad that straddles p0o (HW % 4 != 0 only)

  // register r of the four accumulators of tile j = Y[16 mt + 4 g + r][pq .. pq + 3]
  f32x4 acc[MTW][4];
  if (PWX_RINIT && a.R) {
    // the residual is the accumulators' start value: its loads are the first in flight and the epilogue only stores.
    // Unconditional loads from clamped rows (a row past M and a quad of the previous group are never stored)
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
  int mts[MTW];                                                  // slots past the last tile repeat it (loads stay in range)
#pragma unroll
  for (int j = 0; j < MTW; ++j) mts[j] = min(mt0 + j, a.MT - 1);

  const unsigned uHW = (unsigned)HW, upq = (unsigned)pq;         // K * HW < 2^31 (supported()): 32-bit element offsets
  uint4 A[MTW][WL];
  auto load_a = [&](int kb) {
    const uint4* Ak = Au + (long)kb * a.MT * (3 * 64);
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const uint4* Am = Ak + (long)mts[j] * (3 * 64);
#pragma unroll
      for (int l = 0; l < WL; ++l) A[j][l] = Am[l * 64 + lane];
    }
  };
  auto load_raw = [&](Raw (&raw)[CPG], int kb) {
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      const unsigned k = (unsigned)min(kb * (4 * CPG) + g * CPG + c, K - 1);   // rows past K: finite data times a zero weight
      if constexpr (XB) {
        const h4u v = *reinterpret_cast<const h4u*>(Xh + (k * uHW + upq));
        raw[c] = uint2{(unsigned)v.x | ((unsigned)v.y << 16), (unsigned)v.z | ((unsigned)v.w << 16)};
      } else {
        raw[c] = load4u(Xb + (k * uHW + upq));
      }
    }
  };
  // B fragments of the four N-tiles, three levels each: dword q of a fragment = channel pair q of the lane group
  uint4 bf[4][XL];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int l = 0; l < XL; ++l) bf[e][l] = uint4{0u, 0u, 0u, 0u};
  auto cvt_pair = [](float p, float q) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{p, q}, bf16x2)); };
  auto split_raw = [&](Raw (&raw)[CPG]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (XB) {
        // pixel e of channels (c, c + 1): halves e & 1 of dword e >> 1 of the two channels' quads, one v_perm_b32
        auto pk = [&](const uint2& lo, const uint2& hi) {
          const unsigned l = e < 2 ? lo.x : lo.y, h = e < 2 ? hi.x : hi.y;
          return __builtin_amdgcn_perm(h, l, (e & 1) ? 0x07060302u : 0x05040100u);
        };
        bf[e][0].x = pk(raw[0], raw[1]);
        bf[e][0].y = pk(raw[2], raw[3]);
        if constexpr (NP > 2) bf[e][0].z = pk(raw[4], raw[5]);
        if constexpr (NP > 3) bf[e][0].w = pk(raw[6], raw[7]);
      } else if constexpr (XL == 1) {
        bf[e][0].x = cvt_pair(raw[0][e], raw[1][e]);
        bf[e][0].y = cvt_pair(raw[2][e], raw[3][e]);
        if constexpr (NP > 2) bf[e][0].z = cvt_pair(raw[4][e], raw[5][e]);
        if constexpr (NP > 3) bf[e][0].w = cvt_pair(raw[6][e], raw[7][e]);
      } else {
        split3_pair(raw[0][e], raw[1][e], bf[e][0].x, bf[e][1].x, bf[e][2].x);
        split3_pair(raw[2][e], raw[3][e], bf[e][0].y, bf[e][1].y, bf[e][2].y);
        if constexpr (NP > 2) split3_pair(raw[4][e], raw[5][e], bf[e][0].z, bf[e][1].z, bf[e][2].z);
        if constexpr (NP > 3) split3_pair(raw[6][e], raw[7][e], bf[e][0].w, bf[e][1].w, bf[e][2].w);
      }
    }
  };
  auto burst = [&]() {
    bf16x8 al[MTW][WL];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
      for (int l = 0; l < WL; ++l) al[j][l] = __builtin_bit_cast(bf16x8, A[j][l]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16x8 bl[XL];
#pragma unroll
      for (int l = 0; l < XL; ++l) bl[l] = __builtin_bit_cast(bf16x8, bf[e][l]);
      // small terms first: level sums 2, 1, 0 (a2 b0, a1 b1, a0 b2, a1 b0, a0 b1, a0 b0 with three levels each)
#pragma unroll
      for (int sum = 2; sum >= 0; --sum)
#pragma unroll
        for (int i = sum; i >= 0; --i) {
          if (i >= WL || sum - i >= XL) continue;
#pragma unroll
          for (int j = 0; j < MTW; ++j) acc[j][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[j][i], bl[sum - i], acc[j][e], 0, 0, 0);
        }
    }
  };
  // One k-block: split the activations in `raw`, re-request into the same registers (k-block `kn`, if any), MFMA burst,
  // request the next k-block's weight fragments.  The scheduling fences keep the phases apart: the activation registers
  // are dead (split) before new loads are requested into them and the weight registers are re-requested only after the
  // burst has read them -- without them the scheduler hoists both requests and the five-tile wave spills.  LOAD / LAST are
  // compile-time so that every path is straight-line code with exact counted waits.
  auto step = [&](Raw (&raw)[CPG], int kb, int kn, auto load_tag, auto last_tag) {
    split_raw(raw);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(load_tag)::value) load_raw(raw, kn);
    __builtin_amdgcn_sched_barrier(0);
    burst();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!decltype(last_tag)::value) load_a(kb + 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  const int KB = a.KB;
  if constexpr (DEEP) {
    // two k-blocks of activations in flight (16 KB per wave): X(kb+2) is requested when X(kb) has been split; the burst
    // waits for W(kb), which is younger than X(kb+1) only -- requested a whole step earlier
    Raw r0[CPG], r1[CPG];
    load_raw(r0, 0);
    if (KB > 1) load_raw(r1, 1);
    load_a(0);
    int kb = 0;
    for (; kb + 3 < KB; kb += 2) {
      step(r0, kb, kb + 2, yes, no);
      step(r1, kb + 1, kb + 3, yes, no);
    }
    const int rem = KB - kb;                                     // 1, 2 or 3 k-blocks left, r0 holds the current one
    if (rem == 3) {
      step(r0, kb, kb + 2, yes, no);
      step(r1, kb + 1, 0, no, no);
      step(r0, kb + 2, 0, no, yes);
    } else if (rem == 2) {
      step(r0, kb, 0, no, no);
      step(r1, kb + 1, 0, no, yes);
    } else {
      step(r0, kb, 0, no, yes);
    }
  } else {
    Raw r0[CPG];
    load_raw(r0, 0);
    load_a(0);
    int kb = 0;
    for (; kb + 1 < KB; ++kb) step(r0, kb, kb + 1, yes, no);
    step(r0, kb, 0, no, yes);
  }
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
      const long o = (long)m * HW + pq;
      float v0 = acc[j][0][r], v1 = acc[j][1][r], v2 = acc[j][2][r], v3 = acc[j][3][r];
      const long yo = (long)b * a.y_bs + o;
      const float* rp = (!PWX_RINIT && a.R) ? a.R + (long)b * a.r_bs + o : nullptr;
      if (whole) {
        if (rp) {
          const f32x4 r4 = load4u(rp);
          v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
        }
        if constexpr (YB) {
          h4u hq;
          hq.x = f32_to_bf16(v0); hq.y = f32_to_bf16(v1); hq.z = f32_to_bf16(v2); hq.w = f32_to_bf16(v3);
          *reinterpret_cast<h4u*>(reinterpret_cast<bf16_t*>(a.Y) + yo) = hq;
        } else {
          store4u(reinterpret_cast<float*>(a.Y) + yo, f32x4{v0, v1, v2, v3});
        }
      } else {
        const float vv[4] = {v0, v1, v2, v3};
        for (int e = 0; e < 4; ++e)
          if (pq + e >= p0o) {
            const float t = vv[e] + (rp ? rp[e] : 0.f);
            if constexpr (YB) reinterpret_cast<bf16_t*>(a.Y)[yo + e] = f32_to_bf16(t);
            else reinterpret_cast<float*>(a.Y)[yo + e] = t;
          }
      }
    }
  }
}

struct PwxPlan {
  int KB, MT, WM, MTW, chunks, tiles_per_sample, cpg;
};

inline PwxPlan pwx_plan(int M, int K, long HW) {
  PwxPlan p;
  p.cpg = pwx_cpg(K);
  p.KB = (K + 4 * p.cpg - 1) / (4 * p.cpg);
  p.MT = (M + 15) / 16;
  // a wave holds at most PWX_MAX_MTW channel tiles (16 accumulator + 12 weight registers each); wider layers put 2 or 4
  // waves on a pixel group (each splits the group again), and past 4 x that the block column is cut into chunks (grid.y)
  // that re-read the activations
  constexpr int T = PWX_MAX_MTW;
  p.WM = p.MT <= T ? 1 : (p.MT <= 2 * T ? 2 : 4);
  p.chunks = (p.MT + p.WM * T - 1) / (p.WM * T);
  p.MTW = (p.MT + p.WM * p.chunks - 1) / (p.WM * p.chunks);
  const int block_px = (4 / p.WM) * 64;
  p.tiles_per_sample = (int)((HW + block_px - 1) / block_px);
  return p;
}

template <int MTW, int WM, int WL, int XL, bool XB, bool YB>
void launch_pwx2(const PwxArgs& a, const PwxPlan& p, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * p.tiles_per_sample), (unsigned)p.chunks);
  if (p.cpg == 6) hipLaunchKernelGGL((pwx_kernel<MTW, WM, 6, WL, XL, XB, YB>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((pwx_kernel<MTW, WM, 8, WL, XL, XB, YB>), grid, dim3(kThreads), 0, s, a);
}

template <int MTW, int WL, int XL, bool XB, bool YB>
void launch_pwx1(const PwxArgs& a, const PwxPlan& p, hipStream_t s) {
  if (p.WM == 1) launch_pwx2<MTW, 1, WL, XL, XB, YB>(a, p, s);
  else if (p.WM == 2) launch_pwx2<MTW, 2, WL, XL, XB, YB>(a, p, s);
  else launch_pwx2<MTW, 4, WL, XL, XB, YB>(a, p, s);
}

// instantiated: fp32 tensors with (3, 3) and (1, 1) levels; bf16-stored activations and / or outputs with (1, 1)
template <int MTW>
int launch_pwx(const PwxArgs& a, const PwxPlan& p, int wl, int xl, int x_dt, int y_dt, hipStream_t s) {
  if (wl == 3 && xl == 3 && !x_dt && !y_dt) launch_pwx1<MTW, 3, 3, false, false>(a, p, s);
  else if (wl == 1 && xl == 1 && !x_dt && !y_dt) launch_pwx1<MTW, 1, 1, false, false>(a, p, s);
  else if (wl == 1 && xl == 1 && x_dt && !y_dt) launch_pwx1<MTW, 1, 1, true, false>(a, p, s);
  else if (wl == 1 && xl == 1 && !x_dt && y_dt) launch_pwx1<MTW, 1, 1, false, true>(a, p, s);
  else if (wl == 1 && xl == 1 && x_dt && y_dt) launch_pwx1<MTW, 1, 1, true, true>(a, p, s);
  else return CIDNET_ERR_ARG;
  return CIDNET_OK;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* at least two channel tiles per block column (M > 16): the waves of a block share the split of their pixels */
int cidnet_pw_conv_bf16x3_supported(int M, int K, long HW) {
  return M > 16 && K >= 1 && HW >= 64 && (long)K * HW < (1L << 31) && (long)M * HW < (1L << 31);
}

long cidnet_pw_conv_bf16x3_ws_floats(int B, int M, int K, int per_sample) {
  const int cpg = pwx_cpg(K);
  const long frags = (long)((K + 4 * cpg - 1) / (4 * cpg)) * ((M + 15) / 16) * 3;
  return (per_sample ? (long)B : 1L) * frags * 64 * 4;
}

/* the split alone: Wt -> ws (fragment order); nb = B for per-sample weights (w_bs != 0), else 1 */
int cidnet_pw_conv_bf16x3_prep(const float* Wt, long w_bs, long w_ms, long w_ks, float* ws, long ws_floats, int nb, int M, int K,
                               void* stream) {
  CIDNET_CHECK_ARG(Wt && ws && nb > 0 && M > 0 && K > 0 && (nb == 1 || w_bs != 0));
  if (ws_floats < cidnet_pw_conv_bf16x3_ws_floats(nb, M, K, 1)) return CIDNET_ERR_WS;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  const int cpg = pwx_cpg(K);
  const int KB = (K + 4 * cpg - 1) / (4 * cpg), MT = (M + 15) / 16;
  const long threads = (long)nb * KB * MT * 64;
  hipLaunchKernelGGL(pwx_split_w_kernel, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     Wt, w_bs, w_ms, w_ks, reinterpret_cast<uint4*>(ws), M, K, KB, MT, nb, cpg);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_pw_conv_bf16x3_prep_blocks(int M, int K) {
  const int cpg = pwx_cpg(K);
  const long threads = (long)((K + 4 * cpg - 1) / (4 * cpg)) * ((M + 15) / 16) * 64;
  return (threads + kThreads - 1) / kThreads;
}

int cidnet_pw_conv_bf16x3_prep_batch(const long long* table, int n, long total_blocks, void* stream) {
  CIDNET_CHECK_ARG(table && n > 0 && total_blocks > 0);
  hipLaunchKernelGGL(pwx_split_w_batch_kernel, dim3((unsigned)total_blocks), dim3(kThreads), 0, (hipStream_t)stream, table, n);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

/* the product with weights already split by cidnet_pw_conv_bf16x3_prep (Wprep; per_sample: B consecutive sets) */
int cidnet_pw_conv_bf16x3_pre(const float* X, long x_bs, const float* Wprep, int per_sample, float* Y, long y_bs, const float* R,
                              long r_bs, int B, int M, int K, long HW, void* stream) {
  return cidnet_pw_conv_bf16x3_pre_lv(X, x_bs, Wprep, per_sample, Y, y_bs, R, r_bs, B, M, K, HW, 3, 3, stream);
}

/* w_levels / x_levels: bf16 levels of the weights / activations that enter the products: (3, 3) or (1, 1) */
int cidnet_pw_conv_bf16x3_pre_lv(const float* X, long x_bs, const float* Wprep, int per_sample, float* Y, long y_bs, const float* R,
                                 long r_bs, int B, int M, int K, long HW, int w_levels, int x_levels, void* stream) {
  return cidnet_pw_conv_bf16x3_pre_t(X, CIDNET_F32, x_bs, Wprep, per_sample, Y, CIDNET_F32, y_bs, R, r_bs, B, M, K, HW, w_levels, x_levels, stream);
}

/* typed activations / output (CIDNET_F32 / CIDNET_BF16; bf16 with (1, 1) levels only): the bf16 mode's 1x1 conv */
int cidnet_pw_conv_bf16x3_pre_t(const void* X, int x_dt, long x_bs, const float* Wprep, int per_sample, void* Y, int y_dt, long y_bs,
                                const float* R, long r_bs, int B, int M, int K, long HW, int w_levels, int x_levels, void* stream) {
  CIDNET_CHECK_ARG(X && Wprep && Y && B > 0 && (x_dt | 1) == 1 && (y_dt | 1) == 1);
  if (!cidnet_pw_conv_bf16x3_supported(M, K, HW)) return CIDNET_ERR_SHAPE;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(Wprep) & 15) == 0);
  const PwxPlan p = pwx_plan(M, K, HW);
  hipStream_t s = (hipStream_t)stream;
  PwxArgs a{X, x_bs, reinterpret_cast<const uint4*>(Wprep), per_sample ? (long)p.KB * p.MT * 3 * 64 : 0L, Y, y_bs, R, r_bs, B, M, K, HW,
            p.KB, p.MT, p.tiles_per_sample};
  int rc;
  switch (p.MTW) {
    case 1: rc = launch_pwx<1>(a, p, w_levels, x_levels, x_dt, y_dt, s); break;
    case 2: rc = launch_pwx<2>(a, p, w_levels, x_levels, x_dt, y_dt, s); break;
    case 3: rc = launch_pwx<3>(a, p, w_levels, x_levels, x_dt, y_dt, s); break;
    case 4: rc = launch_pwx<4>(a, p, w_levels, x_levels, x_dt, y_dt, s); break;
    default: rc = launch_pwx<5>(a, p, w_levels, x_levels, x_dt, y_dt, s); break;
  }
  if (rc != CIDNET_OK) return rc;
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_pw_conv_bf16x3(const float* X, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks, float* Y, long y_bs,
                          const float* R, long r_bs, float* ws, long ws_floats, int B, int M, int K, long HW, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && ws && B > 0);
  if (!cidnet_pw_conv_bf16x3_supported(M, K, HW)) return CIDNET_ERR_SHAPE;
  const int per_sample = w_bs != 0;
  const int rc = cidnet_pw_conv_bf16x3_prep(Wt, w_bs, w_ms, w_ks, ws, ws_floats, per_sample ? B : 1, M, K, stream);
  if (rc != CIDNET_OK) return rc;
  return cidnet_pw_conv_bf16x3_pre(X, x_bs, ws, per_sample, Y, y_bs, R, r_bs, B, M, K, HW, stream);
}

}  // extern "C"
