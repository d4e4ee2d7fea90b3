// Pointwise (1x1) convolution with fp32 operands on the BF16 matrix cores ("bf16x3" split products, see conv3x.hip for the
// arithmetic): for the 1x1 convs of the step that the fp32 MFMA rate bounds -- the coarse levels (100x150 and 50x75 planes
// with 72 .. 766 channels: 45 - 75 TFLOP/s on v_mfma_f32_16x16x4_f32, 1 - 2.5 TB/s).  Same contract as cidnet_pw_conv
// (pw.hip; net/LCA.py:13,15,17,51,57, net/transformer_utils.py:60):
//   Y[b] (M x HW) = A_b (M x K) * X[b] (K x HW) [+ R[b]],   A_b[m][k] = Wt[b*w_bs + m*w_ms + k*w_ks].
// Second version (round 3).  Round 2's pws.hip loaded a lane's eight k values of ONE pixel as eight 4-byte loads and
// stored results as 4-byte scalars in 64-byte runs: it won only where K >> M.  Here every global access moves 16 bytes:
//
//  * MFMA 16x16x32: lane (n = lane & 15, g = lane >> 4) holds A[m = n][8g .. 8g+7] and B[8g .. 8g+7][column n].
//    Column n of N-tile e (e = 0..3) is pixel 4 n + e of the wave's 64-pixel group, so ONE float4 load of channel k at
//    pixels 4n .. 4n+3 supplies element k of all four N-tiles' B fragments, and register r of the four accumulators of a
//    channel tile is the float4  Y[m = 4 g + r][4n .. 4n+3]  -- loads, residual loads and stores are 16 B per lane, 256 B
//    contiguous per channel row.
//  * A block is 4 waves = WM (along output channels) x 4 / WM pixel groups.  The WM waves that share a pixel group share
//    its split: each loads 8 / WM of the eight channels of its lane group, splits them (round to nearest, exact: two
//    channels per v_cvt_pk) and writes its dwords of the twelve B fragments (3 levels x 4 N-tiles) to LDS; everyone reads
//    the fragments back as 16-byte rows.  Double-buffered, one barrier per 32-deep k-block.  (M <= 16 -- a single channel
//    tile -- is left to pw.hip.)
//  * Weights: a small kernel splits them once per call into fragment order in a workspace (L2 resident); a wave streams
//    the fragments of its <= 3 channel tiles one k-block ahead.
// No packed-fp32 / SDWA instructions (hvi-cidnet_amd/build.py).
#include "common.h"
#include <type_traits>

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;

struct PwxArgs {
  const float* X; long x_bs;
  const uint4* Af; long a_bs;          // split weights in fragment order; a_bs = fragments-per-sample * 64 (0: shared)
  float* Y; long y_bs;
  const float* R; long r_bs;
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
__global__ __launch_bounds__(kThreads) void pwx_split_w_kernel(const float* __restrict__ Wt, long w_bs, long w_ms, long w_ks,
                                                               uint4* __restrict__ Af, int M, int K, int KB, int MT, int nb) {
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  const int lane = (int)(idx & 63);
  const long t = idx >> 6;
  if (t >= (long)nb * KB * MT) return;
  const int mt = (int)(t % MT), kb = (int)((t / MT) % KB), b = (int)(t / ((long)MT * KB));
  const int m = mt * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (m < M && k0 + j < K) ? Wt[(long)b * w_bs + (long)m * w_ms + (long)(k0 + j) * w_ks] : 0.f;
  uint4 o[3];
  split3_pair(v[0], v[1], o[0].x, o[1].x, o[2].x);
  split3_pair(v[2], v[3], o[0].y, o[1].y, o[2].y);
  split3_pair(v[4], v[5], o[0].z, o[1].z, o[2].z);
  split3_pair(v[6], v[7], o[0].w, o[1].w, o[2].w);
#pragma unroll
  for (int l = 0; l < 3; ++l) Af[(t * 3 + l) * 64 + lane] = o[l];
}

// Pixel quad p .. p + 3 of a row of HW >= 4 pixels when the quad may cross the row's end: one float4 load at the last
// whole quad of the row (pc = min(p, HW - 4)), shifted down by d = p - pc lanes of the vector; pixels past the end read 0.
// Branch-free (the last 64-pixel group of a plane takes this path as a whole).
__device__ __forceinline__ f32x4 load_quad_clamped(const float* row, long pc, int d) {
  const f32x4 v = load4u(row + pc);
  f32x4 r;
  r[0] = d == 0 ? v[0] : (d == 1 ? v[1] : (d == 2 ? v[2] : (d == 3 ? v[3] : 0.f)));
  r[1] = d == 0 ? v[1] : (d == 1 ? v[2] : (d == 2 ? v[3] : 0.f));
  r[2] = d == 0 ? v[2] : (d == 1 ? v[3] : 0.f);
  r[3] = d == 0 ? v[3] : 0.f;
  return r;
}

template <int MTW, int WM>
__global__ __launch_bounds__(kThreads, 2) void pwx_kernel(PwxArgs a) {
  constexpr int NG = 4 / WM;                                     // pixel groups per block
  constexpr int CW = 8 / WM;                                     // channels of its lane group a wave loads and splits
  constexpr int NP = CW / 2;                                     // channel pairs = dwords of every fragment it owns
  static_assert(WM == 2 || WM == 4, "the waves that share a pixel group share its split");
  __shared__ uint4 fr[2][NG][3][4][64];                          // [buffer][group][level][N-tile][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int b = blockIdx.x / a.tiles_per_sample, tile = blockIdx.x - b * a.tiles_per_sample;
  const long HW = a.HW;
  const long p0 = ((long)tile * NG + wn) * 64;
  const long pq = p0 + 4 * n;                                    // this lane's pixel quad
  const bool px_live = p0 < HW;                                  // wave-uniform
  const int K = a.K;
  const float* Xb = a.X + (long)b * a.x_bs;
  const uint4* Ab = a.Af + (long)b * a.a_bs + lane;
  const int mt0 = blockIdx.y * (WM * MTW) + wm;                  // this wave's tiles: mt0 + j * WM

  f32x4 acc[MTW][4];
#pragma unroll
  for (int j = 0; j < MTW; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[j][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  int mts[MTW];                                                  // slots past the last tile repeat it (loads stay in range)
#pragma unroll
  for (int j = 0; j < MTW; ++j) mts[j] = min(mt0 + j * WM, a.MT - 1);

  f32x4 raw[CW];
  const bool full_group = p0 + 64 <= HW;                         // wave-uniform: every lane's quad lies inside the plane
  const long pq_c = pq < HW - 4 ? pq : HW - 4;                   // (HW >= 4: cidnet_pw_conv_bf16x3_supported)
  const int pq_d = (int)min(pq - pq_c, 4L);
  auto load_raw = [&](int kb) {
    if (full_group) {
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const int k = min(kb * 32 + g * 8 + wm * CW + c, K - 1);   // rows past K: finite data times a zero weight
        raw[c] = load4u(Xb + (long)k * HW + pq);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const int k = min(kb * 32 + g * 8 + wm * CW + c, K - 1);
        raw[c] = load_quad_clamped(Xb + (long)k * HW, pq_c, pq_d);
      }
    }
  };
  // this wave's dwords of the twelve fragments (N-tile e, level l): dword q = channel pair q of its CW channels
  unsigned own[NP][4][3];
  auto split_raw = [&]() {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) split3_pair(raw[2 * q][e], raw[2 * q + 1][e], own[q][e][0], own[q][e][1], own[q][e][2]);
  };
  auto publish = [&](int buf) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned* dst = reinterpret_cast<unsigned*>(&fr[buf][wn][l][e][lane]) + wm * NP;
#pragma unroll
        for (int q = 0; q < NP; ++q) dst[q] = own[q][e][l];
      }
  };
  auto load_a = [&](int kb, uint4 (&A)[MTW][3]) {
    const uint4* Ak = Ab + (long)kb * a.MT * (3 * 64);
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const uint4* Am = Ak + (long)mts[j] * (3 * 64);
      A[j][0] = Am[0]; A[j][1] = Am[64]; A[j][2] = Am[128];
    }
  };
  // One k-block (its fragments are in LDS buffer kb & 1): split the NEXT k-block's activations
  // (loaded one step ago) and publish them into the other buffer, put the weights of the next and the activations of the
  // one after in flight, MFMA burst, barrier.  NEXT / AFTER are compile-time, so the steady-state body is straight-line code.
  auto step = [&](int kb, uint4 (&Ac)[MTW][3], uint4 (&An)[MTW][3], auto next, auto after) {
    constexpr bool NEXT = decltype(next)::value, AFTER = decltype(after)::value;
    if constexpr (NEXT) {
      split_raw();
      publish((kb + 1) & 1);
      load_a(kb + 1, An);
      if constexpr (AFTER) load_raw(kb + 2);
    }
    bf16x8 a0[MTW], a1[MTW], a2[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      a0[j] = __builtin_bit_cast(bf16x8, Ac[j][0]); a1[j] = __builtin_bit_cast(bf16x8, Ac[j][1]); a2[j] = __builtin_bit_cast(bf16x8, Ac[j][2]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint4 q0 = fr[kb & 1][wn][0][e][lane], q1 = fr[kb & 1][wn][1][e][lane], q2 = fr[kb & 1][wn][2][e][lane];
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, q0), b1 = __builtin_bit_cast(bf16x8, q1), b2 = __builtin_bit_cast(bf16x8, q2);
#define CIDNET_PWX_TERM(AL, BL)                                                                            \
  _Pragma("unroll") for (int j = 0; j < MTW; ++j)                                                          \
      acc[j][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL[j], BL, acc[j][e], 0, 0, 0)
      CIDNET_PWX_TERM(a2, b0);                                   // small terms first
      CIDNET_PWX_TERM(a1, b1);
      CIDNET_PWX_TERM(a0, b2);
      CIDNET_PWX_TERM(a1, b0);
      CIDNET_PWX_TERM(a0, b1);
      CIDNET_PWX_TERM(a0, b0);
#undef CIDNET_PWX_TERM
    }
    if constexpr (NEXT) __syncthreads();
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  uint4 A0[MTW][3], A1[MTW][3];
  load_a(0, A0);
  load_raw(0);
  split_raw();
  publish(0);
  if (a.KB > 1) load_raw(1);
  __syncthreads();
  int kb = 0;
  for (; kb + 3 < a.KB; kb += 2) {                               // steady state: both steps have a next and an after-next
    step(kb, A0, A1, yes, yes);
    step(kb + 1, A1, A0, yes, yes);
  }
  const int rem = a.KB - kb;                                     // 1, 2 or 3 k-blocks left, A0 holds the current one
  if (rem == 3) {
    step(kb, A0, A1, yes, yes);
    step(kb + 1, A1, A0, yes, no);
    step(kb + 2, A0, A1, no, no);
  } else if (rem == 2) {
    step(kb, A0, A1, yes, no);
    step(kb + 1, A1, A0, no, no);
  } else {
    step(kb, A0, A1, no, no);
  }
  if (!px_live || pq >= HW) return;                              // after the last barrier
  // ---- epilogue: register r of the four accumulators of tile j = Y[16 mt + 4 g + r][pq .. pq + 3] ----
  const bool full = pq + 3 < HW;
#pragma unroll
  for (int j = 0; j < MTW; ++j) {
    const int mt = mt0 + j * WM;
    if (mt >= a.MT) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 16 + 4 * g + r;
      if (m >= a.M) continue;
      const long o = (long)m * HW + pq;
      float v0 = acc[j][0][r], v1 = acc[j][1][r], v2 = acc[j][2][r], v3 = acc[j][3][r];
      float* yp = a.Y + (long)b * a.y_bs + o;
      const float* rp = a.R ? a.R + (long)b * a.r_bs + o : nullptr;
      if (full) {
        if (rp) {
          const f32x4 r4 = load4u(rp);
          v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
        }
        store4u(yp, f32x4{v0, v1, v2, v3});
      } else {
        const float vv[4] = {v0, v1, v2, v3};
        for (int e = 0; e < 4; ++e)
          if (pq + e < HW) yp[e] = vv[e] + (rp ? rp[e] : 0.f);
      }
    }
  }
}

struct PwxPlan {
  int KB, MT, WM, MTW, chunks, tiles_per_sample;
};

inline PwxPlan pwx_plan(int M, int K, long HW) {
  PwxPlan p;
  p.KB = (K + 31) / 32;
  p.MT = (M + 15) / 16;
  // waves along M: a wave holds at most 3 channel tiles (their weight fragments are double-buffered in registers);
  // small M spends the waves on pixels instead
  p.WM = p.MT <= 6 ? 2 : 4;
  const int per_wave = (p.MT + p.WM - 1) / p.WM;
  p.chunks = (per_wave + 2) / 3;
  p.MTW = (per_wave + p.chunks - 1) / p.chunks;
  const int block_px = (4 / p.WM) * 64;
  p.tiles_per_sample = (int)((HW + block_px - 1) / block_px);
  return p;
}

template <int MTW, int WM>
void launch_pwx2(const PwxArgs& a, const PwxPlan& p, hipStream_t s) {
  hipLaunchKernelGGL((pwx_kernel<MTW, WM>), dim3((unsigned)(a.B * p.tiles_per_sample), (unsigned)p.chunks), dim3(kThreads), 0, s, a);
}

template <int MTW>
void launch_pwx(const PwxArgs& a, const PwxPlan& p, hipStream_t s) {
  if (p.WM == 2) launch_pwx2<MTW, 2>(a, p, s);
  else launch_pwx2<MTW, 4>(a, p, s);
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* at least two channel tiles per block column (M > 16): the waves of a block share the split of their pixels */
int cidnet_pw_conv_bf16x3_supported(int M, int K, long HW) {
  return M > 16 && K >= 1 && HW >= 4 && (long)K * HW < (1L << 31) && (long)M * HW < (1L << 31);
}

long cidnet_pw_conv_bf16x3_ws_floats(int B, int M, int K, int per_sample) {
  const long frags = (long)((K + 31) / 32) * ((M + 15) / 16) * 3;
  return (per_sample ? (long)B : 1L) * frags * 64 * 4;
}

int cidnet_pw_conv_bf16x3(const float* X, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks, float* Y, long y_bs,
                          const float* R, long r_bs, float* ws, long ws_floats, int B, int M, int K, long HW, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && ws && B > 0);
  if (!cidnet_pw_conv_bf16x3_supported(M, K, HW)) return CIDNET_ERR_SHAPE;
  const int per_sample = w_bs != 0;
  if (ws_floats < cidnet_pw_conv_bf16x3_ws_floats(B, M, K, per_sample)) return CIDNET_ERR_WS;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  const PwxPlan p = pwx_plan(M, K, HW);
  hipStream_t s = (hipStream_t)stream;
  uint4* Af = reinterpret_cast<uint4*>(ws);
  const int nb = per_sample ? B : 1;
  const long threads = (long)nb * p.KB * p.MT * 64;
  hipLaunchKernelGGL(pwx_split_w_kernel, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, Wt, w_bs, w_ms,
                     w_ks, Af, M, K, p.KB, p.MT, nb);
  CIDNET_LAUNCH_STATUS();
  PwxArgs a{X, x_bs, Af, per_sample ? (long)p.KB * p.MT * 3 * 64 : 0L, Y, y_bs, R, r_bs, B, M, K, HW, p.KB, p.MT, p.tiles_per_sample};
  switch (p.MTW) {
    case 1: launch_pwx<1>(a, p, s); break;
    case 2: launch_pwx<2>(a, p, s); break;
    default: launch_pwx<3>(a, p, s); break;
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
