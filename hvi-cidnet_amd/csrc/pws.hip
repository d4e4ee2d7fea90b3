// Pointwise (1x1) convolution with fp32 operands on the BF16 matrix cores ("bf16x3" split products), for the launches
// of the step that are bound by the fp32 MFMA rate rather than by HBM: the 1x1 convs of the two coarse levels
// (100x150 and 50x75 planes with 72 .. 766 channels: 55 - 75 TFLOP/s on v_mfma_f32_16x16x4_f32, 1 - 2.5 TB/s).
// Same contract as cidnet_pw_conv (pw.hip; net/LCA.py:13,15,17,51,57, net/transformer_utils.py:60):
//   Y[b] (M x HW) = A_b (M x K) * X[b] (K x HW) [+ R[b]],   A_b[m][k] = Wt[b*w_bs + m*w_ms + k*w_ks].
//
// Arithmetic (see conv3s.hip for the derivation): every fp32 operand is the exact sum of three bf16 values, the six
// significant cross products run on v_mfma_f32_16x16x32_bf16 (products exact, fp32 accumulation, dropped terms <= 2^-25
// relative each), so the result is within fp32 rounding of the fp32-MFMA kernel's -- at 1024 instead of 64 FLOP/clk/SIMD,
// on a pipe the VALU does not share.
//
// Mapping.  32 k per MFMA: lane (c = lane & 15, g = lane >> 4) holds A[m = c][8g .. 8g+7] and B[8g .. 8g+7][px = c].
//  * Weights: a small kernel splits them once per call into MFMA fragment order in a workspace (per k-block, per 16-row
//    tile, per level: 64 lanes x 16 bytes, one coalesced 1 KB read per fragment); the main kernel reads the fragments it
//    needs straight from L2 (the whole set is at most a few hundred KB and shared by every block).
//  * Activations never touch LDS: a lane loads its eight k values of one pixel as eight dword loads (16 consecutive pixels
//    per row segment; the planes these launches run on are cache resident), splits them in registers (truncation: the top
//    16 bits of an fp32 ARE a bf16 with its first 8 significand bits; three rounds give all 24) and packs B fragments.
//  * A block is 4 waves arranged WM (along output channels) x 4/WM (along pixels); a wave owns NT = 4 pixel tiles (64
//    pixels) and up to 3 16-row channel tiles, i.e. up to 12 accumulator tiles, and walks the k-blocks with the next
//    k-block's weight fragments and activation loads in flight behind the MFMA burst.  No barriers, no LDS.
#include "common.h"
#include <type_traits>

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int kNT = 4;                 // 16-pixel tiles per wave

#ifdef PWS_TIMING
// cycles of wave 0 of the first 1024 blocks: [0] LDS fragment reads, [1] split + loads issued, [2] MFMA burst, [3] barrier, [4] k-blocks
__device__ unsigned long long g_pws_phase[8 * 1024];
#define PWS_T0() unsigned long long t__ = __builtin_amdgcn_s_memtime()
#define PWS_TICK(slot)                                                                                       \
  do {                                                                                                       \
    const unsigned long long n__ = __builtin_amdgcn_s_memtime();                                             \
    if (wave == 0 && lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_pws_phase[8 * blockIdx.x + (slot)] += n__ - t__; \
    t__ = n__;                                                                                               \
  } while (0)
#else
#define PWS_T0()
#define PWS_TICK(slot)
#endif

struct PwsArgs {
  const float* X; long x_bs;
  const uint4* Af; long a_bs;          // split weights in fragment order; a_bs = fragments-per-sample * 64 (0: shared)
  float* Y; long y_bs;
  const float* R; long r_bs;
  int B, M, K, HW;
  int KB, MT;                          // k-blocks of 32, 16-row tiles of M
  int WM;                              // waves along M in a block (1, 2 or 4)
  int tiles_per_sample;                // block pixel tiles (4/WM * 64 pixels) per sample
};

// round-to-nearest-even bf16 of a finite fp32, as the fp32 whose low 16 bits are zero
__device__ __forceinline__ unsigned rne_hi(float v) {
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

// ---- weights -> three bf16 levels in fragment order: fragment (b, kb, mt, level) = 64 lanes x uint4 ----
__global__ __launch_bounds__(kThreads) void pws_split_w_kernel(const float* __restrict__ Wt, long w_bs, long w_ms, long w_ks,
                                                               uint4* __restrict__ Af, int M, int K, int KB, int MT, int nb) {
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  const int lane = (int)(idx & 63);
  const long t = idx >> 6;
  if (t >= (long)nb * KB * MT) return;
  const int mt = (int)(t % MT), kb = (int)((t / MT) % KB), b = (int)(t / ((long)MT * KB));
  const int m = mt * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 8;
  unsigned h[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = m < M && k0 + j < K;
    const float v = ok ? Wt[(long)b * w_bs + (long)m * w_ms + (long)(k0 + j) * w_ks] : 0.f;
    const unsigned b0 = rne_hi(v);
    const float r1 = v - __uint_as_float(b0);
    const unsigned b1 = rne_hi(r1);
    const float r2 = r1 - __uint_as_float(b1);
    h[0][j] = b0; h[1][j] = b1; h[2][j] = rne_hi(r2);
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    uint4 f;
    f.x = (h[l][0] >> 16) | h[l][1];
    f.y = (h[l][2] >> 16) | h[l][3];
    f.z = (h[l][4] >> 16) | h[l][5];
    f.w = (h[l][6] >> 16) | h[l][7];
    Af[(t * 3 + l) * 64 + lane] = f;
  }
}

// eight fp32 (k = 8g .. 8g+7 of one pixel) -> one B fragment per level
__device__ __forceinline__ void split_b(const float (&v)[8], uint4 (&f)[3]) {
  unsigned h[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned b0 = __float_as_uint(v[j]) & 0xFFFF0000u;
    const float r1 = v[j] - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b1);
    h[0][j] = b0; h[1][j] = b1; h[2][j] = __float_as_uint(r2) & 0xFFFF0000u;
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    f[l].x = (h[l][0] >> 16) | h[l][1];
    f[l].y = (h[l][2] >> 16) | h[l][3];
    f[l].z = (h[l][4] >> 16) | h[l][5];
    f[l].w = (h[l][6] >> 16) | h[l][7];
  }
}

// Shared-split kernel.  The WM waves that share a pixel group each load and split NT/WM of its four 16-pixel tiles and hand
// the fragments to the others through LDS (double-buffered, one barrier per k-block), so the VALU work of the split is
// done once per block instead of once per wave.  NS = NT / WM pixel tiles split per wave.
template <int MTW, int WM>
__global__ __launch_bounds__(kThreads, 2) void pws_kernel(PwsArgs a) {
  constexpr int NG = 4 / WM;                                     // pixel groups per block
  constexpr int NS = kNT / WM;                                   // pixel tiles this wave splits
  __shared__ uint4 fr[2][NG][kNT][3][64];                       // [buffer][group][tile][level][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int b = blockIdx.x / a.tiles_per_sample, tile = blockIdx.x - b * a.tiles_per_sample;
  const int p0 = (tile * NG + wn) * (kNT * 16);
  const int mt0 = blockIdx.y * (WM * MTW) + wm;                  // this wave's tiles: mt0 + j * WM
  const bool px_live = p0 < a.HW;                                // wave-uniform
  const int HW = a.HW, K = a.K;
  const float* Xb = a.X + (long)b * a.x_bs;
  int pc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) pc[i] = min(p0 + (wm * NS + i) * 16 + c, HW - 1);
  const uint4* Ab = a.Af + (long)b * a.a_bs + lane;

  f32x4 acc[MTW][kNT];
#pragma unroll
  for (int j = 0; j < MTW; ++j)
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  int mts[MTW];                                                  // slots past the last tile repeat it (loads stay in range)
#pragma unroll
  for (int j = 0; j < MTW; ++j) mts[j] = min(mt0 + j * WM, a.MT - 1);

  float raw[NS][8];
  auto load_raw = [&](int kb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = min(kb * 32 + g * 8 + j, K - 1);             // rows past K: finite data times a zero weight
      const int ro = k * HW;
#pragma unroll
      for (int i = 0; i < NS; ++i) raw[i][j] = Xb[ro + pc[i]];
    }
  };
  auto split_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      uint4 f[3];
      split_b(raw[i], f);
#pragma unroll
      for (int l = 0; l < 3; ++l) fr[buf][wn][wm * NS + i][l][lane] = f[l];
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
  // One k-block: fetch this k-block's B fragments from LDS, split the NEXT k-block's activations (loaded one step ago) into
  // the other buffer, put the weights of the next and the activations of the one after in flight, MFMA burst, barrier.
  // NEXT / AFTER are compile-time so that the steady-state loop body is straight-line code: with the loads behind run-time
  // conditions the compiler's wait-count insertion falls back to vmcnt(0) before the first MFMA, i.e. it waits for the
  // loads it has just issued and the prefetch buys nothing (measured: the pre-MFMA section took as long as the burst).
  // Waves of a pixel group past the plane (last block of a sample) run on clamped addresses and store nothing.
  auto step = [&](int kb, uint4 (&Ac)[MTW][3], uint4 (&An)[MTW][3], auto next, auto after) {
    constexpr bool NEXT = decltype(next)::value, AFTER = decltype(after)::value;
    PWS_T0();
    PWS_TICK(0);
    if constexpr (NEXT) {
      split_store((kb + 1) & 1);
      load_a(kb + 1, An);
      if constexpr (AFTER) load_raw(kb + 2);
    }
    PWS_TICK(1);
    // MFMA burst, pixel tile by pixel tile: the tile's three B fragments come from LDS one tile ahead; the six products of
    // a (channel tile, pixel tile) pair form a dependent chain on its accumulator, so the channel tiles are walked inside
    // each product and consecutive MFMAs are independent (small terms first)
    bf16x8 a0[MTW], a1[MTW], a2[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      a0[j] = __builtin_bit_cast(bf16x8, Ac[j][0]); a1[j] = __builtin_bit_cast(bf16x8, Ac[j][1]); a2[j] = __builtin_bit_cast(bf16x8, Ac[j][2]);
    }
    uint4 bn[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) bn[l] = fr[kb & 1][wn][0][l][lane];
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, bn[0]), b1 = __builtin_bit_cast(bf16x8, bn[1]), b2 = __builtin_bit_cast(bf16x8, bn[2]);
      if (nt + 1 < kNT) {
#pragma unroll
        for (int l = 0; l < 3; ++l) bn[l] = fr[kb & 1][wn][nt + 1][l][lane];
      }
#define CIDNET_PWS_TERM(AL, BL)                                                                            \
  _Pragma("unroll") for (int j = 0; j < MTW; ++j)                                                          \
      acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL[j], BL, acc[j][nt], 0, 0, 0)
      CIDNET_PWS_TERM(a2, b0);
      CIDNET_PWS_TERM(a1, b1);
      CIDNET_PWS_TERM(a0, b2);
      CIDNET_PWS_TERM(a1, b0);
      CIDNET_PWS_TERM(a0, b1);
      CIDNET_PWS_TERM(a0, b0);
#undef CIDNET_PWS_TERM
    }
    PWS_TICK(2);
    if constexpr (NEXT) __syncthreads();
    PWS_TICK(3);
#ifdef PWS_TIMING
    if (wave == 0 && lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_pws_phase[8 * blockIdx.x + 4] += 1;
#endif
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  uint4 A0[MTW][3], A1[MTW][3];
  load_a(0, A0);
  load_raw(0);
  split_store(0);
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
  if (!px_live) return;                                         // wave-uniform, after the last barrier
  // ---- epilogue: acc[r] = row 4g + r of the tile, column c ----
#pragma unroll
  for (int j = 0; j < MTW; ++j) {
    const int mt = mt0 + j * WM;
    if (mt >= a.MT) continue;
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
      const int p = p0 + nt * 16 + c;
      if (p >= HW) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt * 16 + 4 * g + r;
        if (m < a.M) {
          const long o = (long)m * HW + p;
          float v = acc[j][nt][r];
          if (a.R) v += a.R[(long)b * a.r_bs + o];
          a.Y[(long)b * a.y_bs + o] = v;
        }
      }
    }
  }
}

struct PwsPlan {
  int KB, MT, WM, MTW, chunks, tiles_per_sample;
};

inline PwsPlan pws_plan(int M, int K, long HW) {
  PwsPlan p;
  p.KB = (K + 31) / 32;
  p.MT = (M + 15) / 16;
  // waves along M: a wave holds at most 3 channel tiles (their weight fragments are double-buffered in registers);
  // small M spends the waves on pixels instead
  p.WM = p.MT <= 3 ? 1 : (p.MT <= 6 ? 2 : 4);
  const int per_wave = (p.MT + p.WM - 1) / p.WM;
  p.chunks = (per_wave + 2) / 3;
  p.MTW = (per_wave + p.chunks - 1) / p.chunks;
  const int block_px = (4 / p.WM) * kNT * 16;
  p.tiles_per_sample = (int)((HW + block_px - 1) / block_px);
  return p;
}

template <int MTW, int WM>
void launch_pws2(const PwsArgs& a, const PwsPlan& p, hipStream_t s) {
  hipLaunchKernelGGL((pws_kernel<MTW, WM>), dim3((unsigned)(a.B * p.tiles_per_sample), (unsigned)p.chunks), dim3(kThreads), 0, s, a);
}

template <int MTW>
void launch_pws(const PwsArgs& a, const PwsPlan& p, hipStream_t s) {
  if (p.WM == 1) launch_pws2<MTW, 1>(a, p, s);
  else if (p.WM == 2) launch_pws2<MTW, 2>(a, p, s);
  else launch_pws2<MTW, 4>(a, p, s);
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

#ifdef PWS_TIMING
int cidnet_debug_pws_phases(unsigned long long* host, int nblocks) {
  (void)hipDeviceSynchronize();
  const size_t n = sizeof(unsigned long long) * 8 * (nblocks < 1024 ? nblocks : 1024);
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pws_phase), n);
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_pws_phase)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 8 * 1024);
  return rc;
}
#endif

int cidnet_pw_conv_bf16x3_supported(int M, int K, long HW) {
  return M >= 1 && K >= 1 && HW >= 1 && (long)K * HW < (1L << 31) && (long)M * HW < (1L << 31);
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
  const PwsPlan p = pws_plan(M, K, HW);
  hipStream_t s = (hipStream_t)stream;
  uint4* Af = reinterpret_cast<uint4*>(ws);
  const int nb = per_sample ? B : 1;
  const long threads = (long)nb * p.KB * p.MT * 64;
  hipLaunchKernelGGL(pws_split_w_kernel, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, Wt, w_bs, w_ms,
                     w_ks, Af, M, K, p.KB, p.MT, nb);
  CIDNET_LAUNCH_STATUS();
  PwsArgs a{X, x_bs, Af, per_sample ? (long)p.KB * p.MT * 3 * 64 : 0L, Y, y_bs, R, r_bs, B, M, K, (int)HW, p.KB, p.MT, p.WM,
            p.tiles_per_sample};
  switch (p.MTW) {
    case 1: launch_pws<1>(a, p, s); break;
    case 2: launch_pws<2>(a, p, s); break;
    default: launch_pws<3>(a, p, s); break;
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
