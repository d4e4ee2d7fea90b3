// Dense 3x3 convolution (zero pad 1) with fp32 operands on the BF16 matrix cores: "bf16x3" split products.
//
// Why: on gfx950 the fp32 MFMA (v_mfma_f32_16x16x4_f32) runs on the SIMD's own 32 fp32 lanes -- 64 FLOP/clk/SIMD, the
// VALU rate, and it shares those lanes with every VALU instruction of the kernel -- while
// v_mfma_f32_16x16x32_bf16 runs on the separate matrix cores at 1024 FLOP/clk/SIMD.  An fp32 value is the exact sum of
// three bf16 values  a = a0 + a1 + a2  (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 8 + 8 + 8 significand
// bits, each subtraction exact), and  a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 ab):  six bf16 MFMAs,
// whose products are exact and accumulate in fp32, reproduce the fp32 product to within its own rounding (the dropped
// terms a1b2, a2b1, a2b2 are <= 2^-25 |ab| each).  Six products at 16x the rate: 2.7x the fp32 MFMA's throughput, on
// a pipe the VALU does not contend for.  Same reference semantics as conv3.hip (transformer_utils.py:39,58).
//
// Mapping.  GEMM view  Y[m][px] = sum_k A[m][k] B[k][px],  k = tap * KC + channel  (KC = K rounded up to 8, zero padded),
// 32 k per MFMA: lane (c = lane & 15, g = lane >> 4) holds A[m = c][k0 .. k0+7] and B[k0 .. k0+7][px = c], k0 = 32 kb +
// 8 g, i.e. EIGHT CONSECUTIVE CHANNELS OF ONE TAP -- so the activation tile is staged in LDS pixel-major with the
// channels innermost ([level][row][x][KC] bf16, pre-split into the three levels), and a B fragment is one 16-byte LDS
// read at the tap's pixel offset.  The weights never touch LDS: wave w owns output-channel tile w and keeps its A
// fragments (NKB k-blocks x 3 levels x 4 VGPRs) in registers for the whole block, which walks many tiles (persistent
// grid).  A block = 4 waves = up to 64 output channels; a tile = TH x 32 output pixels (+1 halo); the four waves stage a
// tile together, then each computes its channel tile; two blocks per CU cover each other's staging latency.
// (A producer-wave variant -- three compute waves, one wave staging the next tile into a second LDS buffer, one block per
// CU -- measured 757-1013 us against this version's 531 us at 8x36x400x600: a single wave cannot stage as fast as three
// compute; phase timing of that variant: 35.8 kcycles staging vs 18.7 kcycles compute per tile.)
#include "common.h"

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef C3S_TIMING
// cycles of wave 0 per block: [0] staging, [1] barriers, [2] compute, [4] tiles
__device__ unsigned long long g_c3s_phase[8 * 1024];
#define C3S_T0() unsigned long long t__ = __builtin_amdgcn_s_memtime()
#define C3S_TICK(slot)                                                                  \
  do {                                                                                  \
    const unsigned long long n__ = __builtin_amdgcn_s_memtime();                        \
    if (lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_c3s_phase[8 * blockIdx.x + (slot)] += n__ - t__; \
    t__ = n__;                                                                          \
  } while (0)
#else
#define C3S_T0()
#define C3S_TICK(slot)
#endif

constexpr int kS3Threads = 256;
constexpr int kS3TW = 32;

struct S3Args {
  const float* X; long x_bs;
  const float* Wt; long w_ms, w_ks;   // A[m][k][tap] = Wt[m*w_ms + k*w_ks + tap']
  const float* R; long r_bs;          // optional addend
  float* Y; long y_bs;
  int B, M, K, H, W, flip;
  int tiles_x, tiles_y;
};

// round-to-nearest-even bf16 of a finite fp32, as the fp32 whose low 16 bits are zero (integer arithmetic only)
__device__ __forceinline__ unsigned rne_hi(float v) {
  const unsigned u = __float_as_uint(v);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

__device__ __forceinline__ void split3(float v, bf16_t& h0, bf16_t& h1, bf16_t& h2) {
  const unsigned b0 = rne_hi(v);
  const float r1 = v - __uint_as_float(b0);
  const unsigned b1 = rne_hi(r1);
  const float r2 = r1 - __uint_as_float(b1);
  h0 = (bf16_t)(b0 >> 16); h1 = (bf16_t)(b1 >> 16); h2 = (bf16_t)(rne_hi(r2) >> 16);
}

template <int KC, int TH>
struct S3T {
  static constexpr int NKB = (9 * KC + 31) / 32;               // k-blocks of 32
  static constexpr int PW = kS3TW + 2, PH = TH + 2, NPX = PW * PH;
  // bf16 elements per pixel in LDS.  With the natural stride (KC = 40: 20 dwords) every 16-byte B-fragment read is a 2-way
  // bank conflict (16 consecutive pixels x 4 lane groups that differ by 8 channels); 24 dwords per pixel: 1.2-way.
  static constexpr int PS = KC == 40 ? 48 : KC;
  static constexpr int LEVEL_HALFS = NPX * PS + 8;             // bf16 elements per level (+ pad: keeps 16-byte alignment)
  static constexpr int LDS_BYTES = 3 * LEVEL_HALFS * 2;
  static constexpr int NT = 2 * TH;                            // 16-pixel N-tiles per block tile
};

template <int KC, int TH>
__global__ __launch_bounds__(kS3Threads, 2) void conv3s_kernel(S3Args a) {
  using T = S3T<KC, TH>;
  extern __shared__ __attribute__((aligned(16))) bf16_t xs[];  // [3][PH][PW][KC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, K = a.K, M = a.M;
  const int m0 = blockIdx.y * 64 + wave * 16;                  // this wave's output-channel tile
  const bool wave_live = m0 < M;

  // ---- A fragments of this wave's channel tile, all k-blocks, three levels: registers for the block's lifetime ----
  uint4 A[T::NKB][3];
  int boff[T::NKB];                                            // byte offset of this lane's B fragment inside a level, per k-block
#pragma unroll
  for (int kb = 0; kb < T::NKB; ++kb) {
    const int k0 = 32 * kb + 8 * g;
    const int tap = k0 / KC, ch0 = k0 - tap * KC;
    const bool tap_ok = tap < 9;
    const int dy = tap_ok ? tap / 3 : 0, dx = tap_ok ? tap - 3 * (tap / 3) : 0;
    boff[kb] = ((dy * T::PW + dx) * T::PS + ch0) * 2;
    const int m = m0 + c;
    const int tp = a.flip ? 8 - tap : tap;
    bf16_t h[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = ch0 + j;
      const bool ok = wave_live && m < M && tap_ok && ch < K;
      const float wv = ok ? a.Wt[(long)m * a.w_ms + (long)ch * a.w_ks + tp] : 0.f;
      split3(wv, h[0][j], h[1][j], h[2][j]);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      A[kb][l].x = (unsigned)h[l][0] | ((unsigned)h[l][1] << 16);
      A[kb][l].y = (unsigned)h[l][2] | ((unsigned)h[l][3] << 16);
      A[kb][l].z = (unsigned)h[l][4] | ((unsigned)h[l][5] << 16);
      A[kb][l].w = (unsigned)h[l][6] | ((unsigned)h[l][7] << 16);
    }
  }

  const long HW = (long)H * W;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  const int ntiles = a.B * tiles_per_img;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * kS3TW;
    C3S_T0();
    __syncthreads();                                           // previous tile's reads are done
    if (wave == 0) { C3S_TICK(1); }
    // ---- stage the (TH+2) x 34 input tile: fp32 -> three bf16 levels (exact split by truncation: the top 16 bits of an
    //      fp32 are a bf16 holding its first 8 significand bits, the exact remainder holds the other 16), two channels per
    //      lane and store; the loads of a batch are issued before its first conversion; channel pairs past K are zeros ----
    {
      const float* xb = a.X + (long)b * a.x_bs;              // wave-uniform base; lanes add 32-bit offsets
      constexpr int NPAIR = T::NPX * (KC / 2), TRIPS = (NPAIR + kS3Threads - 1) / kS3Threads;
      const int iHW = (int)HW;
      constexpr int BATCH = 8;                                 // trips whose loads are in flight together
#pragma unroll 1
      for (int t0 = 0; t0 < TRIPS; t0 += BATCH) {
        float v0[BATCH], v1[BATCH];
        unsigned inm = 0;                                      // bits 2u / 2u+1: the pair's elements lie inside the image and below K
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int i = tid + (t0 + u) * kS3Threads;
          const int ic = min(i, NPAIR - 1);
          const int cp = ic / T::NPX, px = ic - cp * T::NPX;   // consecutive lanes: consecutive pixels of one channel pair
          const int py = px / T::PW, pxx = px - py * T::PW;
          const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
          const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
          const int ch = 2 * cp;
          const int o = min(ch, K - 1) * iHW + min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
          if (in && ch < K) inm |= 1u << (2 * u);
          if (in && ch + 1 < K) inm |= 2u << (2 * u);
          v0[u] = xb[o];
          v1[u] = xb[o + (ch + 1 < K ? iHW : 0)];
        }
        __builtin_amdgcn_sched_barrier(0);                     // all loads of the batch issued before the first conversion
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int i = tid + (t0 + u) * kS3Threads;
          if (t0 + u >= TRIPS || i >= NPAIR) continue;
          const int cp = i / T::NPX, px = i - cp * T::NPX;
          unsigned hi[2][3];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float x = ((inm >> (2 * u + e)) & 1) ? (e == 0 ? v0[u] : v1[u]) : 0.f;
            const unsigned b0 = __float_as_uint(x) & 0xFFFF0000u;
            const float r1 = x - __uint_as_float(b0);
            const unsigned b1 = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(b1);
            hi[e][0] = b0; hi[e][1] = b1; hi[e][2] = __float_as_uint(r2) & 0xFFFF0000u;
          }
#pragma unroll
          for (int l = 0; l < 3; ++l)
            *reinterpret_cast<unsigned*>(xs + l * T::LEVEL_HALFS + px * T::PS + 2 * cp) = (hi[0][l] >> 16) | hi[1][l];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (wave == 0) { C3S_TICK(0); }
    __syncthreads();
    if (wave == 0) { C3S_TICK(1); }
    if (wave_live) {
      // ---- compute: this wave's 16 channels x every 16-pixel N-tile of the block tile ----
      const char* xsb = reinterpret_cast<const char*>(xs);
#pragma unroll 2
      for (int nt = 0; nt < T::NT; ++nt) {
        const int row = nt >> 1, xh = nt & 1;
        const int pbase = ((row * T::PW + xh * 16 + c) * T::PS) * 2;   // byte offset of this lane's pixel (tap (0,0)) in a level
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // B fragments are requested one k-block ahead of the MFMAs that use them
        uint4 bn[3];
        {
          const char* p = xsb + pbase + boff[0];
          bn[0] = *reinterpret_cast<const uint4*>(p);
          bn[1] = *reinterpret_cast<const uint4*>(p + T::LEVEL_HALFS * 2);
          bn[2] = *reinterpret_cast<const uint4*>(p + 2 * T::LEVEL_HALFS * 2);
        }
#pragma unroll
        for (int kb = 0; kb < T::NKB; ++kb) {
          const bf16x8 b0 = __builtin_bit_cast(bf16x8, bn[0]), b1 = __builtin_bit_cast(bf16x8, bn[1]), b2 = __builtin_bit_cast(bf16x8, bn[2]);
          if (kb + 1 < T::NKB) {
            const char* p = xsb + pbase + boff[kb + 1];
            bn[0] = *reinterpret_cast<const uint4*>(p);
            bn[1] = *reinterpret_cast<const uint4*>(p + T::LEVEL_HALFS * 2);
            bn[2] = *reinterpret_cast<const uint4*>(p + 2 * T::LEVEL_HALFS * 2);
          }
          const bf16x8 a0 = __builtin_bit_cast(bf16x8, A[kb][0]), a1 = __builtin_bit_cast(bf16x8, A[kb][1]),
                       a2 = __builtin_bit_cast(bf16x8, A[kb][2]);
          // small terms first
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0);
        }
        const int y = y0 + row, x = x0 + xh * 16 + c;
        if (y < H && x < W) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * g + r;
            if (m < M) {
              const long o = (long)m * HW + (long)y * W + x;
              float v = acc[r];
              if (a.R) v += a.R[(long)b * a.r_bs + o];
              a.Y[(long)b * a.y_bs + o] = v;
            }
          }
        }
      }
    }
    if (wave == 0) {
      C3S_TICK(2);
#ifdef C3S_TIMING
      if (lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0) g_c3s_phase[8 * blockIdx.x + 4] += 1;
#endif
    }
  }
}

template <int KC, int TH>
int launch_conv3s(S3Args a, hipStream_t s) {
  using T = S3T<KC, TH>;
  a.tiles_x = (a.W + kS3TW - 1) / kS3TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  static bool attr = false;                                   // idempotent: raises the kernel's dynamic-LDS limit once
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s_kernel<KC, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
    attr = true;
  }
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y;
  const int mchunks = (a.M + 63) / 64;
  long nblk = 512 / mchunks;                                  // persistent: about two resident blocks per CU in total
  if (nblk > ntiles) nblk = ntiles;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL((conv3s_kernel<KC, TH>), dim3((unsigned)nblk, (unsigned)mchunks), dim3(kS3Threads), T::LDS_BYTES, s, a);
  return 0;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* channel counts the split-product kernel is instantiated for (the A fragments of a wave's channel tile must fit its
 * registers: K <= 40) */
#ifdef C3S_TIMING
int cidnet_debug_c3s_phases(unsigned long long* host, int nblocks) {
  (void)hipDeviceSynchronize();
  const size_t n = sizeof(unsigned long long) * 8 * (nblocks < 1024 ? nblocks : 1024);
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c3s_phase), n);
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_c3s_phase)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 8 * 1024);
  return rc;
}
#endif

int cidnet_conv3x3_bf16x3_supported(int M, int K) { return (K == 36 || K == 12 || K == 24) && M >= 1; }

int cidnet_conv3x3_bf16x3(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, const float* R, long r_bs,
                          float* Y, long y_bs, int B, int M, int K, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0);
  if (!cidnet_conv3x3_bf16x3_supported(M, K)) return CIDNET_ERR_SHAPE;
  S3Args a{X, x_bs, Wt, w_ms, w_ks, R, r_bs, Y, y_bs, B, M, K, H, W, flip, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (K == 36) launch_conv3s<40, 4>(a, s);
  else if (K == 24) launch_conv3s<24, 4>(a, s);
  else launch_conv3s<16, 4>(a, s);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
