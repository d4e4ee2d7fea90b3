// Backward of a 1x1 convolution y = W x in ONE kernel: data gradient gx = W^T gy AND weight gradient dW = gy x^T, so that gy --
// for the IEL project_in the largest tensor of the layer (2h = 190 / 382 / 766 channels against 36 / 72 / 144 of x) -- is read
// from HBM once instead of twice (cidnet_pw_conv* with transposed strides on the main stream + cidnet_pw_wgrad* on the
// weight-gradient stream).  Reference: autograd of nn.Conv2d(k=1) in IEL / CAB, net/LCA.py:13,15,17,51,57.
//
// Arithmetic: fp32 operands split exactly into three bf16 values, six bf16 MFMAs per product term, fp32 accumulation
// ("bf16x3", conv3x.hip).  Every element of gy and x is split ONCE, while its 32-pixel chunk is staged into LDS, and feeds both
// products (the two separate kernels split gy twice).
//
//  * Block = 4 waves, persistent (two per CU), walks 32-pixel chunks.  Per chunk all M rows of gy and all N rows of x are
//    staged: [level][row][32 px] bf16, 80-byte rows (the transposing reads of four lane groups then hit disjoint banks).
//  * Data gradient  D1[n][px] = sum_m W[m][n] gy[m][px]:  k = m.  A = W^T in fragment order, split once per call by
//    pwb_split_w_kernel and held in REGISTERS for the whole launch (wave w owns n-tiles w, w + 4, ..); B fragments = four
//    consecutive ROWS (m) of one pixel column: the transposing LDS read (__builtin_amdgcn_ds_read_tr16_b64).  Stored per chunk.
//  * Weight gradient  D2[m][n] += sum_px gy[m][px] x[n][px]:  k = the chunk's 32 pixels = one MFMA depth; both operands are
//    plain 16-byte row reads.  Wave w owns m-tiles w, w + 4, ..; the accumulators live in registers for the whole launch, one
//    slab per block at the end, a second kernel sums the slabs in fixed order (bitwise reproducible).
// All LDS reads are compiler builtins (no inline assembly): the compiler places every wait.  No packed-fp32 / SDWA (build.py).
#include "common.h"

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBThreads = 256;
constexpr int kBP = 32;                                      // pixels per chunk = one MFMA depth of the weight gradient
constexpr int kBPitch = 80;                                  // bytes per staged row and level (64 of data)
constexpr int kBMaxBlocks = 512;

struct PwbArgs {
  const float* gy; long gy_bs;
  const float* x; long x_bs;
  const uint4* wf;                     // W^T in fragment order (pwb_split_w_kernel)
  float* gx; long gx_bs;
  float* slabs;                        // [block][M][N]
  int B, M, N; long HW;
  int chunks, nchunks_all;             // 32-pixel chunks per sample, B * chunks
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

// wf[(nt * KB + kb) * 3 + level][lane]: lane (r = lane & 15, g = lane >> 4) holds, for row n = 16 nt + r of W^T, the eight k
// (= m) 32 kb + 8 g .. + 7; zero past M or N.
__global__ __launch_bounds__(256) void pwb_split_w_kernel(const float* __restrict__ Wt, int M, int N, int KB, uint4* __restrict__ wf,
                                                          int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, f = idx >> 6;
  const int kb = f % KB, nt = f / KB;
  const int n = 16 * nt + (lane & 15), m0 = 32 * kb + 8 * (lane >> 4);
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = (n < N && m0 + c < M) ? Wt[(long)(m0 + c) * N + n] : 0.f;
  uint4 o[3];
  split3_pair(v[0], v[1], o[0].x, o[1].x, o[2].x);
  split3_pair(v[2], v[3], o[0].y, o[1].y, o[2].y);
  split3_pair(v[4], v[5], o[0].z, o[1].z, o[2].z);
  split3_pair(v[6], v[7], o[0].w, o[1].w, o[2].w);
  uint4* dst = wf + (long)f * 3 * 64 + lane;
  dst[0] = o[0]; dst[64] = o[1]; dst[128] = o[2];
}

__device__ __forceinline__ bf16x8 frag_of(uint4 q) { return __builtin_bit_cast(bf16x8, q); }
__device__ __forceinline__ bf16x8 frag_of(s16x4 lo, s16x4 hi) {
  const unsigned long long a = __builtin_bit_cast(unsigned long long, lo), b = __builtin_bit_cast(unsigned long long, hi);
  const u32x4 q = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
  return __builtin_bit_cast(bf16x8, q);
}

// six split products, small terms first
#define PWB_MFMA6(acc, a0, a1, a2, b0, b1, b2)                                   \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, acc, 0, 0, 0);           \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);           \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, acc, 0, 0, 0);           \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc, 0, 0, 0);           \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc, 0, 0, 0);           \
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0)

template <int MT, int NT>
struct PwbShape {
  static constexpr int KB = (MT + 1) / 2;                    // 32-row k-blocks of the data gradient
  static constexpr int ROWS_Y = 32 * KB, ROWS_X = 16 * NT;
  static constexpr int LY = ROWS_Y * kBPitch, LX = ROWS_X * kBPitch;    // bytes per level
  static constexpr int LDS = 3 * (LY + LX);
  static constexpr int NTW = (NT + 3) / 4, MTW = (MT + 3) / 4;          // tiles per wave
};

template <int MT, int NT>
__global__ __launch_bounds__(kBThreads, 2) void pwb_kernel(PwbArgs a) {
  using S = PwbShape<MT, NT>;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ys = smem;                                  // [3][ROWS_Y][80]
  unsigned char* xs = smem + 3 * S::LY;                      // [3][ROWS_X][80]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4, q = r >> 2, p = r & 3;
  const int M = a.M, N = a.N;
  const long HW = a.HW;

  // rows past M / N (and the pad bytes) are read with zero weights / never stored: they must hold finite values
  for (int i = tid; i < S::LDS / 16; i += kBThreads) reinterpret_cast<uint4*>(smem)[i] = uint4{0u, 0u, 0u, 0u};

  // W^T fragments of this wave's n-tiles: registers for the whole launch
  uint4 wf[S::NTW][S::KB][3];
#pragma unroll
  for (int j = 0; j < S::NTW; ++j) {
    const int nt = wave + 4 * j;
#pragma unroll
    for (int kb = 0; kb < S::KB; ++kb)
#pragma unroll
      for (int l = 0; l < 3; ++l)
        wf[j][kb][l] = nt < NT ? a.wf[((long)(nt * S::KB + kb) * 3 + l) * 64 + lane] : uint4{0u, 0u, 0u, 0u};
  }
  // the fragments have landed BEFORE the chunk loop: otherwise the compiler, which cannot order these loads against the
  // loop's prefetch loads, waits for everything (vmcnt(0)) in front of the loop's first MFMA -- and with it for the prefetch
#pragma unroll
  for (int j = 0; j < S::NTW; ++j)
#pragma unroll
    for (int kb = 0; kb < S::KB; ++kb)
      asm volatile("" :: "v"(frag_of(wf[j][kb][0])), "v"(frag_of(wf[j][kb][1])), "v"(frag_of(wf[j][kb][2])));
  f32x4 wacc[S::MTW][NT];
#pragma unroll
  for (int i = 0; i < S::MTW; ++i)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wacc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int UNITS = 8 * 16 * (MT + NT);                  // (row, pixel quad) units of the padded tile; rows past M / N skip
  constexpr int ROUNDS = (UNITS + kBThreads - 1) / kBThreads;

  // unit = (row, 4 pixels); rows 0 .. M - 1 of gy, then rows 0 .. N - 1 of x.  The loads of the NEXT chunk are requested
  // before the MFMA phase of the current one and waited for (by the compiler) when they are split at the top of the next
  // iteration: one chunk of memory latency is hidden behind the MFMAs and the stores.
  f32x4 v[ROUNDS];
  auto load_chunk = [&](int c) {
    const int b = c / a.chunks;
    const long p0 = (long)(c - b * a.chunks) * kBP;
    const float* gyb = a.gy + (long)b * a.gy_bs;
    const float* xb = a.x + (long)b * a.x_bs;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int u = tid + rd * kBThreads;
      const int row = u >> 3, quad = u & 7;
      const long px = p0 + 4 * quad;
      const bool isy = row < 16 * MT;
      const int rr = isy ? row : row - 16 * MT;
      const bool live = u < UNITS && rr < (isy ? M : N) && px < HW;                  // (HW % 4 == 0: a live quad is whole)
      const float* src = (isy ? gyb : xb) + (long)rr * HW + px;
      v[rd] = live ? load4u(src) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if ((int)blockIdx.x < a.nchunks_all) load_chunk(blockIdx.x);

  for (int c = blockIdx.x; c < a.nchunks_all; c += gridDim.x) {
    const int b = c / a.chunks;
    const long p0 = (long)(c - b * a.chunks) * kBP;
    __syncthreads();                                         // the previous chunk's fragment reads are done (first: the zero fill)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
      const int u = tid + rd * kBThreads;
      const int row = u >> 3, quad = u & 7;
      const bool isy = row < 16 * MT;
      const int rr = isy ? row : row - 16 * MT;
      if (u >= UNITS || rr >= (isy ? M : N)) continue;
      unsigned a0, a1, a2, b0, b1, b2;
      split3_pair(v[rd][0], v[rd][1], a0, a1, a2);
      split3_pair(v[rd][2], v[rd][3], b0, b1, b2);
      unsigned char* dst = (isy ? ys : xs) + rr * kBPitch + quad * 8;
      const int lvl = isy ? S::LY : S::LX;
      *reinterpret_cast<uint2*>(dst) = uint2{a0, b0};
      *reinterpret_cast<uint2*>(dst + lvl) = uint2{a1, b1};
      *reinterpret_cast<uint2*>(dst + 2 * lvl) = uint2{a2, b2};
    }
    __syncthreads();
    // (Measured alternatives, both slower at 190 x 36, 200x300: reloading a round's registers right after its split inside the
    // loop above -- the compiler cannot order the previous iteration's loads against this iteration's and waits vmcnt(0)
    // before every round: 168 -> 265 us; the data gradient computed transposed for 16-byte stores along pixels -- 64 lanes
    // then write 64 different channel rows per instruction: 168 -> 191 us.)
    if (c + (int)gridDim.x < a.nchunks_all) load_chunk(c + gridDim.x);

    // ---- data gradient: this wave's n-tiles x the chunk's two 16-pixel tiles, k = all rows of gy ----
    if (wave < NT) {
      f32x4 dacc[S::NTW][2];
#pragma unroll
      for (int j = 0; j < S::NTW; ++j) { dacc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dacc[j][1] = dacc[j][0]; }
#pragma unroll
      for (int kb = 0; kb < S::KB; ++kb) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          // lane 4 q + p of a 16-lane group addresses row (k) 32 kb + 8 g + 4 h + q, pixels 16 pt + 4 p .. + 3
          const unsigned char* rp = ys + (32 * kb + 8 * g + q) * kBPitch + (16 * pt + 4 * p) * 2;
          bf16x8 bl[3];
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_u8*)(rp + l * S::LY));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(lds_u8*)(rp + l * S::LY + 4 * kBPitch));
            bl[l] = frag_of(lo, hi);
          }
#pragma unroll
          for (int j = 0; j < S::NTW; ++j) {
            if (wave + 4 * j >= NT) continue;
            PWB_MFMA6(dacc[j][pt], frag_of(wf[j][kb][0]), frag_of(wf[j][kb][1]), frag_of(wf[j][kb][2]), bl[0], bl[1], bl[2]);
          }
        }
      }
      // lane (col r, g) holds rows 4 g + reg of each tile: gx[n][p0 + 16 pt + r]
      float* gxb = a.gx + (long)b * a.gx_bs + p0 + r;
#pragma unroll
      for (int j = 0; j < S::NTW; ++j) {
        const int nt = wave + 4 * j;
        if (nt >= NT) continue;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          if (p0 + 16 * pt + r >= HW) continue;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int n = 16 * nt + 4 * g + reg;
            if (n < N) gxb[(long)n * HW + 16 * pt] = dacc[j][pt][reg];
          }
        }
      }
    }

    // ---- weight gradient: this wave's m-tiles x all n-tiles, k = the chunk's 32 pixels ----
    {
      bf16x8 al[S::MTW][3];
#pragma unroll
      for (int i = 0; i < S::MTW; ++i) {
        const int mt = wave + 4 * i;
#pragma unroll
        for (int l = 0; l < 3; ++l)
          al[i][l] = frag_of(mt < MT ? *reinterpret_cast<const uint4*>(ys + l * S::LY + (16 * mt + r) * kBPitch + g * 16) : uint4{0u, 0u, 0u, 0u});
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 bl[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) bl[l] = frag_of(*reinterpret_cast<const uint4*>(xs + l * S::LX + (16 * nt + r) * kBPitch + g * 16));
#pragma unroll
        for (int i = 0; i < S::MTW; ++i) {
          if (wave + 4 * i >= MT) continue;
          PWB_MFMA6(wacc[i][nt], al[i][0], al[i][1], al[i][2], bl[0], bl[1], bl[2]);
        }
      }
    }
  }

  // ---- this block's partial dW: lane (col r, g) holds rows 4 g + reg: slab[m][n] ----
  float* slab = a.slabs + (long)blockIdx.x * M * N;
#pragma unroll
  for (int i = 0; i < S::MTW; ++i) {
    const int mt = wave + 4 * i;
    if (mt >= MT) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = 16 * nt + r;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = 16 * mt + 4 * g + reg;
        if (m < M && n < N) slab[(long)m * N + n] = wacc[i][nt][reg];
      }
    }
  }
}

// dW[i] = sum over the blocks' slabs, fixed order: 32 elements per block, eight partial sums per element folded through LDS
__global__ __launch_bounds__(256) void pwb_reduce_kernel(const float* __restrict__ slabs, int nslab, long ne, float* __restrict__ dW) {
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + e;
  float t0 = 0.f, t1 = 0.f;
  if (i < ne) {
    int k = sub;
    for (; k + 8 < nslab; k += 16) { t0 += slabs[(long)k * ne + i]; t1 += slabs[(long)(k + 8) * ne + i]; }
    if (k < nslab) t0 += slabs[(long)k * ne + i];
  }
  part[sub][e] = t0 + t1;
  __syncthreads();
  if (sub == 0 && i < ne) {
    float t = part[0][e];
#pragma unroll
    for (int s = 1; s < 8; ++s) t += part[s][e];
    dW[i] = t;
  }
}

inline int pwb_blocks(int B, long HW) {
  const long nch = (long)B * ((HW + kBP - 1) / kBP);
  return (int)(nch < kBMaxBlocks ? nch : kBMaxBlocks);
}

template <int MT, int NT>
int launch_pwb(PwbArgs a, const float* Wt, float* ws, hipStream_t s) {
  using S = PwbShape<MT, NT>;
  const int total = NT * S::KB * 64;
  uint4* wf = reinterpret_cast<uint4*>(ws);
  hipLaunchKernelGGL(pwb_split_w_kernel, dim3((total + 255) / 256), dim3(256), 0, s, Wt, a.M, a.N, S::KB, wf, total);
  a.wf = wf;
  a.slabs = ws + (long)NT * S::KB * 3 * 64 * 4;
  static LdsLimit lds;                                        // once per device: this instantiation's dynamic-LDS limit
  if (const hipError_t e = lds.raise(reinterpret_cast<const void*>(&pwb_kernel<MT, NT>), S::LDS); e != hipSuccess) return -(int)e;   // < 0: HIP error
  const int nblk = pwb_blocks(a.B, a.HW);
  hipLaunchKernelGGL((pwb_kernel<MT, NT>), dim3(nblk), dim3(kBThreads), S::LDS, s, a);
  return nblk;
}

// instantiated (16-row tiles of M, of N): the 36 / 72-channel IEL and CAB layers
inline bool pwb_shape(int M, int N, int& MT, int& NT) {
  MT = (M + 15) / 16; NT = (N + 15) / 16;
  return (MT == 12 && NT == 3) || (MT == 3 && NT == 6) || (MT == 3 && NT == 3) || (MT == 5 && NT == 3);
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_pw_bwd_fused_supported(int M, int N, long HW) {
  int MT, NT;
  return M > 0 && N > 0 && HW >= 4 && HW % 4 == 0 && pwb_shape(M, N, MT, NT) ? 1 : 0;
}

long cidnet_pw_bwd_fused_ws_floats(int B, int M, int N, long HW) {
  int MT, NT;
  if (!cidnet_pw_bwd_fused_supported(M, N, HW) || !pwb_shape(M, N, MT, NT)) return 0;
  const long KB = (MT + 1) / 2;
  return (long)NT * KB * 3 * 64 * 4 + (long)pwb_blocks(B, HW) * M * N;
}

int cidnet_pw_bwd_fused(const float* gY, long gy_bs, const float* X, long x_bs, const float* Wt, float* gX, long gx_bs, float* dW,
                        float* ws, long ws_floats, int B, int M, int N, long HW, void* stream) {
  CIDNET_CHECK_ARG(gY && X && Wt && gX && dW && ws && B > 0 && M > 0 && N > 0 && HW > 0);
  int MT, NT;
  if (!cidnet_pw_bwd_fused_supported(M, N, HW) || !pwb_shape(M, N, MT, NT)) return CIDNET_ERR_SHAPE;
  if (ws_floats < cidnet_pw_bwd_fused_ws_floats(B, M, N, HW)) return CIDNET_ERR_WS;
  PwbArgs a{gY, gy_bs, X, x_bs, nullptr, gX, gx_bs, nullptr, B, M, N, HW, (int)((HW + kBP - 1) / kBP), 0};
  a.nchunks_all = B * a.chunks;
  hipStream_t s = (hipStream_t)stream;
  int nblk;
  if (MT == 12 && NT == 3) nblk = launch_pwb<12, 3>(a, Wt, ws, s);
  else if (MT == 3 && NT == 6) nblk = launch_pwb<3, 6>(a, Wt, ws, s);
  else if (MT == 3 && NT == 3) nblk = launch_pwb<3, 3>(a, Wt, ws, s);
  else nblk = launch_pwb<5, 3>(a, Wt, ws, s);
  if (nblk < 0) return -nblk;                                 // the HIP status of the LDS-limit call
  CIDNET_LAUNCH_STATUS();
  const long ne = (long)M * N;
  const long KB = (MT + 1) / 2;
  hipLaunchKernelGGL(pwb_reduce_kernel, dim3((unsigned)((ne + 31) / 32)), dim3(256), 0, s, ws + (long)NT * KB * 3 * 64 * 4, nblk, ne, dW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
