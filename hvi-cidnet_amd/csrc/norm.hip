// K3: channels-first LayerNorm (per-pixel statistics over C), forward and backward.
// Reference: LayerNorm.forward, net/transformer_utils.py:24-29 (biased variance, eps inside sqrt).
//
// NCHW makes the reduction axis (C) the strided one: a lane owns VEC consecutive pixels and walks
// the C planes, so every access is a coalesced row segment and the per-pixel reductions need no
// cross-lane traffic.  The whole C-column of a lane is held in registers (C*VEC = 144 floats for
// the CIDNet widths 36/72/144), so x is read from HBM exactly once; a generic multi-pass kernel
// covers other widths.  The backward is two kernels: a per-pixel one (gx; reads x, gy once) and a
// channel-major one (d weight, d bias; block = one channel x one pixel chunk, fixed-order two-level
// sum => bitwise reproducible, no atomics).
#include <cstdlib>
#include "common.h"
#include "cidnet_hip.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

template <int VEC> struct Vec;
template <> struct Vec<4> {
  typedef f32x4 T;
  static __device__ __forceinline__ T ld(const float* p) { return load4u(p); }
  static __device__ __forceinline__ void st(float* p, T v) { store4u(p, v); }
  static __device__ __forceinline__ T ld(const bf16_t* p) { return ld4t(p, 0, 1); }        // bf16-stored tensors (bf16 mode)
  static __device__ __forceinline__ void st(bf16_t* p, T v) { st4t(p, 0, 1, v); }
  static __device__ __forceinline__ float hsum(T v) { return (v[0] + v[1]) + (v[2] + v[3]); }
  template <int LPP> static __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 1; m < LPP; m <<= 1) v[e] += __shfl_xor(v[e], m);
    return v;
  }
};
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <> struct Vec<2> {
  typedef f32x2 T;
  static __device__ __forceinline__ T ld(const float* p) { f2u v = *reinterpret_cast<const f2u*>(p); T r = {v.x, v.y}; return r; }
  static __device__ __forceinline__ void st(float* p, T v) { f2u s; s.x = v[0]; s.y = v[1]; *reinterpret_cast<f2u*>(p) = s; }
  static __device__ __forceinline__ T ld(const bf16_t* p) { const h2u v = *reinterpret_cast<const h2u*>(p); T r = {bf16_to_f32(v.x), bf16_to_f32(v.y)}; return r; }
  static __device__ __forceinline__ void st(bf16_t* p, T v) { h2u s; s.x = f32_to_bf16(v[0]); s.y = f32_to_bf16(v[1]); *reinterpret_cast<h2u*>(p) = s; }
  static __device__ __forceinline__ float hsum(T v) { return v[0] + v[1]; }
  template <int LPP> static __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int m = 1; m < LPP; m <<= 1) v[e] += __shfl_xor(v[e], m);
    return v;
  }
};
typedef float f32x1 __attribute__((ext_vector_type(1)));
template <> struct Vec<1> {
  typedef float T;
  static __device__ __forceinline__ T ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, T v) { *p = v; }
  static __device__ __forceinline__ T ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, T v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ float hsum(T v) { return v; }
  template <int LPP> static __device__ __forceinline__ T group_sum(T v) {
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) v += __shfl_xor(v, m);
    return v;
  }
};
template <int VEC> __device__ __forceinline__ typename Vec<VEC>::T vrsqrt_eps(typename Vec<VEC>::T v, float invC, float eps);
template <> __device__ __forceinline__ float vrsqrt_eps<1>(float v, float invC, float eps) { return 1.0f / sqrtf(v * invC + eps); }
template <> __device__ __forceinline__ f32x2 vrsqrt_eps<2>(f32x2 v, float invC, float eps) {
  f32x2 r = {1.0f / sqrtf(v[0] * invC + eps), 1.0f / sqrtf(v[1] * invC + eps)};
  return r;
}
template <> __device__ __forceinline__ f32x4 vrsqrt_eps<4>(f32x4 v, float invC, float eps) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = 1.0f / sqrtf(v[e] * invC + eps);
  return r;
}

// ---- register-resident column kernels (HW % VEC == 0) -----------------------------------------
// w2 / bias2 / y2 (all or none): a second LayerNorm MODULE applied to the same tensor (the x-norm of one LCA block and the
// y-norm of its partner, net/LCA.py:79,91): the normalised value is the same, so the tensor is read once and written twice
template <int C, int VEC, class TY = float>                 // TY: storage type of the outputs y / y2 (bf16_t in the bf16 mode)
__global__ __launch_bounds__(kThreads) void ln_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, TY* __restrict__ y,
                                                              const float* __restrict__ w2, const float* __restrict__ bias2,
                                                              TY* __restrict__ y2,
                                                              float* __restrict__ mean, float* __restrict__ rstd, int B,
                                                              long HW, float eps) {
  typedef typename Vec<VEC>::T V;
  const long nq = HW / VEC;
  const long total = (long)B * nq;
  const float invC = 1.0f / (float)C;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) * VEC;
    const float* xb = x + b * C * HW + p;
    V col[C];
#pragma unroll
    for (int c = 0; c < C; ++c) col[c] = Vec<VEC>::ld(xb + (long)c * HW);
    V s = col[0];
#pragma unroll
    for (int c = 1; c < C; ++c) s += col[c];
    const V u = s * invC;
    V v = (col[0] - u) * (col[0] - u);
#pragma unroll
    for (int c = 1; c < C; ++c) { const V d = col[c] - u; v += d * d; }
    const V rs = vrsqrt_eps<VEC>(v, invC, eps);
    TY* yb = y + b * C * HW + p;
#pragma unroll
    for (int c = 0; c < C; ++c) Vec<VEC>::st(yb + (long)c * HW, (col[c] - u) * rs * w[c] + bias[c]);
    if (y2) {
      TY* y2b = y2 + b * C * HW + p;
#pragma unroll
      for (int c = 0; c < C; ++c) Vec<VEC>::st(y2b + (long)c * HW, (col[c] - u) * rs * w2[c] + bias2[c]);
    }
    if (mean) { Vec<VEC>::st(mean + b * HW + p, u); Vec<VEC>::st(rstd + b * HW + p, rs); }
  }
}

// gx = rstd * (g*w - mean_c(g*w) - xhat * mean_c(g*w*xhat)); xhat (and g*w if CACHE_G) stay in
// registers; without CACHE_G the second use of g re-reads it (it was just touched: L1/L2 hit)
template <int C, int VEC, bool CACHE_G>
__global__ __launch_bounds__(kThreads) void ln_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ gy, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ addend,
                                                              float* __restrict__ gx, int B, long HW) {
  typedef typename Vec<VEC>::T V;
  const long nq = HW / VEC;
  const long total = (long)B * nq;
  const float invC = 1.0f / (float)C;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) * VEC;
    const float* xb = x + b * C * HW + p;
    const float* gb = gy + b * C * HW + p;
    const V u = Vec<VEC>::ld(mean + b * HW + p), rs = Vec<VEC>::ld(rstd + b * HW + p);
    V xh[C], gw[CACHE_G ? C : 1];
#pragma unroll
    for (int c = 0; c < C; ++c) xh[c] = Vec<VEC>::ld(xb + (long)c * HW);
    V s1 = u * 0.f, s2 = s1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const V g = Vec<VEC>::ld(gb + (long)c * HW) * w[c];
      if (CACHE_G) gw[c] = g;
      xh[c] = (xh[c] - u) * rs;
      s1 += g;
      s2 += g * xh[c];
    }
    s1 = s1 * invC; s2 = s2 * invC;
    float* gxb = gx + b * C * HW + p;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const V g = CACHE_G ? gw[c] : Vec<VEC>::ld(gb + (long)c * HW) * w[c];
      V o = rs * (g - s1 - xh[c] * s2);
      if (addend) o += Vec<VEC>::ld(addend + b * C * HW + p + (long)c * HW);      // gradient of the residual branch (uniform)
      Vec<VEC>::st(gxb + (long)c * HW, o);
    }
  }
}

// Wide C on small planes (C = 144 on 50x75): one lane per pixel would hold 144 channels (437 VGPRs, one wave per
// SIMD) and the whole level is only 30 000 pixels = 0.46 waves per SIMD.  Here FOUR adjacent lanes share a pixel,
// each keeps C/4 channels in registers, and the two channel sums are combined across the quad (xor 1, xor 2):
// four times the lanes at a quarter of the registers.
template <int CQ, bool CACHE_G>
__global__ __launch_bounds__(kThreads) void ln_bwd_quad_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ gy, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ addend,
                                                               float* __restrict__ gx, int B, long HW) {
  constexpr int C = 4 * CQ;
  const long total = (long)B * HW;
  const float invC = 1.0f / (float)C;
  const int q = threadIdx.x & 3;
  for (long it = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 2; it < total; it += ((long)gridDim.x * blockDim.x) >> 2) {
    const long b = it / HW, p = it - b * HW;
    // uniform channel pointer + one 32-bit per-lane offset (the launcher checks B*C*HW < 2^31): a single offset
    // register serves all 2*CQ loads instead of a 64-bit address pair each
    const unsigned off = (unsigned)((b * C + (long)q * CQ) * HW + p);
    const float* wq = w + q * CQ;
    const float u = mean[b * HW + p], rs = rstd[b * HW + p];
    float xh[CQ], gw[CACHE_G ? CQ : 1];
#pragma unroll
    for (int c = 0; c < CQ; ++c) xh[c] = (x + (long)c * HW)[off];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      const float g = (gy + (long)c * HW)[off] * wq[c];
      if (CACHE_G) gw[c] = g;
      xh[c] = (xh[c] - u) * rs;
      s1 += g;
      s2 += g * xh[c];
    }
    s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
    s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
    s1 *= invC; s2 *= invC;
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      const float g = CACHE_G ? gw[c] : (gy + (long)c * HW)[off] * wq[c];
      float o = rs * (g - s1 - xh[c] * s2);
      if (addend) o += (addend + (long)c * HW)[off];
      (gx + (long)c * HW)[off] = o;
    }
  }
}

// Whole LayerNorm backward in ONE pass for C = 36 / 72: gx, and the per-block partial sums of d weight[c] = sum gy*xhat
// and d bias[c] = sum gy (the separate ln_wb_kernel pass re-read x and gy: 5 tensor passes instead of 3).  With one
// lane per pixel column the extra 2C accumulators do not fit (264 / 360 VGPRs when tried), so FOUR adjacent lanes
// share a pixel vector and each keeps C/4 channels: x-hat, g*w and the accumulators are CQ*VEC + CQ*VEC + 2*CQ
// registers, and the two channel sums are combined across the quad.  part: [gridDim.x][2C] = (dw | db) per block.
// DUAL: two LayerNorm modules were applied to the same x (ln_fwd_reg_kernel's second output): the input gradient is linear
// in g = gy*w, so gx = LN'(gy*w + gy2*w2) in one pass over x; the second module's dw / db partials follow the first's in
// `part` (4 C floats per block).
// TG: storage type of the incoming gradients gy / gy2 (bf16_t in the bf16 mode: they are outputs of 1x1 convs there)
template <int CQ, int VEC, int LPP, bool DUAL = false, class TG = float>      // LPP lanes per pixel vector (4 or 8), CQ = C / LPP channels per lane
__global__ __launch_bounds__(kThreads) void ln_bwd_quad_fused_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const TG* __restrict__ gy, const float* __restrict__ w2,
                                                                     const TG* __restrict__ gy2, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ addend,
                                                                     float* __restrict__ gx, float* __restrict__ part, int B, long HW) {
  typedef typename Vec<VEC>::T V;
  constexpr int C = LPP * CQ;
  constexpr int LOG = LPP == 8 ? 3 : 2;
  constexpr int NP = DUAL ? 2 : 1;
  __shared__ float red[(kThreads / 64) * LPP * 2 * CQ * NP];
  const long nq = HW / VEC;
  const long total = (long)B * nq;
  const float invC = 1.0f / (float)C;
  const int q = threadIdx.x & (LPP - 1);
  const float* wq = w + q * CQ;
  const float* wq2 = DUAL ? w2 + q * CQ : nullptr;
  float aw[CQ], ab[CQ], aw2[DUAL ? CQ : 1], ab2[DUAL ? CQ : 1];
#pragma unroll
  for (int c = 0; c < CQ; ++c) { aw[c] = 0.f; ab[c] = 0.f; if (DUAL) { aw2[c] = 0.f; ab2[c] = 0.f; } }
  for (long it = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> LOG; it < total; it += ((long)gridDim.x * blockDim.x) >> LOG) {
    const long b = it / nq, p = (it - b * nq) * VEC;
    const long base = (b * C + (long)q * CQ) * HW + p;
    const float* xb = x + base;
    const TG* gb = gy + base;
    const V u = Vec<VEC>::ld(mean + b * HW + p), rs = Vec<VEC>::ld(rstd + b * HW + p);
    V xh[CQ], gw[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) xh[c] = Vec<VEC>::ld(xb + (long)c * HW);
    V s1 = u * 0.f, s2 = s1;
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      const V g = Vec<VEC>::ld(gb + (long)c * HW);
      xh[c] = (xh[c] - u) * rs;
      aw[c] += Vec<VEC>::hsum(g * xh[c]);
      ab[c] += Vec<VEC>::hsum(g);
      gw[c] = g * wq[c];
      if (DUAL) {
        const V g2 = Vec<VEC>::ld(gy2 + base + (long)c * HW);
        aw2[c] += Vec<VEC>::hsum(g2 * xh[c]);
        ab2[c] += Vec<VEC>::hsum(g2);
        gw[c] += g2 * wq2[c];
      }
      s1 += gw[c];
      s2 += gw[c] * xh[c];
    }
    s1 = Vec<VEC>::template group_sum<LPP>(s1) * invC;
    s2 = Vec<VEC>::template group_sum<LPP>(s2) * invC;
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      V o = rs * (gw[c] - s1 - xh[c] * s2);
      if (addend) o += Vec<VEC>::ld(addend + base + (long)c * HW);
      Vec<VEC>::st(gx + base + (long)c * HW, o);
    }
  }
  // lanes with equal q: butterfly over the other lane bits, then the four waves through LDS (fixed order)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WS = LPP * 2 * CQ;                          // floats per wave and module in red
#pragma unroll
  for (int c = 0; c < CQ; ++c) {
    float a = aw[c], d = ab[c];
#pragma unroll
    for (int m = LPP; m < 64; m <<= 1) { a += __shfl_xor(a, m); d += __shfl_xor(d, m); }
    if (lane < LPP) { red[(wave * LPP + q) * 2 * CQ + c] = a; red[(wave * LPP + q) * 2 * CQ + CQ + c] = d; }
    if (DUAL) {
      float a2 = aw2[c], d2 = ab2[c];
#pragma unroll
      for (int m = LPP; m < 64; m <<= 1) { a2 += __shfl_xor(a2, m); d2 += __shfl_xor(d2, m); }
      if (lane < LPP) { red[4 * WS + (wave * LPP + q) * 2 * CQ + c] = a2; red[4 * WS + (wave * LPP + q) * 2 * CQ + CQ + c] = d2; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C * NP; i += kThreads) {
    const int mod = i / (2 * C), r = i - mod * 2 * C;
    const int which = r / C, ch = r - which * C;           // 0: dw, 1: db
    const int qq = ch / CQ, k = ch - qq * CQ;
    const int o = mod * 4 * WS + qq * 2 * CQ + which * CQ + k;
    part[(long)blockIdx.x * 2 * C * NP + i] = (red[o] + red[WS + o]) + (red[2 * WS + o] + red[3 * WS + o]);
  }
}

// gw[c], gb[c] = sum over blocks of part[blk][c], part[blk][C + c]: 8 outputs per block, 32 lanes per output
// accumulate != 0: gw / gb += the sums (a LayerNorm module applied several times per step adds up its uses' gradients
// in place, in backward order, instead of through separate autograd accumulation launches)
// `part` rows are `stride` floats apart and this module's 2 C values start at `off` (dual backward: stride 4 C, off 0 / 2 C)
__global__ __launch_bounds__(256) void ln_wb2_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ gw,
                                                            float* __restrict__ gb, int accumulate, int stride, int off) {
  __shared__ float fold[32][9];
  const int ex = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + ex;
  float t0 = 0.f, t1 = 0.f;
  if (i < 2 * C) {
    int k = sl;
    for (; k + 32 < nblk; k += 64) { t0 += part[(long)k * stride + off + i]; t1 += part[(long)(k + 32) * stride + off + i]; }
    if (k < nblk) t0 += part[(long)k * stride + off + i];
  }
  fold[sl][ex] = t0 + t1;
  __syncthreads();
  if (sl == 0 && i < 2 * C) {
    float v[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) v[g] = (fold[4 * g][ex] + fold[4 * g + 1][ex]) + (fold[4 * g + 2][ex] + fold[4 * g + 3][ex]);
    const float tot = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    float* dst = i < C ? gw + i : gb + (i - C);
    *dst = accumulate ? *dst + tot : tot;
  }
}

// ---- generic multi-pass kernels (any C, any HW) ---------------------------------------------------
__device__ __forceinline__ f32x4 ld4(const float* row, long p, int n) {
  if (n == 4) return load4u(row + p);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < n; ++e) v[e] = row[p + e];
  return v;
}
__device__ __forceinline__ void st4(float* row, long p, int n, f32x4 v) {
  if (n == 4) { store4u(row + p, v); return; }
  for (int e = 0; e < n; ++e) row[p + e] = v[e];
}

__global__ __launch_bounds__(kThreads) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int B, int C,
                                                          long HW, float eps) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  const float invC = 1.0f / (float)C;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const float* xb = x + b * C * HW;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) s += ld4(xb + (long)c * HW, p, n);
    const f32x4 u = s * invC;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const f32x4 d = ld4(xb + (long)c * HW, p, n) - u;
      v += d * d;
    }
    const f32x4 rs = vrsqrt_eps<4>(v, invC, eps);
    float* yb = y + b * C * HW;
    for (int c = 0; c < C; ++c) st4(yb + (long)c * HW, p, n, (ld4(xb + (long)c * HW, p, n) - u) * rs * w[c] + bias[c]);
    if (mean) { st4(mean + b * HW, p, n, u); st4(rstd + b * HW, p, n, rs); }
  }
}

__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ gy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ addend,
                                                          float* __restrict__ gx, int B, int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  const float invC = 1.0f / (float)C;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const float* xb = x + b * C * HW;
    const float* gb = gy + b * C * HW;
    const f32x4 u = ld4(mean + b * HW, p, n), rs = ld4(rstd + b * HW, p, n);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    for (int c = 0; c < C; ++c) {
      const f32x4 gw = ld4(gb + (long)c * HW, p, n) * w[c];
      s1 += gw;
      s2 += gw * ((ld4(xb + (long)c * HW, p, n) - u) * rs);
    }
    s1 = s1 * invC; s2 = s2 * invC;
    float* gxb = gx + b * C * HW;
    for (int c = 0; c < C; ++c) {
      const f32x4 xh = (ld4(xb + (long)c * HW, p, n) - u) * rs;
      f32x4 o = rs * (ld4(gb + (long)c * HW, p, n) * w[c] - s1 - xh * s2);
      if (addend) o += ld4(addend + b * C * HW + (long)c * HW, p, n);
      st4(gxb + (long)c * HW, p, n, o);
    }
  }
}

// ---- d weight / d bias: block = (pixel chunk, channel) ---------------------------------------------
__global__ __launch_bounds__(kThreads) void ln_wb_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ part, int B, int C, long HW, int nchunk) {
  __shared__ float red[kThreads / 64];
  const int c = blockIdx.y;
  const long nq = (HW + 3) >> 2;
  float aw = 0.f, ab = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* xp = x + ((long)b * C + c) * HW;
    const float* gp = gy + ((long)b * C + c) * HW;
    const float* mp = mean + (long)b * HW;
    const float* rp = rstd + (long)b * HW;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)nchunk * blockDim.x) {
      const long p = q << 2;
      const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
      const f32x4 g = ld4(gp, p, n);
      const f32x4 xh = (ld4(xp, p, n) - ld4(mp, p, n)) * ld4(rp, p, n);
      const f32x4 t = g * xh;
      aw += (t[0] + t[1]) + (t[2] + t[3]);
      ab += (g[0] + g[1]) + (g[2] + g[3]);
    }
  }
  const float sw = block_sum(aw, red);
  const float sb = block_sum(ab, red);
  if (threadIdx.x == 0) { part[((long)c * nchunk + blockIdx.x) * 2] = sw; part[((long)c * nchunk + blockIdx.x) * 2 + 1] = sb; }
}

__global__ void ln_wb_reduce_kernel(const float* __restrict__ part, int nchunk, int C, float* __restrict__ gw,
                                    float* __restrict__ gb, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int k = 0; k < nchunk; ++k) { a += part[((long)c * nchunk + k) * 2]; b += part[((long)c * nchunk + k) * 2 + 1]; }
  gw[c] = accumulate ? gw[c] + a : a;
  gb[c] = accumulate ? gb[c] + b : b;
}

inline int grid_for(long items) {
  long g = (items + kThreads - 1) / kThreads;
  return (int)(g > kMaxBlocks ? kMaxBlocks : (g < 1 ? 1 : g));
}

inline int wb_chunks(int B, int C, long HW) {
  // aim for ~2048 blocks over (chunk, channel) without dropping below 1024 pixels per thread-block pass
  long per = ((HW + 3) / 4 + kThreads - 1) / kThreads;
  long want = (2048 + C - 1) / C;
  long n = per < want ? per : want;
  return (int)(n < 1 ? 1 : n);
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_ln_cf_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean, float* rstd, int B,
                     int C, long HW, float eps, void* stream) {
  return cidnet_ln_cf_fwd_t(x, weight, bias, y, CIDNET_F32, mean, rstd, B, C, HW, eps, stream);
}

/* shapes whose LayerNorm output / incoming gradient may be STORED as bf16 (y_dt / gy_dt = CIDNET_BF16): the register-resident
 * forward and the fused backward kernels -- CIDNet's 36 / 72 / 144-channel levels */
int cidnet_ln_cf_typed_supported(int B, int C, long HW) {
  return B > 0 && HW > 0 && ((C == 36 && HW % 4 == 0) || (C == 72 && HW % 2 == 0) || (C == 144 && (long)B * HW <= (1L << 18))) ? 1 : 0;
}

int cidnet_ln_cf_fwd_t(const float* x, const float* weight, const float* bias, void* y, int y_dt, float* mean, float* rstd, int B,
                       int C, long HW, float eps, void* stream) {
  CIDNET_CHECK_ARG(x && weight && bias && y && B > 0 && C > 0 && HW > 0 && (y_dt | 1) == 1);
  CIDNET_CHECK_ARG((mean == nullptr) == (rstd == nullptr));
  hipStream_t s = (hipStream_t)stream;
  if (y_dt) {
    if (!cidnet_ln_cf_typed_supported(B, C, HW)) return CIDNET_ERR_SHAPE;
    bf16_t* yh = reinterpret_cast<bf16_t*>(y);
#define CIDNET_LN_FWD_H(CC)                                                                                                        \
    hipLaunchKernelGGL((ln_fwd_reg_kernel<CC, 1, bf16_t>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, yh, \
                       (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, mean, rstd, B, HW, eps)
    if (C == 36) CIDNET_LN_FWD_H(36); else if (C == 72) CIDNET_LN_FWD_H(72); else CIDNET_LN_FWD_H(144);
#undef CIDNET_LN_FWD_H
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  float* yf = reinterpret_cast<float*>(y);
  // C = 36: one pixel per lane (36 registers of column) -- 4 pixels per lane left under two waves per SIMD for the whole
  // launch at 200x300, i.e. one read burst followed by one write burst: 26.8 -> 23.7 us (tools/micro_ln.py)
  if (C == 36)
    hipLaunchKernelGGL((ln_fwd_reg_kernel<36, 1>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, yf,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, mean, rstd, B, HW, eps);
  else if (C == 72)                                           // one pixel per lane, as for C = 36: 21 -> 17 us at 100x150
    hipLaunchKernelGGL((ln_fwd_reg_kernel<72, 1>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, yf,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, mean, rstd, B, HW, eps);
  else if (C == 144)
    hipLaunchKernelGGL((ln_fwd_reg_kernel<144, 1>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, yf,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, mean, rstd, B, HW, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid_for((long)B * ((HW + 3) / 4))), dim3(kThreads), 0, s, x, weight, bias, yf, mean,
                       rstd, B, C, HW, eps);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

constexpr int kFusedBlocks = 1024;   // blocks of ln_bwd_quad_fused_kernel
bool g_ln_unfused = false;           // A/B switch (environment CIDNET_LN_UNFUSED=1 at first use)

long cidnet_ln_cf_bwd_ws_floats(int C) {
  const long a = 2L * 2048 + 4L * C * 64, b = 2L * C * kFusedBlocks;
  return a > b ? a : b;
}

int cidnet_ln_cf_bwd_res(const float* x, const float* weight, const float* gy, const float* mean, const float* rstd,
                         const float* addend, float* gx, float* gw, float* gb, int accumulate, float* ws, long ws_floats, int B,
                         int C, long HW, void* stream) {
  return cidnet_ln_cf_bwd_res_t(x, weight, gy, CIDNET_F32, mean, rstd, addend, gx, gw, gb, accumulate, ws, ws_floats, B, C, HW, stream);
}

/* gy stored as fp32 or bf16 (gy_dt; bf16: cidnet_ln_cf_typed_supported shapes, gx required) */
int cidnet_ln_cf_bwd_res_t(const float* x, const float* weight, const void* gy_, int gy_dt, const float* mean, const float* rstd,
                           const float* addend, float* gx, float* gw, float* gb, int accumulate, float* ws, long ws_floats, int B,
                           int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(x && weight && gy_ && mean && rstd && gw && gb && ws && B > 0 && C > 0 && HW > 0 && (gy_dt | 1) == 1);
  const int nchunk = wb_chunks(B, C, HW);
  if (ws_floats < 2L * C * nchunk) return CIDNET_ERR_WS;
  static const bool env_read = (g_ln_unfused = std::getenv("CIDNET_LN_UNFUSED") != nullptr, true);
  (void)env_read;
  hipStream_t s = (hipStream_t)stream;
  if (gy_dt) {
    if (!gx || !cidnet_ln_cf_typed_supported(B, C, HW) || ws_floats < 2L * C * kFusedBlocks) return CIDNET_ERR_SHAPE;
    const bf16_t* gh = reinterpret_cast<const bf16_t*>(gy_);
    const long lanes = C == 36 ? (long)B * HW : (C == 72 ? (long)B * HW * 4 : (HW % 2 == 0 ? (long)B * HW * 4 : (long)B * HW * 8));
    const int nblk = (int)((lanes + kThreads - 1) / kThreads < kFusedBlocks ? (lanes + kThreads - 1) / kThreads : kFusedBlocks);
#define CIDNET_LN_BWD_H(CQ, VEC, LPP)                                                                                                   \
    hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<CQ, VEC, LPP, false, bf16_t>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gh,         \
                       (const float*)nullptr, (const bf16_t*)nullptr, mean, rstd, addend, gx, ws, B, HW)
    if (C == 36) CIDNET_LN_BWD_H(9, 4, 4); else if (C == 72) CIDNET_LN_BWD_H(9, 2, 8); else if (HW % 2 == 0) CIDNET_LN_BWD_H(18, 2, 8); else CIDNET_LN_BWD_H(18, 1, 8);
#undef CIDNET_LN_BWD_H
    CIDNET_LAUNCH_STATUS();
    hipLaunchKernelGGL(ln_wb2_reduce_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, s, ws, nblk, C, gw, gb, accumulate, 2 * C, 0);
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  const float* gy = reinterpret_cast<const float*>(gy_);
  if (gx && ((C == 36 && HW % 4 == 0) || (C == 72 && HW % 2 == 0) || (C == 144 && (long)B * HW <= (1L << 18))) &&
      ws_floats >= 2L * C * kFusedBlocks && !g_ln_unfused) {
    const long lanes = C == 36 ? (long)B * HW : (C == 72 ? (long)B * HW * 4 : (HW % 2 == 0 ? (long)B * HW * 4 : (long)B * HW * 8));
    const int nblk = (int)((lanes + kThreads - 1) / kThreads < kFusedBlocks ? (lanes + kThreads - 1) / kThreads : kFusedBlocks);
    if (C == 36)
      hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<9, 4, 4>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, (const float*)nullptr,
                         (const float*)nullptr, mean, rstd, addend, gx, ws, B, HW);
    else if (C == 72)                                         // eight lanes x 9 channels per pixel pair: 44.8 -> 39 us at 100x150 (four x 18: more registers, fewer waves)
      hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<9, 2, 8>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, (const float*)nullptr,
                         (const float*)nullptr, mean, rstd, addend, gx, ws, B, HW);
    else if (HW % 2 == 0)                                     // two pixels per lane group: 38.5 -> 30.4 us at 8x144x50x75
      hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<18, 2, 8>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, (const float*)nullptr,
                         (const float*)nullptr, mean, rstd, addend, gx, ws, B, HW);
    else
      hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<18, 1, 8>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, (const float*)nullptr,
                         (const float*)nullptr, mean, rstd, addend, gx, ws, B, HW);
    CIDNET_LAUNCH_STATUS();
    hipLaunchKernelGGL(ln_wb2_reduce_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, s, ws, nblk, C, gw, gb, accumulate, 2 * C, 0);
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  if (gx) {
    if (C == 36 && HW % 2 == 0)
      hipLaunchKernelGGL((ln_bwd_reg_kernel<36, 2, true>), dim3(grid_for((long)B * HW / 2)), dim3(kThreads), 0, s, x, weight, gy, mean,
                         rstd, addend, gx, B, HW);
    else if (C == 72)
      hipLaunchKernelGGL((ln_bwd_reg_kernel<72, 1, true>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, gy, mean,
                         rstd, addend, gx, B, HW);
    else if (C == 144 && (long)B * C * HW < (1L << 31) && (long)B * HW <= (1L << 18))      // small planes: four lanes per pixel
      hipLaunchKernelGGL((ln_bwd_quad_kernel<36, true>), dim3(grid_for((long)B * HW * 4)), dim3(kThreads), 0, s, x, weight, gy,
                         mean, rstd, addend, gx, B, HW);
    else if (C == 144)
      hipLaunchKernelGGL((ln_bwd_reg_kernel<144, 1, false>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, gy,
                         mean, rstd, addend, gx, B, HW);
    else
      hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid_for((long)B * ((HW + 3) / 4))), dim3(kThreads), 0, s, x, weight, gy, mean, rstd,
                         addend, gx, B, C, HW);
    CIDNET_LAUNCH_STATUS();
  }
  hipLaunchKernelGGL(ln_wb_kernel, dim3((unsigned)nchunk, (unsigned)C), dim3(kThreads), 0, s, x, gy, mean, rstd, ws, B, C, HW,
                     nchunk);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(ln_wb_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, s, ws, nchunk, C, gw, gb, accumulate);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_ln_cf_bwd(const float* x, const float* weight, const float* gy, const float* mean, const float* rstd, float* gx,
                     float* gw, float* gb, float* ws, long ws_floats, int B, int C, long HW, void* stream) {
  return cidnet_ln_cf_bwd_res(x, weight, gy, mean, rstd, nullptr, gx, gw, gb, 0, ws, ws_floats, B, C, HW, stream);
}

/* ---- two LayerNorm modules on one tensor (see cidnet_hip.h) ---- */
static bool ln_dual_shape(int B, int C, long HW) {
  return (C == 36 && HW % 4 == 0) || (C == 72 && HW % 2 == 0) || (C == 144 && (long)B * HW <= (1L << 18));
}

int cidnet_ln_cf_dual_supported(int B, int C, long HW) { return B > 0 && HW > 0 && ln_dual_shape(B, C, HW) ? 1 : 0; }

long cidnet_ln_cf_bwd2_ws_floats(int C) { return 4L * C * kFusedBlocks; }

int cidnet_ln_cf_fwd2(const float* x, const float* weight, const float* bias, float* y, const float* weight2, const float* bias2,
                      float* y2, float* mean, float* rstd, int B, int C, long HW, float eps, void* stream) {
  CIDNET_CHECK_ARG(x && weight && bias && y && weight2 && bias2 && y2 && mean && rstd && B > 0 && C > 0 && HW > 0);
  if (!ln_dual_shape(B, C, HW)) return CIDNET_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (C == 36)
    hipLaunchKernelGGL((ln_fwd_reg_kernel<36, 1>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, y, weight2,
                       bias2, y2, mean, rstd, B, HW, eps);
  else if (C == 72)
    hipLaunchKernelGGL((ln_fwd_reg_kernel<72, 1>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, y, weight2,
                       bias2, y2, mean, rstd, B, HW, eps);
  else
    hipLaunchKernelGGL((ln_fwd_reg_kernel<144, 1>), dim3(grid_for((long)B * HW)), dim3(kThreads), 0, s, x, weight, bias, y, weight2,
                       bias2, y2, mean, rstd, B, HW, eps);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_ln_cf_bwd2(const float* x, const float* weight, const float* gy, const float* weight2, const float* gy2,
                      const float* mean, const float* rstd, const float* addend, float* gx, float* gw, float* gb, int accumulate,
                      float* gw2, float* gb2, int accumulate2, float* ws, long ws_floats, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(x && weight && gy && weight2 && gy2 && mean && rstd && gx && gw && gb && gw2 && gb2 && ws && B > 0 && C > 0 && HW > 0);
  if (!ln_dual_shape(B, C, HW)) return CIDNET_ERR_SHAPE;
  if (ws_floats < cidnet_ln_cf_bwd2_ws_floats(C)) return CIDNET_ERR_WS;
  hipStream_t s = (hipStream_t)stream;
  const long lanes = C == 36 ? (long)B * HW : (C == 72 ? (long)B * HW * 4 : (HW % 2 == 0 ? (long)B * HW * 4 : (long)B * HW * 8));
  const int nblk = (int)((lanes + kThreads - 1) / kThreads < kFusedBlocks ? (lanes + kThreads - 1) / kThreads : kFusedBlocks);
  if (C == 36)            // two pixels per lane: 179 VGPRs; four (as in the single-module kernel) need 295 and run no faster than two passes
    hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<9, 2, 4, true>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, weight2, gy2, mean,
                       rstd, addend, gx, ws, B, HW);
  else if (C == 72)
    hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<9, 2, 8, true>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, weight2, gy2, mean,
                       rstd, addend, gx, ws, B, HW);
  else if (HW % 2 == 0)
    hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<18, 2, 8, true>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, weight2, gy2, mean,
                       rstd, addend, gx, ws, B, HW);
  else
    hipLaunchKernelGGL((ln_bwd_quad_fused_kernel<18, 1, 8, true>), dim3(nblk), dim3(kThreads), 0, s, x, weight, gy, weight2, gy2, mean,
                       rstd, addend, gx, ws, B, HW);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(ln_wb2_reduce_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, s, ws, nblk, C, gw, gb, accumulate, 4 * C, 0);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(ln_wb2_reduce_kernel, dim3((2 * C + 7) / 8), dim3(256), 0, s, ws, nblk, C, gw2, gb2, accumulate2, 4 * C, 2 * C);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
