// K3: channels-first LayerNorm (per-pixel statistics over C), forward and backward.
// Reference: LayerNorm.forward, net/transformer_utils.py:24-29 (biased variance, eps inside sqrt).
//
// NCHW makes the reduction axis (C) the strided one: a lane owns 4 consecutive pixels and walks
// the C planes, so every access is a coalesced 16 B/lane row segment and the per-pixel reductions
// need no cross-lane traffic at all.  The cross-pixel reductions of the backward (d weight, d bias)
// use wavefront shuffles, one private LDS row per wave and a fixed-order two-level sum
// (bitwise reproducible).  HBM-bound: fwd reads x once from HBM (the two re-reads of the 4-pixel
// column hit L1/L2) and writes y; bwd reads x, gy and writes gx.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

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
    f32x4 rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) rs[e] = 1.0f / sqrtf(v[e] * invC + eps);
    float* yb = y + b * C * HW;
    for (int c = 0; c < C; ++c) {
      const f32x4 d = (ld4(xb + (long)c * HW, p, n) - u) * rs;
      st4(yb + (long)c * HW, p, n, d * w[c] + bias[c]);
    }
    if (mean) { st4(mean + b * HW, p, n, u); st4(rstd + b * HW, p, n, rs); }
  }
}

// gx = rstd * (g*w - mean_c(g*w) - xhat * mean_c(g*w*xhat));  per-block partials of gw, gb.
__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ gy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ gx,
                                                          float* __restrict__ part, int B, int C, long HW) {
  extern __shared__ float sm[];                 // [4 waves][2C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* mine = sm + wave * 2 * C;
  for (int i = lane; i < 2 * C; i += 64) mine[i] = 0.f;
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  const long stride = (long)gridDim.x * blockDim.x;
  const long iters = (total + stride - 1) / stride;   // uniform trip count: shuffles need every lane
  const float invC = 1.0f / (float)C;
  for (long i = 0; i < iters; ++i) {
    const long it = i * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = it < total;
    const long b = live ? it / nq : 0, p = live ? (it - b * nq) << 2 : 0;
    const int n = live ? ((HW - p >= 4) ? 4 : (int)(HW - p)) : 0;
    const float* xb = x + b * C * HW;
    const float* gb = gy + b * C * HW;
    f32x4 u = {0.f, 0.f, 0.f, 0.f}, rs = u, s1 = u, s2 = u;
    if (live) { u = ld4(mean + b * HW, p, n); rs = ld4(rstd + b * HW, p, n); }
    for (int c = 0; c < C; ++c) {
      f32x4 g = {0.f, 0.f, 0.f, 0.f}, xh = g;
      if (live) { g = ld4(gb + (long)c * HW, p, n); xh = (ld4(xb + (long)c * HW, p, n) - u) * rs; }
      const f32x4 gw = g * w[c];
      s1 += gw;
      s2 += gw * xh;
      const f32x4 t = g * xh;
      const float pw = wave_sum((t[0] + t[1]) + (t[2] + t[3]));
      const float pb = wave_sum((g[0] + g[1]) + (g[2] + g[3]));
      if (lane == 0) { mine[c] += pw; mine[C + c] += pb; }
    }
    if (live && gx) {
      s1 = s1 * invC; s2 = s2 * invC;
      float* gxb = gx + b * C * HW;
      for (int c = 0; c < C; ++c) {
        const f32x4 g = ld4(gb + (long)c * HW, p, n);
        const f32x4 xh = (ld4(xb + (long)c * HW, p, n) - u) * rs;
        st4(gxb + (long)c * HW, p, n, rs * (g * w[c] - s1 - xh * s2));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x)
    part[(long)blockIdx.x * 2 * C + i] = (sm[i] + sm[2 * C + i]) + (sm[4 * C + i] + sm[6 * C + i]);
}

__global__ void ln_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ gw,
                                 float* __restrict__ gb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * C) return;
  float t = 0.f;
  for (int k = 0; k < nblk; ++k) t += part[(long)k * 2 * C + i];
  if (i < C) gw[i] = t; else gb[i - C] = t;
}

inline int grid_for(int B, long HW) {
  const long quads = (long)B * ((HW + 3) >> 2);
  long g = (quads + kThreads - 1) / kThreads;
  return (int)(g > kMaxBlocks ? kMaxBlocks : (g < 1 ? 1 : g));
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_ln_cf_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean, float* rstd, int B,
                     int C, long HW, float eps, void* stream) {
  CIDNET_CHECK_ARG(x && weight && bias && y && B > 0 && C > 0 && HW > 0);
  CIDNET_CHECK_ARG((mean == nullptr) == (rstd == nullptr));
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid_for(B, HW)), dim3(kThreads), 0, (hipStream_t)stream, x, weight, bias, y,
                     mean, rstd, B, C, HW, eps);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_ln_cf_bwd_ws_floats(int C) { return (long)kMaxBlocks * 2 * C; }

int cidnet_ln_cf_bwd(const float* x, const float* weight, const float* gy, const float* mean, const float* rstd, float* gx,
                     float* gw, float* gb, float* ws, long ws_floats, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(x && weight && gy && mean && rstd && gw && gb && ws && B > 0 && C > 0 && HW > 0);
  if (ws_floats < cidnet_ln_cf_bwd_ws_floats(C)) return CIDNET_ERR_WS;
  const int grid = grid_for(B, HW);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid), dim3(kThreads), (size_t)8 * C * sizeof(float), (hipStream_t)stream, x,
                     weight, gy, mean, rstd, gx, ws, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(ln_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, grid, C, gw, gb);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
