// K13: SpatialAttention of the MSSA variant (net/CIDNet_MSSA.py:10-25):
//   out = x * sigmoid(conv7x7([mean_c(x), max_c(x)]))      (zero pad 3, no bias)
// Forward: one pass computes the two statistics planes (and the arg-max channel for the backward),
// a second computes the 7x7 gate per pixel and scales all channels.  Backward: d(gate) needs
// sum_c g*x per pixel; the 7x7 transposed conv and the 98 weight gradients act on the small
// 2-plane tensors; the final pass assembles g_x.  HBM-bound: x is read twice forward, x and g twice
// backward (C planes each); lanes own 4 consecutive pixels.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kK = 7, kR = 3;

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

// stats[b][0] = mean_c, stats[b][1] = max_c, amax[b] = first arg-max channel
__global__ __launch_bounds__(kThreads) void sa_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                            int* __restrict__ amax, int B, int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const float* xb = x + b * C * HW;
    f32x4 s = ld4(xb, p, n), m = s;
    int am[4] = {0, 0, 0, 0};
    for (int c = 1; c < C; ++c) {
      const f32x4 v = ld4(xb + (long)c * HW, p, n);
      s += v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > m[e]) { m[e] = v[e]; am[e] = c; }
    }
    st4(stats + b * 2 * HW, p, n, s * (1.0f / (float)C));
    st4(stats + b * 2 * HW + HW, p, n, m);
    for (int e = 0; e < n; ++e) amax[b * HW + p + e] = am[e];
  }
}

__device__ __forceinline__ float gate_logit(const float* __restrict__ st, const float* __restrict__ w, int y, int x, int H, int W) {
  const long HW = (long)H * W;
  float acc = 0.f;
  for (int ch = 0; ch < 2; ++ch)
    for (int dy = 0; dy < kK; ++dy) {
      const int yy = y + dy - kR;
      if (yy < 0 || yy >= H) continue;
      const float* row = st + ch * HW + (long)yy * W;
#pragma unroll
      for (int dx = 0; dx < kK; ++dx) {
        const int xx = x + dx - kR;
        if (xx >= 0 && xx < W) acc += w[(ch * kK + dy) * kK + dx] * row[xx];
      }
    }
  return acc;
}

// att = sigmoid(conv7(stats)); out[c] = x[c] * att
__global__ __launch_bounds__(kThreads) void sa_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ w, float* __restrict__ att,
                                                            float* __restrict__ out, int B, int C, int H, int W) {
  __shared__ float ws[2 * kK * kK];
  for (int i = threadIdx.x; i < 2 * kK * kK; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const long HW = (long)H * W;
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < n; ++e) {
      const long q = p + e;
      const float l = gate_logit(stats + b * 2 * HW, ws, (int)(q / W), (int)(q % W), H, W);
      a[e] = 1.0f / (1.0f + expf(-l));
    }
    st4(att + b * HW, p, n, a);
    const float* xb = x + b * C * HW;
    float* ob = out + b * C * HW;
    for (int c = 0; c < C; ++c) st4(ob + (long)c * HW, p, n, ld4(xb + (long)c * HW, p, n) * a);
  }
}

// gl = (sum_c g*x) * att * (1 - att)   (gradient at the gate's pre-sigmoid logit)
__global__ __launch_bounds__(kThreads) void sa_bwd_logit_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                const float* __restrict__ att, float* __restrict__ gl, int B,
                                                                int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const float* xb = x + b * C * HW;
    const float* gb = g + b * C * HW;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) s += ld4(xb + (long)c * HW, p, n) * ld4(gb + (long)c * HW, p, n);
    const f32x4 a = ld4(att + b * HW, p, n);
    st4(gl + b * HW, p, n, s * a * (1.0f - a));
  }
}

// gs[b][ch][y][x] = sum_{dy,dx} w[ch][dy][dx] * gl[b][y-dy+3][x-dx+3];  partial gw[98] per block
__global__ __launch_bounds__(kThreads) void sa_bwd_stats_kernel(const float* __restrict__ gl, const float* __restrict__ stats,
                                                                const float* __restrict__ w, float* __restrict__ gs,
                                                                float* __restrict__ gw_part, int B, int H, int W) {
  __shared__ float ws[2 * kK * kK];
  __shared__ float red[kThreads / 64];
  for (int i = threadIdx.x; i < 2 * kK * kK; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const long HW = (long)H * W;
  const long total = (long)B * HW;
  float gwa[2 * kK * kK];
#pragma unroll
  for (int i = 0; i < 2 * kK * kK; ++i) gwa[i] = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const long iters = (total + stride - 1) / stride;
  for (long k = 0; k < iters; ++k) {
    const long it = k * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= total) continue;
    const long b = it / HW, q = it - b * HW;
    const int y = (int)(q / W), x = (int)(q % W);
    const float* glb = gl + b * HW;
    const float* stb = stats + b * 2 * HW;
    const float gc = glb[q];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int dy = 0; dy < kK; ++dy) {
      const int yo = y - dy + kR;         // output pixel whose tap (dy,dx) reads (y,x)
      const int yi = y + dy - kR;         // input pixel read by tap (dy,dx) of output (y,x)
#pragma unroll
      for (int dx = 0; dx < kK; ++dx) {
        const int xo = x - dx + kR, xi = x + dx - kR;
        if (yo >= 0 && yo < H && xo >= 0 && xo < W) {
          const float gv = glb[(long)yo * W + xo];
          s0 += ws[dy * kK + dx] * gv;
          s1 += ws[(kK + dy) * kK + dx] * gv;
        }
        if (yi >= 0 && yi < H && xi >= 0 && xi < W) {
          gwa[dy * kK + dx] += gc * stb[(long)yi * W + xi];
          gwa[(kK + dy) * kK + dx] += gc * stb[HW + (long)yi * W + xi];
        }
      }
    }
    gs[b * 2 * HW + q] = s0;
    gs[b * 2 * HW + HW + q] = s1;
  }
#pragma unroll
  for (int i = 0; i < 2 * kK * kK; ++i) {
    const float s = block_sum(gwa[i], red);
    if (threadIdx.x == 0) gw_part[(long)blockIdx.x * 2 * kK * kK + i] = s;
  }
}

__global__ void sa_gw_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ gw) {
  const int i = threadIdx.x;
  if (i >= 2 * kK * kK) return;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += part[(long)k * 2 * kK * kK + i];
  gw[i] = s;
}

// gx[c] = g[c]*att + gs_mean / C + (c == amax) * gs_max
__global__ __launch_bounds__(kThreads) void sa_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ att,
                                                            const float* __restrict__ gs, const int* __restrict__ amax,
                                                            float* __restrict__ gx, int B, int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  const float invC = 1.0f / (float)C;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const f32x4 a = ld4(att + b * HW, p, n);
    const f32x4 gm = ld4(gs + b * 2 * HW, p, n) * invC;
    const f32x4 gmax = ld4(gs + b * 2 * HW + HW, p, n);
    int am[4] = {-1, -1, -1, -1};
    for (int e = 0; e < n; ++e) am[e] = amax[b * HW + p + e];
    const float* gb = g + b * C * HW;
    float* ob = gx + b * C * HW;
    for (int c = 0; c < C; ++c) {
      f32x4 v = ld4(gb + (long)c * HW, p, n) * a + gm;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (am[e] == c) v[e] += gmax[e];
      st4(ob + (long)c * HW, p, n, v);
    }
  }
}

inline int grid_for(long items, int cap) {
  long g = (items + kThreads - 1) / kThreads;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}
constexpr int kStatBlocks = 512;

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_sa_fwd(const float* x, const float* w, float* stats, int* amax, float* att, float* out, int B, int C, int H, int W,
                  void* stream) {
  CIDNET_CHECK_ARG(x && w && stats && amax && att && out && B > 0 && C > 0 && H > 0 && W > 0);
  const long HW = (long)H * W;
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for((long)B * ((HW + 3) / 4), 4096);
  hipLaunchKernelGGL(sa_stats_kernel, dim3(grid), dim3(kThreads), 0, s, x, stats, amax, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(sa_apply_kernel, dim3(grid), dim3(kThreads), 0, s, x, stats, w, att, out, B, C, H, W);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_sa_bwd_ws_floats(int B, int H, int W) { return (long)B * H * W * 3 + (long)kStatBlocks * 2 * kK * kK; }

int cidnet_sa_bwd(const float* x, const float* w, const float* stats, const int* amax, const float* att, const float* g,
                  float* gx, float* gw, float* ws, long ws_floats, int B, int C, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(x && w && stats && amax && att && g && gx && gw && ws && B > 0 && C > 0 && H > 0 && W > 0);
  if (ws_floats < cidnet_sa_bwd_ws_floats(B, H, W)) return CIDNET_ERR_WS;
  const long HW = (long)H * W;
  hipStream_t s = (hipStream_t)stream;
  float* gl = ws;
  float* gs = ws + (long)B * HW;
  float* part = ws + (long)B * HW * 3;
  const int grid = grid_for((long)B * ((HW + 3) / 4), 4096);
  hipLaunchKernelGGL(sa_bwd_logit_kernel, dim3(grid), dim3(kThreads), 0, s, x, g, att, gl, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  const int g2 = grid_for((long)B * HW, kStatBlocks);
  hipLaunchKernelGGL(sa_bwd_stats_kernel, dim3(g2), dim3(kThreads), 0, s, gl, stats, w, gs, part, B, H, W);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(sa_gw_reduce_kernel, dim3(1), dim3(128), 0, s, part, g2, gw);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(sa_bwd_x_kernel, dim3(grid), dim3(kThreads), 0, s, g, att, gs, amax, gx, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
