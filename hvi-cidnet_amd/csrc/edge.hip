// "Next" row f1: the Laplacian-pyramid EdgeLoss on device.
// Reference: EdgeLoss, loss/losses.py:41-65.  With G = depthwise 5x5 blur, taps outer(k, k), k = [.05 .25 .4 .25 .05],
// replicate padding, and U = "keep the even pixels, times 4, zero elsewhere":
//   laplacian(z) = z - G(U(G(z))),   loss = mean((laplacian(x) - laplacian(y))^2) * weight      (mse_loss, 'mean')
// The operator is linear, so laplacian(x) - laplacian(y) = L(x - y) and d loss / dx = (2 weight / n) L^T(L(x - y)) with
// L^T = I - G^T U G^T; G^T is the adjoint of the replicate-padded blur (border pixels collect the clamped taps).
// Images have 3 channels (46 MB at 8x3x400x600): every pass is a plain one-thread-per-pixel gather, nothing to tune.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 2048;

__device__ __forceinline__ float tap(int d) { return d == 0 ? 0.4f : ((d == 1 || d == -1) ? 0.25f : 0.05f); }
__device__ __forceinline__ int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// m = U(G(x - y))
__global__ __launch_bounds__(kThreads) void edge_down_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             float* __restrict__ m, long planes, int H, int W) {
  const long total = planes * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int px = (int)(i % W), py = (int)((i / W) % H);
    float v = 0.f;
    if (((px | py) & 1) == 0) {
      const long base = (i / ((long)W * H)) * (long)H * W;
      float s = 0.f;
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const long row = base + (long)clampi(py + dy, H) * W;
        float rs = 0.f;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const long o = row + clampi(px + dx, W);
          rs += tap(dx) * (x[o] - y[o]);
        }
        s += tap(dy) * rs;
      }
      v = 4.f * s;
    }
    m[i] = v;
  }
}

// lap = (x - y) - G(m);  per-block partial of sum lap^2
__global__ __launch_bounds__(kThreads) void edge_lap_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ m, float* __restrict__ lap,
                                                            float* __restrict__ part, long planes, int H, int W) {
  __shared__ float red[kThreads / 64];
  const long total = planes * H * W;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int px = (int)(i % W), py = (int)((i / W) % H);
    const long base = (i / ((long)W * H)) * (long)H * W;
    float s = 0.f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
      const long row = base + (long)clampi(py + dy, H) * W;
      float rs = 0.f;
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) rs += tap(dx) * m[row + clampi(px + dx, W)];
      s += tap(dy) * rs;
    }
    const float l = (x[i] - y[i]) - s;
    lap[i] = l;
    acc += l * l;
  }
  const float bs = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = bs;
}

__global__ void edge_finish_kernel(const float* __restrict__ part, int n_part, float scale, float* __restrict__ loss) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) loss[0] = s * scale;
}

// coefficients of the 1-D adjoint at output index q: c[a] multiplies r[q + a - 2] (a = 0..4) and is the sum of the taps
// d whose replicate-clamped target clamp(p + d) is q
__device__ __forceinline__ void adj_coef(int q, int n, float (&c)[5]) {
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const int p = q + a - 2;
    float s = 0.f;
    if (p >= 0 && p < n) {
#pragma unroll
      for (int d = -2; d <= 2; ++d)
        if (clampi(p + d, n) == q) s += tap(d);
    }
    c[a] = s;
  }
}

// out = post(G^T(r)):  MODE 0: U (even pixels x4, zero elsewhere);  MODE 1: gx = gs * (lap - G^T(r)) with r = t2
template <int MODE>
__global__ __launch_bounds__(kThreads) void edge_adj_kernel(const float* __restrict__ r, const float* __restrict__ lap,
                                                            const float* __restrict__ gloss, float scale, float* __restrict__ out,
                                                            long planes, int H, int W) {
  const long total = planes * H * W;
  const float gs = MODE == 1 ? scale * gloss[0] : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int qx = (int)(i % W), qy = (int)((i / W) % H);
    if (MODE == 0 && ((qx | qy) & 1)) { out[i] = 0.f; continue; }
    const long base = (i / ((long)W * H)) * (long)H * W;
    float cy[5], cx[5];
    adj_coef(qy, H, cy);
    adj_coef(qx, W, cx);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      if (cy[a] == 0.f) continue;
      const long row = base + (long)(qy + a - 2) * W;
      float rs = 0.f;
#pragma unroll
      for (int b = 0; b < 5; ++b)
        if (cx[b] != 0.f) rs += cx[b] * r[row + qx + b - 2];
      s += cy[a] * rs;
    }
    out[i] = MODE == 0 ? 4.f * s : gs * (lap[i] - s);
  }
}

inline int grid_for(long n) {
  const long g = (n + kThreads - 1) / kThreads;
  return (int)(g > kBlocks ? kBlocks : (g < 1 ? 1 : g));
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

long cidnet_edge_ws_floats(int B, int C, int H, int W) { return (long)B * C * H * W + kBlocks; }

int cidnet_edge_fwd(const float* x, const float* y, float weight, float* loss, float* lap, float* ws, long ws_floats, int B, int C,
                    int H, int W, void* stream) {
  CIDNET_CHECK_ARG(x && y && loss && lap && ws && B > 0 && C > 0 && H > 0 && W > 0);
  const long n = (long)B * C * H * W;
  if (ws_floats < n + kBlocks) return CIDNET_ERR_WS;
  float* m = ws;
  float* part = ws + n;
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for(n);
  hipLaunchKernelGGL(edge_down_kernel, dim3(grid), dim3(kThreads), 0, s, x, y, m, (long)B * C, H, W);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(edge_lap_kernel, dim3(grid), dim3(kThreads), 0, s, x, y, m, lap, part, (long)B * C, H, W);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(edge_finish_kernel, dim3(1), dim3(256), 0, s, part, grid, weight / (float)n, loss);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_edge_bwd(const float* lap, const float* gloss, float weight, float* gx, float* ws, long ws_floats, int B, int C, int H,
                    int W, void* stream) {
  CIDNET_CHECK_ARG(lap && gloss && gx && ws && B > 0 && C > 0 && H > 0 && W > 0);
  const long n = (long)B * C * H * W;
  if (ws_floats < n) return CIDNET_ERR_WS;
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for(n);
  hipLaunchKernelGGL((edge_adj_kernel<0>), dim3(grid), dim3(kThreads), 0, s, lap, (const float*)nullptr, (const float*)nullptr, 0.f, ws,
                     (long)B * C, H, W);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL((edge_adj_kernel<1>), dim3(grid), dim3(kThreads), 0, s, ws, lap, gloss, 2.f * weight / (float)n, gx, (long)B * C,
                     H, W);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
