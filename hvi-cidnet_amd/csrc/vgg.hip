// "Next" row f4: pieces of the VGG19 perceptual loss around the dense 3x3 convolutions (which run on conv3.hip).
// Reference: VGGFeatureExtractor.forward, loss/vgg_arch.py:217-239 (range_norm (x+1)/2, ImageNet mean / std, conv + bias,
// ReLU, MaxPool2d(2, 2)) and PerceptualLoss.forward with criterion 'mse', loss/losses.py:126-142 (train.py:192).
// All streaming, HBM-bound kernels; the VGG weights are frozen, so only data gradients exist.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void vgg_norm_kernel(const float* __restrict__ x, float* __restrict__ y, long HW, long total,
                                                            int range_norm) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % 3);
    float v = x[i];
    if (range_norm) v = (v + 1.f) / 2.f;
    y[i] = (v - mean[c]) / stdv[c];
  }
}

__global__ __launch_bounds__(kThreads) void vgg_norm_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx, long HW, long total,
                                                                int range_norm) {
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % 3);
    gx[i] = g[i] * (range_norm ? 0.5f : 1.f) / stdv[c];
  }
}

// pre = x + bias[c] (in place), act = max(pre, 0) (separate buffer, or in place when act == x)
__global__ __launch_bounds__(kThreads) void bias_relu_kernel(float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ act,
                                                             int C, long HW, long planes) {
  const long plane = blockIdx.y;
  if (plane >= planes) return;
  const float b = bias[plane % C];
  float* xp = x + plane * HW;
  float* ap = act ? act + plane * HW : nullptr;
  const long n4 = HW >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = load4u(xp + 4 * i);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += b; r[e] = v[e] > 0.f ? v[e] : 0.f; }
    if (ap != xp) store4u(xp + 4 * i, v);
    if (ap) store4u(ap + 4 * i, r);
  }
  if (blockIdx.x == 0 && threadIdx.x < (HW & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float v = xp[i] + b;
    if (ap != xp) xp[i] = v;
    if (ap) ap[i] = v > 0.f ? v : 0.f;
  }
}

// gx = g where act > 0 (act = post-ReLU value: positive exactly where the pre-activation was)
__global__ __launch_bounds__(kThreads) void relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ act, float* __restrict__ gx,
                                                            long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = load4u(act + 4 * i), gg = load4u(g + 4 * i);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] > 0.f ? gg[e] : 0.f;
    store4u(gx + 4 * i, o);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    gx[i] = act[i] > 0.f ? g[i] : 0.f;
  }
}

// nn.MaxPool2d(kernel_size=2, stride=2): output floor(H/2) x floor(W/2)
__global__ __launch_bounds__(kThreads) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho,
                                                               int Wo, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const long t = i / Wo;
    const int yo = (int)(t % Ho);
    const long plane = t / Ho;
    const float* p = x + plane * (long)H * W + (long)(2 * yo) * W + 2 * xo;
    y[i] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
  }
}

// gradient goes to the FIRST maximum of the window in scan order (ATen's max_pool2d_with_indices); the other inputs
// -- and the last row / column of odd planes, which no window covers -- get zero.  One thread per input 2x2 window.
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                                               int H, int W, int Ho, int Wo, long planes) {
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;          // cells incl. the uncovered odd tail
  const long total = planes * Hc * Wc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xc = (int)(i % Wc);
    const long t = i / Wc;
    const int yc = (int)(t % Hc);
    const long plane = t / Hc;
    const long base = plane * (long)H * W + (long)(2 * yc) * W + 2 * xc;
    const bool in_x = 2 * xc + 1 < W, in_y = 2 * yc + 1 < H;
    if (in_x && in_y) {
      const float v00 = x[base], v01 = x[base + 1], v10 = x[base + W], v11 = x[base + W + 1];
      const float g = gy[plane * (long)Ho * Wo + (long)yc * Wo + xc];
      int k = 0;
      float m = v00;
      if (v01 > m) { m = v01; k = 1; }
      if (v10 > m) { m = v10; k = 2; }
      if (v11 > m) { m = v11; k = 3; }
      gx[base] = k == 0 ? g : 0.f;
      gx[base + 1] = k == 1 ? g : 0.f;
      gx[base + W] = k == 2 ? g : 0.f;
      gx[base + W + 1] = k == 3 ? g : 0.f;
    } else {
      gx[base] = 0.f;
      if (in_x) gx[base + 1] = 0.f;
      if (in_y) gx[base + W] = 0.f;
    }
  }
}

// sum((a - b)^2) block partials, grad = gscale * (a - b)
__global__ __launch_bounds__(kThreads) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ grad,
                                                       float* __restrict__ part, long n, float gscale) {
  __shared__ float red[kThreads / 64];
  float acc = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = load4u(a + 4 * i) - load4u(b + 4 * i);
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc += d[e] * d[e]; g[e] = gscale * d[e]; }
    if (grad) store4u(grad + 4 * i, g);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float d = a[i] - b[i];
    acc += d * d;
    if (grad) grad[i] = gscale * d;
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// loss[0] (+)= scale * sum(part): fixed-order, one block
__global__ void mse_finish_kernel(const float* __restrict__ part, int n, float scale, int accumulate, float* __restrict__ loss) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + s * scale;
}

inline int grid_for(long n, int cap) {
  long g = (n + kThreads - 1) / kThreads;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_vgg_normalize(const float* x, float* y, int range_norm, int B, long HW, void* stream) {
  CIDNET_CHECK_ARG(x && y && B > 0 && HW > 0);
  const long total = (long)B * 3 * HW;
  hipLaunchKernelGGL(vgg_norm_kernel, dim3(grid_for(total, 4096)), dim3(kThreads), 0, (hipStream_t)stream, x, y, HW, total, range_norm);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_vgg_normalize_bwd(const float* g, float* gx, int range_norm, int B, long HW, void* stream) {
  CIDNET_CHECK_ARG(g && gx && B > 0 && HW > 0);
  const long total = (long)B * 3 * HW;
  hipLaunchKernelGGL(vgg_norm_bwd_kernel, dim3(grid_for(total, 4096)), dim3(kThreads), 0, (hipStream_t)stream, g, gx, HW, total, range_norm);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_bias_relu(float* x, const float* bias, float* act, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(x && bias && B > 0 && C > 0 && HW > 0);
  const long planes = (long)B * C;
  if (planes > 65535) return CIDNET_ERR_SHAPE;
  hipLaunchKernelGGL(bias_relu_kernel, dim3(grid_for((HW + 3) / 4, 64), (unsigned)planes), dim3(kThreads), 0, (hipStream_t)stream, x, bias,
                     act, C, HW, planes);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_relu_bwd(const float* g, const float* act, float* gx, long n, void* stream) {
  CIDNET_CHECK_ARG(g && act && gx && n > 0);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((n + 3) / 4, 4096)), dim3(kThreads), 0, (hipStream_t)stream, g, act, gx, n);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_maxpool2_fwd(const float* x, float* y, long planes, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(x && y && planes > 0 && H >= 2 && W >= 2);
  const int Ho = H / 2, Wo = W / 2;
  const long total = planes * Ho * Wo;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total, 8192)), dim3(kThreads), 0, (hipStream_t)stream, x, y, H, W, Ho, Wo, total);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_maxpool2_bwd(const float* x, const float* gy, float* gx, long planes, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(x && gy && gx && planes > 0 && H >= 2 && W >= 2);
  const long total = planes * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 8192)), dim3(kThreads), 0, (hipStream_t)stream, x, gy, gx, H, W, H / 2, W / 2,
                     planes);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_mse_ws_floats(void) { return 1024; }

int cidnet_mse_loss(const float* a, const float* b, float* grad, float* loss, float weight, int accumulate, float* ws, long ws_floats,
                    long n, void* stream) {
  CIDNET_CHECK_ARG(a && b && loss && ws && n > 0);
  if (ws_floats < 1024) return CIDNET_ERR_WS;
  const int grid = grid_for((n + 3) / 4, 1024);
  hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a, b, grad, ws, n, 2.0f * weight / (float)n);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, grid, weight / (float)n, accumulate, loss);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
