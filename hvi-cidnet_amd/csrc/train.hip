// Training-step pieces around the network (reference: train.py:56-73): the L1 loss
// mean(|out - gt|) with its gradient in one pass, and a fused Adam update over one flat parameter
// buffer (torch.optim.Adam semantics, train.py:166) so the 191 parameter tensors cost one launch.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 1024;

__global__ __launch_bounds__(kThreads) void l1_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                                                      float* __restrict__ grad, float* __restrict__ part, long n, float inv_n) {
  __shared__ float red[kThreads / 64];
  float acc = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = load4u(out + 4 * i) - load4u(gt + 4 * i);
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc += fabsf(d[e]);
      g[e] = d[e] > 0.f ? inv_n : (d[e] < 0.f ? -inv_n : 0.f);
    }
    if (grad) store4u(grad + 4 * i, g);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float d = out[i] - gt[i];
    acc += fabsf(d);
    if (grad) grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void l1_finish_kernel(const float* __restrict__ part, int n, float inv_n, float* __restrict__ loss) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) loss[0] = s * inv_n;
}

// y = x * s[0] * mult: the upstream scalar of a loss's backward applied on the device (no host read of s)
__global__ __launch_bounds__(kThreads) void scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float mult,
                                                         float* __restrict__ y, long n) {
  const float f = (s ? s[0] : 1.f) * mult;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = load4u(x + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= f;
    store4u(y + 4 * i, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    y[i] = x[i] * f;
  }
}

// Gradient of the HVI image at its fan-out (net/CIDNet.py:73-77, 119: hvi feeds the HV stem, its third plane feeds the I stem
// and the whole image is the residual of the output): out = ga + gc, plane 2 also + gi.  Any of the three may be null.
__global__ __launch_bounds__(kThreads) void hvi_grad_sum_kernel(const float* __restrict__ ga, const float* __restrict__ gc,
                                                                const float* __restrict__ gi, float* __restrict__ out, long HW,
                                                                long quads_per_plane, long total_quads) {
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total_quads; q += (long)gridDim.x * blockDim.x) {
    const long plane = q / quads_per_plane;                       // b * 3 + c
    const long px = (q - plane * quads_per_plane) * 4;
    const long o = plane * HW + px;
    const int nv = (int)min(4L, HW - px);
    const bool i_plane = gi && plane % 3 == 2;
    const long oi = (plane / 3) * HW + px;
    if (nv == 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ga) v = load4u(ga + o);
      if (gc) { const f32x4 t = load4u(gc + o); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      if (i_plane) { const f32x4 t = load4u(gi + oi); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      store4u(out + o, v);
    } else {
      for (int e = 0; e < nv; ++e) out[o + e] = (ga ? ga[o + e] : 0.f) + (gc ? gc[o + e] : 0.f) + (i_plane ? gi[oi + e] : 0.f);
    }
  }
}

__global__ __launch_bounds__(kThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                        float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

long cidnet_l1_loss_ws_floats(void) { return kBlocks; }

int cidnet_l1_loss(const float* out, const float* gt, float* grad, float* loss, float* ws, long ws_floats, long n,
                   void* stream) {
  CIDNET_CHECK_ARG(out && gt && loss && ws && n > 0);
  if (ws_floats < kBlocks) return CIDNET_ERR_WS;
  long g = ((n + 3) / 4 + kThreads - 1) / kThreads;
  const int grid = (int)(g > kBlocks ? kBlocks : (g < 1 ? 1 : g));
  const float inv_n = 1.0f / (float)n;
  hipLaunchKernelGGL(l1_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, out, gt, grad, ws, n, inv_n);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, grid, inv_n, loss);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_scale(const float* x, const float* s, float mult, float* y, long n, void* stream) {
  CIDNET_CHECK_ARG(x && y && n > 0);
  long g = ((n + 3) / 4 + kThreads - 1) / kThreads;
  const int grid = (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
  hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, x, s, mult, y, n);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_hvi_grad_sum(const float* ga, const float* gc, const float* gi, float* out, int B, long HW, void* stream) {
  CIDNET_CHECK_ARG(out && B > 0 && HW > 0);
  const long qpp = (HW + 3) / 4, total = qpp * 3 * B;
  long g = (total + kThreads - 1) / kThreads;
  const int grid = (int)(g > 8192 ? 8192 : g);
  hipLaunchKernelGGL(hvi_grad_sum_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, ga, gc, gi, out, HW, qpp, total);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, void* stream) {
  CIDNET_CHECK_ARG(p && g && m && v && n > 0 && step > 0);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  long gsz = (n + kThreads - 1) / kThreads;
  const int grid = (int)(gsz > 4096 ? 4096 : gsz);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2_sqrt, grad_scale);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
