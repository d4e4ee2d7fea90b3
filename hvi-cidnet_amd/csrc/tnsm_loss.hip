// The TNSM variant's extra training objective (reference: train_tnsm.py:68-72, inline in the training script):
//   consistency = mean(|noise_map - (1 - sigmoid(mean_c |output_rgb - im1|))|)          (target broadcast over the map's channels)
//   smoothing   = mean(|nm[.., x] - nm[.., x+1]|) + mean(|nm[.., y, :] - nm[.., y+1, :]|)
//   loss        = weight * (consistency + smoothing)
// One pass per pixel: the loss partial of the pixel (fixed-order block partials, no atomics) and BOTH gradients -- wrt
// the fused noise map (own term + the four neighbour differences, gather form) and wrt output_rgb (through the
// consistency target; im1 is the network input and gets no gradient in the reference's step either).
// d|t|/dt = sign(t) with sign(0) = 0, as torch.abs' backward.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 1024;

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(kThreads) void tnsm_noise_loss_kernel(const float* __restrict__ nm, const float* __restrict__ out,
                                                                   const float* __restrict__ im, float* __restrict__ g_nm,
                                                                   float* __restrict__ g_out, float* __restrict__ part, int B, int C,
                                                                   int H, int W, float weight) {
  __shared__ float red[kThreads / 64];
  const long HW = (long)H * W, npx = (long)B * HW;
  const float inv_c = weight / (float)((double)B * C * HW);
  const float inv_x = W > 1 ? weight / (float)((double)B * C * H * (W - 1)) : 0.f;
  const float inv_y = H > 1 ? weight / (float)((double)B * C * (H - 1) * W) : 0.f;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const long p = i - (long)b * HW;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const float* ob = out + (long)b * 3 * HW + p;
    const float* ib = im + (long)b * 3 * HW + p;
    const float d0 = ob[0] - ib[0], d1 = ob[HW] - ib[HW], d2 = ob[2 * HW] - ib[2 * HW];
    const float d = (fabsf(d0) + fabsf(d1) + fabsf(d2)) * (1.f / 3.f);
    const float sg = 1.f / (1.f + expf(-d));
    const float t = 1.f - sg;
    float ssum = 0.f;                                            // sum over the map's channels of sign(nm - t)
    for (int c = 0; c < C; ++c) {
      const float* np = nm + ((long)b * C + c) * HW + p;
      const float v = np[0];
      const float e = v - t;
      acc += fabsf(e) * inv_c;
      float g = sgn(e) * inv_c;
      ssum += sgn(e);
      if (x + 1 < W) { const float q = v - np[1]; acc += fabsf(q) * inv_x; g += sgn(q) * inv_x; }
      if (x > 0) g -= sgn(np[-1] - v) * inv_x;
      if (y + 1 < H) { const float q = v - np[W]; acc += fabsf(q) * inv_y; g += sgn(q) * inv_y; }
      if (y > 0) g -= sgn(np[-W] - v) * inv_y;
      if (g_nm) g_nm[((long)b * C + c) * HW + p] = g;
    }
    if (g_out) {
      // d loss / d t = -ssum * inv_c ; dt/dd = -sg (1 - sg) ; dd/do_c = sign(o_c - i_c) / 3
      const float k = ssum * inv_c * sg * (1.f - sg) * (1.f / 3.f);
      float* gb = g_out + (long)b * 3 * HW + p;
      gb[0] = k * sgn(d0); gb[HW] = k * sgn(d1); gb[2 * HW] = k * sgn(d2);
    }
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void tnsm_loss_finish_kernel(const float* __restrict__ part, int n, float* __restrict__ loss) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) loss[0] = s;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

long cidnet_tnsm_noise_loss_ws_floats(void) { return kBlocks; }

int cidnet_tnsm_noise_loss(const float* noise_map, const float* out_rgb, const float* im, float weight, float* loss, float* g_noise,
                           float* g_out, float* ws, long ws_floats, int B, int C, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(noise_map && out_rgb && im && loss && ws && B > 0 && C > 0 && H > 0 && W > 0);
  if (ws_floats < kBlocks) return CIDNET_ERR_WS;
  const long npx = (long)B * H * W;
  long g = (npx + kThreads - 1) / kThreads;
  const int grid = (int)(g > kBlocks ? kBlocks : g);
  hipLaunchKernelGGL(tnsm_noise_loss_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, noise_map, out_rgb, im, g_noise,
                     g_out, ws, B, C, H, W, weight);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(tnsm_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, grid, loss);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
