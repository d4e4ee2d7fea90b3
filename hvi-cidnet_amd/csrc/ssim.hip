// "Next" row f1: the SSIM training loss on device.
// Reference: SSIM.forward, loss/losses.py:166-190, and map_ssim / create_window / gaussian, loss/loss_utils.py:113-145:
//   window = outer(g, g), g = normalised 11-tap Gaussian (sigma 1.5); depthwise conv, zero padding 5
//   mu1 = W*x, mu2 = W*y, s1 = W*(x^2) - mu1^2, s2 = W*(y^2) - mu2^2, s12 = W*(xy) - mu1 mu2
//   S = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),  C1 = 1e-4, C2 = 9e-4
//   loss = (1 - mean(S)) * weight
// Forward kernel: a block owns a 16 x 64 tile of one (b,c) plane, stages x and y (+5 halo) in LDS, every thread forms
// the five window sums of its 4 pixels, S, and the three per-pixel derivative maps the backward needs
//   A = dS/dmu1 (total), B = dS/d(W*x^2), C = dS/d(W*xy);   sum(S) goes to per-block partials (fixed order).
// Backward (wrt x only, y is the ground truth): dS_total/dx(q) = (W*A)(q) + 2 x(q) (W*B)(q) + y(q) (W*C)(q), the
// window being symmetric; the same tile scheme filters A, B, C.  Images are 3 channels: the whole loss moves ~0.3 GB.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kTH = 16, kTW = 64, kR = 5, kWin = 11;
constexpr int kLH = kTH + 2 * kR, kLW = kTW + 2 * kR;      // 26 x 74 LDS tile

struct Gauss {
  float g[kWin];
};

inline Gauss make_gauss() {                      // loss_utils.gaussian(11, 1.5), evaluated like the reference (float32 tensor ops)
  Gauss w;
  float s = 0.f;
  for (int i = 0; i < kWin; ++i) {
    const double d = (double)(i - kWin / 2);
    w.g[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
    s += w.g[i];
  }
  for (int i = 0; i < kWin; ++i) w.g[i] = w.g[i] / s;
  return w;
}

__device__ __forceinline__ void load_tile(const float* __restrict__ plane, float* __restrict__ tile, int ty0, int tx0, int H, int W) {
  for (int i = threadIdx.x; i < kLH * kLW; i += kThreads) {
    const int r = i / kLW, c = i - r * kLW;
    const int y = ty0 - kR + r, x = tx0 - kR + c;
    tile[i] = (y >= 0 && y < H && x >= 0 && x < W) ? plane[(long)y * W + x] : 0.f;
  }
}

__global__ __launch_bounds__(kThreads) void ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, Gauss gw,
                                                            float* __restrict__ dA, float* __restrict__ dB, float* __restrict__ dC,
                                                            float* __restrict__ part, int H, int W) {
  __shared__ float tx[kLH * kLW], ty[kLH * kLW];
  __shared__ float red[kThreads / 64];
  const long plane = blockIdx.z;
  const int ty0 = blockIdx.y * kTH, tx0 = blockIdx.x * kTW;
  const float* p1 = img1 + plane * (long)H * W;
  const float* p2 = img2 + plane * (long)H * W;
  load_tile(p1, tx, ty0, tx0, H, W);
  load_tile(p2, ty, ty0, tx0, H, W);
  __syncthreads();
  const int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  float m1[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0}, e11[4] = {0, 0, 0, 0}, e22[4] = {0, 0, 0, 0}, e12[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int dy = 0; dy < kWin; ++dy) {
    const float gy = gw.g[dy];
    const float* rx = tx + (r + dy) * kLW + c4;
    const float* ry = ty + (r + dy) * kLW + c4;
    float xv[kWin + 3], yv[kWin + 3];
#pragma unroll
    for (int i = 0; i < kWin + 3; ++i) { xv[i] = rx[i]; yv[i] = ry[i]; }
#pragma unroll
    for (int dx = 0; dx < kWin; ++dx) {
      const float wgt = gy * gw.g[dx];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = xv[dx + e], y = yv[dx + e];
        m1[e] += wgt * x; m2[e] += wgt * y;
        e11[e] += wgt * (x * x); e22[e] += wgt * (y * y); e12[e] += wgt * (x * y);
      }
    }
  }
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float ssum = 0.f;
  const int y = ty0 + r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int x = tx0 + c4 + e;
    if (y >= H || x >= W) continue;
    const float s1 = e11[e] - m1[e] * m1[e], s2 = e22[e] - m2[e] * m2[e], s12 = e12[e] - m1[e] * m2[e];
    const float N1 = 2.f * m1[e] * m2[e] + C1, N2 = 2.f * s12 + C2;
    const float D1 = m1[e] * m1[e] + m2[e] * m2[e] + C1, D2 = s1 + s2 + C2;
    const float inv = 1.f / (D1 * D2);
    const float S = N1 * N2 * inv;
    ssum += S;
    const long o = plane * (long)H * W + (long)y * W + x;
    // dS/dmu1 with the raw moments held fixed: N1, N2 (through s12), D1, D2 (through s1) all depend on mu1
    dA[o] = (2.f * m2[e] * N2 - 2.f * m2[e] * N1) * inv - S * 2.f * m1[e] / D1 + S * 2.f * m1[e] / D2;
    dB[o] = -S / D2;
    dC[o] = 2.f * N1 * inv;
  }
  const float bs = block_sum(ssum, red);
  if (threadIdx.x == 0) part[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = bs;
}

__global__ void ssim_finish_kernel(const float* __restrict__ part, long n_part, float inv_n, float weight, float* __restrict__ loss) {
  __shared__ float red[4];
  float a = 0.f;
  for (long i = threadIdx.x; i < n_part; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) loss[0] = (1.f - s * inv_n) * weight;
}

__global__ __launch_bounds__(kThreads) void ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                            const float* __restrict__ dA, const float* __restrict__ dB,
                                                            const float* __restrict__ dC, const float* __restrict__ gloss, Gauss gw,
                                                            float scale, float* __restrict__ gimg1, int H, int W) {
  __shared__ float ta[kLH * kLW], tb[kLH * kLW], tc[kLH * kLW];
  const long plane = blockIdx.z;
  const int ty0 = blockIdx.y * kTH, tx0 = blockIdx.x * kTW;
  const long po = plane * (long)H * W;
  load_tile(dA + po, ta, ty0, tx0, H, W);
  load_tile(dB + po, tb, ty0, tx0, H, W);
  load_tile(dC + po, tc, ty0, tx0, H, W);
  __syncthreads();
  const int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  float fa[4] = {0, 0, 0, 0}, fb[4] = {0, 0, 0, 0}, fc[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int dy = 0; dy < kWin; ++dy) {
    const float gy = gw.g[dy];
    const float* ra = ta + (r + dy) * kLW + c4;
    const float* rb = tb + (r + dy) * kLW + c4;
    const float* rc = tc + (r + dy) * kLW + c4;
#pragma unroll
    for (int dx = 0; dx < kWin; ++dx) {
      const float wgt = gy * gw.g[dx];
#pragma unroll
      for (int e = 0; e < 4; ++e) { fa[e] += wgt * ra[dx + e]; fb[e] += wgt * rb[dx + e]; fc[e] += wgt * rc[dx + e]; }
    }
  }
  const float gs = scale * gloss[0];             // -weight / n * d(total)/d(loss)
  const int y = ty0 + r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int x = tx0 + c4 + e;
    if (y >= H || x >= W) continue;
    const long o = po + (long)y * W + x;
    gimg1[o] = gs * (fa[e] + 2.f * img1[o] * fb[e] + img2[o] * fc[e]);
  }
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

long cidnet_ssim_ws_floats(int B, int C, int H, int W) {
  return (long)B * C * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
}

int cidnet_ssim_fwd(const float* img1, const float* img2, float weight, float* loss, float* dA, float* dB, float* dC, float* ws,
                    long ws_floats, int B, int C, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(img1 && img2 && loss && dA && dB && dC && ws && B > 0 && C > 0 && H > 0 && W > 0);
  const long n_part = cidnet_ssim_ws_floats(B, C, H, W);
  if (ws_floats < n_part) return CIDNET_ERR_WS;
  const dim3 grid((unsigned)((W + kTW - 1) / kTW), (unsigned)((H + kTH - 1) / kTH), (unsigned)(B * C));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(kThreads), 0, s, img1, img2, make_gauss(), dA, dB, dC, ws, H, W);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(256), 0, s, ws, n_part, 1.0f / (float)((long)B * C * H * W), weight, loss);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_ssim_bwd(const float* img1, const float* img2, const float* dA, const float* dB, const float* dC, const float* gloss,
                    float weight, float* gimg1, int B, int C, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(img1 && img2 && dA && dB && dC && gloss && gimg1 && B > 0 && C > 0 && H > 0 && W > 0);
  const dim3 grid((unsigned)((W + kTW - 1) / kTW), (unsigned)((H + kTH - 1) / kTH), (unsigned)(B * C));
  const float scale = -weight / (float)((long)B * C * H * W);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, img1, img2, dA, dB, dC, gloss, make_gauss(), scale,
                     gimg1, H, W);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
