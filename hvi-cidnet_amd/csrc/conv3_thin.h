// Streaming (non-MFMA) 3x3 convolution paths for layers with <= 4 channels on one side; see conv3_thin.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace cidnet {

bool c3_thin_applies(int M, int K);
int c3_thin_conv(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, int replicate, float* Y, long y_bs,
                 int B, int M, int K, int H, int W, hipStream_t s);
int c3_thin_wgrad_chunks(int H, int W);
// writes per-block partials to slabs[B][chunks][M*N*9]; the caller reduces them
int c3_thin_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int replicate, float* slabs, int B, int M, int N, int H,
                  int W, hipStream_t s);

}  // namespace cidnet
