// dev tool: achievable v_mfma_f32_16x16x4_f32 rate for the operand patterns the conv kernels use
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip && gpurun -- ./tools/bin/mfma_peak
// (MI355X: 155 TFLOP/s with >= 16 independent accumulators per wave, i.e. 98.7 % of the 157.3 nominal)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, float seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; b[i] = seed * 0.5f + threadIdx.x * 0.002f - i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int WAVES>
void run(int blocks_per_cu) {
  float* out;
  const int blocks = 256 * blocks_per_cu;
  hipMalloc(&out, blocks * WAVES * 64 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, WAVES><<<blocks, WAVES * 64>>>(out, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, WAVES><<<blocks, WAVES * 64>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * blocks * WAVES;
  printf("NACC=%2d waves/block=%d blocks/CU=%d: %.1f TFLOP/s\n", NACC, WAVES, blocks_per_cu, flop / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<4, 4>(1); run<16, 4>(1); run<16, 4>(2); run<48, 4>(1); run<48, 4>(2); run<2, 4>(2); run<1, 4>(2); run<16, 8>(1);
  return 0;
}
