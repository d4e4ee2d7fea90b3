// dev tool: which ingredient of the conv3 inner loop costs MFMA issue slots?  Each variant runs the k-group burst of
// conv3_kernel<MT=2, LEFT=1> (144 v_mfma_f32_16x16x4 + 72 v_mfma_f32_4x4x1 on 24 + 8 accumulators) with register operands,
// then adds one ingredient at a time.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_mix tools/mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE bit0: A operands re-read from LDS every group; bit1: replace the 4x4x1 by 16x16x4 (padded tile);
// bit2: 48 selects + 24 moves per group on the B operands; bit3: sched_barrier fences around the burst
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int groups, float seed) {
  __shared__ float As[36 * 9 * 48];
  for (int i = threadIdx.x; i < 36 * 9 * 48; i += 256) As[i] = seed + 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, c = lane & 15, j = lane >> 4;
  f32x4 acc[2][2][4], accl[2][4];
  for (int r = 0; r < 2; ++r)
    for (int e = 0; e < 4; ++e) {
      accl[r][e] = f32x4{0, 0, 0, 0};
      for (int mt = 0; mt < 2; ++mt) acc[r][mt][e] = f32x4{0, 0, 0, 0};
    }
  float win[4][6];
  for (int iy = 0; iy < 4; ++iy)
    for (int i = 0; i < 6; ++i) win[iy][i] = seed * 0.5f + 0.01f * (lane + iy * 6 + i);
  float av[9][2], al[9];
  for (int t = 0; t < 9; ++t) { al[t] = seed + t; for (int mt = 0; mt < 2; ++mt) av[t][mt] = seed - t + mt; }
  for (int g = 0; g < groups; ++g) {
    if (MODE & 4) {
#pragma unroll
      for (int iy = 0; iy < 4; ++iy)
#pragma unroll
        for (int i = 0; i < 6; ++i) win[iy][i] = (lane + g + i) & 1 ? win[iy][i] : win[iy][(i + 1) % 6];
    }
    if (MODE & 1) {
      const int gg = g % 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) av[t][mt] = As[((gg * 4 + j) * 9 + t) * 48 + mt * 16 + c];
        al[t] = As[((gg * 4 + j) * 9 + t) * 48 + 32 + (lane & 3)];
      }
    }
    if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int iy = 0; iy < 4; ++iy)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int dy = iy - r;
        if (dy < 0 || dy > 2) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[r][mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[dy * 3 + dx][mt], win[iy][e + dx], acc[r][mt][e], 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (MODE & 2) accl[r][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(al[dy * 3 + dx], win[iy][e + dx], accl[r][e], 0, 0, 0);
            else accl[r][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(al[dy * 3 + dx], win[iy][e + dx], accl[r][e], 0, 0, 0);
          }
        }
      }
    if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int r = 0; r < 2; ++r)
    for (int e = 0; e < 4; ++e) {
      s += accl[r][e][0] + accl[r][e][1] + accl[r][e][2] + accl[r][e][3];
      for (int mt = 0; mt < 2; ++mt) s += acc[r][mt][e][0] + acc[r][mt][e][1] + acc[r][mt][e][2] + acc[r][mt][e][3];
    }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what) {
  float* out;
  const int blocks = 512, groups = 2000;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, groups, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double big = 144.0 + ((MODE & 2) ? 72.0 : 0.0), small = (MODE & 2) ? 0.0 : 72.0;
  const double cyc = big * 32 + small * 8;                       // MFMA-pipe cycles per group per wave
  const double waves = blocks * 4.0;
  const double ideal_ms = cyc * groups * waves / (1024.0 * 2.4e9) * 1e3;
  printf("%-62s %8.2f ms  MFMA-pipe utilisation %5.1f %%\n", what, ms, 100.0 * ideal_ms / ms);
  hipFree(out);
}

int main() {
  run<0>("registers only, 144 x 16x16x4 + 72 x 4x4x1");
  run<2>("registers only, 216 x 16x16x4 (padded tile)");
  run<1>("+ A operands re-read from LDS per group");
  run<3>("+ LDS, padded tile");
  run<5>("+ LDS + selects/moves on B");
  run<13>("+ LDS + selects + sched_barrier fences");
  run<8>("registers only + fences");
  return 0;
}
