// Stand-alone probe for the interaction described in DESIGN.md section 4 (i): does a wave that streams MFMAs disturb the
// packed-fp32 FMAs (v_pk_fma_f32 with op_sel) of ANOTHER kernel's waves on the same SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_pk_probe tools/mfma_pk_probe.hip && gpurun_out/mfma_pk_probe
//
// Victim: every lane repeats   acc.lo += x.lo * w.hi ; acc.hi += x.hi * w.lo   (one v_pk_fma_f32 with op_sel:[0,1,0]
// op_sel_hi:[1,0,1], the form LLVM emits for the stem conv) on operands it re-reads from LDS with ds_read_b128 every
// iteration, and compares with the same sum formed by two scalar v_fma_f32; mismatches are counted.
// Aggressors (second stream, launched first so they own the SIMDs when the victim arrives): bf16 MFMA 16x16x32 stream,
// fp32 MFMA 16x16x4 stream, or a plain VALU stream, each with or without a large register footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)

// VAR 0: packed op (op_sel crossing) first, scalar check after;  1: scalar first, packed after;  2: two wait states between the
// LDS wait and the packed op;  3: packed op without op_sel;  4: no packed op at all -- the same register read by two scalar FMAs,
// one right after the LDS wait and one a few instructions later
template <int VAR>
__global__ __launch_bounds__(256) void victim(unsigned long long* bad, int iters) {
  __shared__ f32x4 wts[64];
  if (threadIdx.x < 64) wts[threadIdx.x] = f32x4{0.5f + threadIdx.x, 1.25f + threadIdx.x, -0.75f, 2.0f};
  __syncthreads();
  const float x0 = 1.0f + 0.001f * (threadIdx.x & 63), x1 = 2.0f - 0.002f * (threadIdx.x & 63);
  unsigned long long n = 0;
  for (int it = 0; it < iters; ++it) {
    const f32x4 w = *(volatile f32x4*)&wts[(it + blockIdx.x) & 63];      // ds_read_b128, broadcast address
    f32x2 acc = {0.25f, -0.5f};
    const f32x2 x = {x0 + it * 0.5f, x1};
    const f32x2 wp = {w[0], w[1]};
    float e0, e1;
    if (VAR == 1) { e0 = __builtin_fmaf(x[0], wp[1], 0.25f); e1 = __builtin_fmaf(x[1], wp[0], -0.5f); asm volatile("" : "+v"(e0), "+v"(e1)); }
    if (VAR == 2) asm volatile("s_nop 1");
    if (VAR == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(wp));
    else if (VAR == 4) { float a0 = 0.25f, a1 = -0.5f; asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %4, %5, %1" : "+v"(a0), "+v"(a1) : "v"(x[0]), "v"(wp[1]), "v"(x[1]), "v"(wp[0])); acc = f32x2{a0, a1}; }
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(wp));
    if (VAR == 3) { e0 = __builtin_fmaf(x[0], wp[0], 0.25f); e1 = __builtin_fmaf(x[1], wp[1], -0.5f); }
    else if (VAR != 1) { e0 = __builtin_fmaf(x[0], wp[1], 0.25f); e1 = __builtin_fmaf(x[1], wp[0], -0.5f); }
    n += (acc[0] != e0) + (acc[1] != e1);
  }
  if (n) atomicAdd(bad, n);
}

template <int KIND, int FAT>   // KIND 0: bf16 MFMA, 1: fp32 MFMA, 2: VALU ; FAT: keep ~200 VGPRs live
__global__ __launch_bounds__(256, 2) void aggressor(float* out, int iters) {
  __shared__ u32x4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float fat[FAT ? 192 : 1];
#pragma unroll
  for (int i = 0; i < (FAT ? 192 : 1); ++i) fat[i] = threadIdx.x * 0.001f + i;
  const unsigned seed = threadIdx.x * 2654435761u;
  uint4 au = {seed | 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bu = {0x3f803f80u, seed & 0x00010001u | 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
      else if (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f + j, 0.5f, acc[j], 0, 0, 0);
      else if (KIND == 2) { acc[j] = acc[j] * 1.0001f + f32x4{1.f, 2.f, 3.f, 4.f}; }
      else if (KIND == 3) {                                      // the SDWA form of conv3s.hip's bf16 pair packing
        unsigned r = __float_as_uint(acc[j][0]), q = __float_as_uint(acc[j][1]) + it;
        asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                     "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
                     "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                     "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(r) : "v"(q));
        acc[j][0] = __uint_as_float(r & 0x3fffffffu);
      } else if (KIND == 7) {                                    // two ds_read_b64 per fragment, as conv3x.hip (round 3), feeding bf16 MFMAs
        const char* base = reinterpret_cast<const char*>(lds) + (((threadIdx.x * 9 + it * 64 + j * 256) & 4095) * 8);
        unsigned long long lo, hi;
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"((unsigned)(size_t)base));
        const uint4 f = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, f), acc[j], 0, 0, 0);
      } else {                                                   // LDS fragment reads, as conv3s.hip: feeding bf16 MFMAs (4), VALU (5), fp32 MFMAs (6)
        const u32x4 f = *(volatile u32x4*)&lds[(threadIdx.x * 7 + it * 64 + j * 256) & 2047];
        if (KIND == 4) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, f), acc[j], 0, 0, 0);
        else if (KIND == 5) acc[j] = acc[j] * 1.0001f + f32x4{(float)f[0], (float)f[1], (float)f[2], (float)f[3]};
        else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(f[0]), __uint_as_float(f[1]), acc[j], 0, 0, 0);
      }
    }
    if (FAT) {
#pragma unroll
      for (int i = 0; i < 192; i += 48) fat[i] += acc[0][0];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < (FAT ? 192 : 1); ++i) s += fat[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
}

template <int KIND, int FAT, int VAR>
unsigned long long run(unsigned long long* bad, float* out, hipStream_t sv, hipStream_t sa, bool concurrent) {
  CK(hipMemset(bad, 0, 8));
  CK(hipDeviceSynchronize());
  if (concurrent) hipLaunchKernelGGL((aggressor<KIND, FAT>), dim3(512), dim3(256), 0, sa, out, 40000);
  hipLaunchKernelGGL(victim<VAR>, dim3(2048), dim3(256), 0, sv, bad, 20000);
  CK(hipDeviceSynchronize());
  unsigned long long h = 0;
  CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
  return h;
}

template <int KIND, int FAT>
void row(const char* name, unsigned long long* bad, float* out, hipStream_t sv, hipStream_t sa, bool concurrent = true) {
  printf("%-46s", name);
  printf(" %8llu", run<KIND, FAT, 0>(bad, out, sv, sa, concurrent));
  printf(" %8llu", run<KIND, FAT, 1>(bad, out, sv, sa, concurrent));
  printf(" %8llu", run<KIND, FAT, 2>(bad, out, sv, sa, concurrent));
  printf(" %8llu", run<KIND, FAT, 3>(bad, out, sv, sa, concurrent));
  printf(" %8llu\n", run<KIND, FAT, 4>(bad, out, sv, sa, concurrent));
}

int main() {
  unsigned long long* bad; float* out;
  CK(hipMalloc(&bad, 8)); CK(hipMalloc(&out, 512 * 256 * 4));
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  printf("victim mismatches (of 10.5e9 packed FMAs) by victim variant:   pk first | scalar first | s_nop | no op_sel | scalar only\n");
  for (int rep = 0; rep < 2; ++rep) {
    row<2, 0>("victim alone", bad, out, sv, sa, false);
    row<2, 1>("beside VALU stream, 200-VGPR waves", bad, out, sv, sa);
    row<1, 1>("beside fp32 MFMA stream, 200-VGPR waves", bad, out, sv, sa);
    row<0, 1>("beside bf16 MFMA stream, 200-VGPR waves", bad, out, sv, sa);
    row<3, 1>("beside SDWA stream, 200-VGPR waves", bad, out, sv, sa);
    row<4, 0>("beside LDS b128 reads + bf16 MFMA", bad, out, sv, sa);
    row<4, 1>("beside LDS reads + bf16 MFMA, 200-VGPR waves", bad, out, sv, sa);
    row<7, 0>("beside LDS 2 x b64 reads + bf16 MFMA", bad, out, sv, sa);
    row<7, 1>("beside LDS 2 x b64 reads + bf16 MFMA, 200 VGPRs", bad, out, sv, sa);
    row<5, 1>("beside LDS reads + VALU, 200-VGPR waves", bad, out, sv, sa);
    row<6, 1>("beside LDS reads + fp32 MFMA, 200-VGPR waves", bad, out, sv, sa);
  }
  return 0;
}
