// Shared helpers for the CIDNet gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CIDNET_WAVE 64

// ---- C-ABI status codes (include/cidnet_hip.h) -------------------------------------------
#define CIDNET_OK 0
#define CIDNET_ERR_ARG (-1)     // null pointer / non-positive size
#define CIDNET_ERR_SHAPE (-2)   // shape the kernel family does not support
#define CIDNET_ERR_WS (-3)      // workspace too small

#define CIDNET_CHECK_ARG(cond) \
  do {                         \
    if (!(cond)) return CIDNET_ERR_ARG; \
  } while (0)

// launch + translate the (sticky-free) launch status into the ABI's int
#define CIDNET_LAUNCH_STATUS()                           \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return (int)e__;              \
  } while (0)

namespace cidnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 4 floats with only dword alignment promised: hipcc emits global_load_dwordx4 /
// global_store_dwordx4 for it on gfx950 (unaligned access mode), so rows whose byte offset is
// not a multiple of 16 (e.g. 75- or 150-pixel rows) still move 16 B per lane.
struct __attribute__((packed, aligned(4))) f4u {
  float x, y, z, w;
};

__device__ __forceinline__ f32x4 load4u(const float* p) {
  f4u v = *reinterpret_cast<const f4u*>(p);
  f32x4 r = {v.x, v.y, v.z, v.w};
  return r;
}
__device__ __forceinline__ void store4u(float* p, f32x4 v) {
  f4u s;
  s.x = v[0]; s.y = v[1]; s.z = v[2]; s.w = v[3];
  *reinterpret_cast<f4u*>(p) = s;
}

// ---- bf16 storage (row J1: activations / saved tensors may be stored as bf16; all arithmetic stays fp32) ------------
// Element type codes of the C ABI (include/cidnet_hip.h): CIDNET_F32 = 0, CIDNET_BF16 = 1.  A tensor's type is a
// kernel argument (wave-uniform), so one kernel serves both; the branch is scalar.
typedef unsigned short bf16_t;
struct __attribute__((packed, aligned(2))) h4u { bf16_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) h2u { bf16_t x, y; };

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {          // round to nearest even; v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
// 4 consecutive elements at element offset `off` of a tensor whose type is `dt`
__device__ __forceinline__ f32x4 ld4t(const void* base, long off, int dt) {
  if (dt) {
    const h4u v = *reinterpret_cast<const h4u*>(reinterpret_cast<const bf16_t*>(base) + off);
    return f32x4{bf16_to_f32(v.x), bf16_to_f32(v.y), bf16_to_f32(v.z), bf16_to_f32(v.w)};
  }
  return load4u(reinterpret_cast<const float*>(base) + off);
}
__device__ __forceinline__ void st4t(void* base, long off, int dt, f32x4 v) {
  if (dt) {
    h4u s;
    s.x = f32_to_bf16(v[0]); s.y = f32_to_bf16(v[1]); s.z = f32_to_bf16(v[2]); s.w = f32_to_bf16(v[3]);
    *reinterpret_cast<h4u*>(reinterpret_cast<bf16_t*>(base) + off) = s;
  } else {
    store4u(reinterpret_cast<float*>(base) + off, v);
  }
}
__device__ __forceinline__ float ld1t(const void* base, long off, int dt) {
  return dt ? bf16_to_f32(reinterpret_cast<const bf16_t*>(base)[off]) : reinterpret_cast<const float*>(base)[off];
}
__device__ __forceinline__ void st1t(void* base, long off, int dt, float v) {
  if (dt) reinterpret_cast<bf16_t*>(base)[off] = f32_to_bf16(v);
  else reinterpret_cast<float*>(base)[off] = v;
}

// tanh on the hardware exp2 / rcp units (about 10 VALU ops instead of the library's ~50): 1 - 2/(e^{2x} + 1) away
// from zero, the odd Taylor polynomial through x^7 for |x| < 0.2 where that form cancels.  Absolute error < 1.5e-7.
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);          // e^{2x}; inf / 0 at the extremes
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float x2 = x * x;
  const float small = x * (1.0f + x2 * (-0.333333333f + x2 * (0.133333333f + x2 * -0.053968254f)));
  return fabsf(x) < 0.2f ? small : big;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the block; result valid in thread 0.  `red` = LDS scratch of >= blockDim/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
  }
  return t;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Raises a kernel's dynamic-LDS limit (the 64 KB default is below what conv3x / conv3xw / pwb use).  The attribute is per
// DEVICE, so the "done" flags are indexed by the current device (one static LdsLimit per kernel instantiation); the flag is
// an atomic and setting the attribute twice is harmless, so two host threads may race here.  Returns the HIP status.
struct LdsLimit {
  static constexpr int kMaxDev = 64;
  unsigned char done[kMaxDev] = {};
  hipError_t raise(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < kMaxDev;
    if (cached && __atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && cached) __atomic_store_n(&done[dev], (unsigned char)1, __ATOMIC_RELEASE);
    return e;
  }
};

}  // namespace cidnet
