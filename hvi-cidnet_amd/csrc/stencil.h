// Register-window loaders shared by the 3x3 stencil kernels (dw.hip, conv3_thin.hip).
//
// A lane owns 4 consecutive pixels x0..x0+3 of a row and needs them plus a 1- or 2-pixel halo.  With ceil(W/4)
// lanes per row nearly every 64-lane wave contains a lane that touches the left or right image border, so a
// loader that branches to a per-element path for border lanes makes almost every wave execute both paths.
// The fast loaders below are straight-line code for every lane:
//   * the lane's loads are issued at addresses clamped into the row, and a few selects rebuild the window of a
//     border lane from the shifted data (zero or replicated halo);
//   * the last lane of a row is pulled back to x0 = W-4 so it also owns 4 full pixels; `dup` counts the leading
//     pixels it shares with its left neighbour (both write identical values; reductions skip the duplicates);
//   * rows outside the image are read from the clamped row and zeroed (or kept, for replicate padding) by a select.
// They need W >= 8; narrower images take the generic per-element path.  The choice is a template parameter of the
// kernels (NARROW), not a run-time branch: register allocation is per kernel, and the generic path's footprint
// would otherwise set the occupancy of the fast path too.
#pragma once
#include "common.h"

namespace cidnet {

struct Win6 {      // pixels x0-1 .. x0+4
  float v[6];
};
struct Win8 {      // pixels x0-2 .. x0+5
  float v[8];
};
struct __attribute__((packed, aligned(4))) f2u { float x, y; };

// element access for the two storage types of a plane (fp32, or bf16 in the bf16 storage mode: converted on load / store)
template <class T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ f32x4 ld4(const float* p) { return load4u(p); }
  static __device__ __forceinline__ float2 ld2(const float* p) { const f2u v = *reinterpret_cast<const f2u*>(p); return float2{v.x, v.y}; }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st4(float* p, f32x4 v) { store4u(p, v); }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ f32x4 ld4(const bf16_t* p) { return ld4t(p, 0, 1); }
  static __device__ __forceinline__ float2 ld2(const bf16_t* p) {
    const h2u v = *reinterpret_cast<const h2u*>(p);
    return float2{bf16_to_f32(v.x), bf16_to_f32(v.y)};
  }
  static __device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st4(bf16_t* p, f32x4 v) { st4t(p, 0, 1, v); }
  static __device__ __forceinline__ void st1(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// x origin of lane xl of a row and the number of leading pixels it shares with lane xl-1
template <bool NARROW>
__device__ __forceinline__ int lane_x0(int xl, int W, int& dup) {
  const int x = xl * 4;
  if (NARROW) { dup = 0; return x; }
  const int x0 = min(x, W - 4);
  dup = x - x0;
  return x0;
}

template <class T>
__device__ __forceinline__ Win6 load_win6_generic(const T* __restrict__ plane, int yy, int x0, int H, int W, bool replicate) {
  Win6 r;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = 0.f;
  if (yy < 0 || yy >= H) {
    if (!replicate) return r;
    yy = yy < 0 ? 0 : H - 1;
  }
  const T* row = plane + (long)yy * W;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    int x = x0 - 1 + i;
    if (x < 0) { if (!replicate) continue; x = 0; }
    if (x >= W) { if (!replicate) continue; x = W - 1; }
    r.v[i] = Elem<T>::ld1(row + x);
  }
  return r;
}

template <bool REP, class T>
__device__ __forceinline__ Win6 load_win6_fast(const T* __restrict__ plane, int yy, int x0, int H, int W) {
  const int yc = min(max(yy, 0), H - 1);
  const T* row = plane + (long)yc * W;
  const f32x4 a4 = Elem<T>::ld4(row + max(x0 - 1, 0));          // row[x0-1 .. x0+2], or row[0..3] at the left border
  const float2 b = Elem<T>::ld2(row + min(x0 + 3, W - 2));      // row[x0+3 .. x0+4], or row[W-2 .. W-1] at the right border
  const struct { float x, y, z, w; } a = {a4[0], a4[1], a4[2], a4[3]};
  const bool left = x0 == 0, right = x0 + 4 >= W;
  Win6 r;
  r.v[0] = left ? (REP ? a.x : 0.f) : a.x;
  r.v[1] = left ? a.x : a.y;
  r.v[2] = left ? a.y : a.z;
  r.v[3] = left ? a.z : a.w;
  r.v[4] = right ? b.y : b.x;
  r.v[5] = right ? (REP ? b.y : 0.f) : b.y;
  if (!REP) {
    const bool ok = yy == yc;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.v[i] = ok ? r.v[i] : 0.f;
  }
  return r;
}

template <bool REP, bool NARROW, class T>
__device__ __forceinline__ Win6 load_win6(const T* __restrict__ plane, int yy, int x0, int H, int W) {
  if constexpr (NARROW) return load_win6_generic(plane, yy, x0, H, W, REP);
  else return load_win6_fast<REP>(plane, yy, x0, H, W);
}

template <class T>
__device__ __forceinline__ Win8 load_win8_generic(const T* __restrict__ plane, int yy, int x0, int H, int W) {
  Win8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0.f;
  if (yy < 0 || yy >= H) return r;
  const T* row = plane + (long)yy * W;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int x = x0 - 2 + i;
    if (x >= 0 && x < W) r.v[i] = Elem<T>::ld1(row + x);
  }
  return r;
}

// zero padding only
template <bool NARROW, class T>
__device__ __forceinline__ Win8 load_win8(const T* __restrict__ plane, int yy, int x0, int H, int W) {
  if constexpr (NARROW) return load_win8_generic(plane, yy, x0, H, W);
  const int yc = min(max(yy, 0), H - 1);
  const T* row = plane + (long)yc * W;
  const int xb = min(x0 + 2, W - 4);
  const f32x4 a4 = Elem<T>::ld4(row + max(x0 - 2, 0));          // row[x0-2 .. x0+1], or row[0..3] at the left border
  const f32x4 b4 = Elem<T>::ld4(row + xb);                      // row[x0+2 .. x0+5], pulled back by s at the right border
  const struct { float x, y, z, w; } a = {a4[0], a4[1], a4[2], a4[3]}, b = {b4[0], b4[1], b4[2], b4[3]};
  const bool left = x0 == 0;
  const int s = x0 + 2 - xb;                                                   // 0, 1 (x0 = W-5) or 2 (x0 = W-4)
  Win8 r;
  r.v[0] = left ? 0.f : a.x;
  r.v[1] = left ? 0.f : a.y;
  r.v[2] = left ? a.x : a.z;
  r.v[3] = left ? a.y : a.w;
  r.v[4] = s == 0 ? b.x : (s == 1 ? b.y : b.z);
  r.v[5] = s == 0 ? b.y : (s == 1 ? b.z : b.w);
  r.v[6] = s == 0 ? b.z : (s == 1 ? b.w : 0.f);
  r.v[7] = s == 0 ? b.w : 0.f;
  const bool ok = yy == yc;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = ok ? r.v[i] : 0.f;
  return r;
}

// the lane's own 4 pixels of row y (y inside the image)
template <bool NARROW, class T>
__device__ __forceinline__ f32x4 load_px4(const T* __restrict__ plane, int y, int x0, int W) {
  const T* row = plane + (long)y * W;
  if (!NARROW) return Elem<T>::ld4(row + x0);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (x0 + e < W) v[e] = Elem<T>::ld1(row + x0 + e);
  return v;
}

template <bool NARROW, class T>
__device__ __forceinline__ void store_px4(T* __restrict__ plane, int y, int x0, int W, f32x4 v) {
  T* row = plane + (long)y * W;
  if (!NARROW) {
    Elem<T>::st4(row + x0, v);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (x0 + e < W) Elem<T>::st1(row + x0 + e, v[e]);
  }
}

}  // namespace cidnet
