// K14: the pieces of the TNSM variant (net/TNSM.py) that the CAB / IEL kernels do not cover:
//   global_pool      : adaptive avg + max pool to 1x1 per (b,c) plane (DynamicNoiseMap :39-40)
//   noise_global     : the two tiny FCs + sigmoid of DynamicNoiseMap (:43-47), folded with
//                      noise_branch[2] and final_conv into ONE per-sample row vector v_b, so that
//                      noise_map = sigmoid(v_b . leaky(dw3x3(x)))                     (:50-54)
//   leaky            : LeakyReLU(0.2) elementwise, forward / backward
//   rowdot_sigmoid   : noise_map[b][p] = sigmoid(sum_c v[b][c] * t[b][c][p])
//   modulate         : v' = v * sigmoid(ws[c] * noise_map)        (NoiseAwareAttention :103-112)
//   blend            : out = nm * a + (1 - nm) * d                (AdaptiveFilter :165-166, after the
//                      fusion 1x1 conv has been distributed over the concat)
//   resize_bilinear  : F.interpolate(bilinear, align_corners=False) forward / adjoint (CIDNet_TNSM.py:258)
//   sigmoid          : elementwise forward / backward
// All are HBM-bound streaming kernels (4 pixels per lane) or tiny per-sample kernels.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr float kLeaky = 0.2f;

__device__ __forceinline__ f32x4 ld4(const float* row, long p, int n) {
  if (n == 4) return load4u(row + p);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < n; ++e) v[e] = row[p + e];
  return v;
}
__device__ __forceinline__ void st4(float* row, long p, int n, f32x4 v) {
  if (n == 4) { store4u(row + p, v); return; }
  for (int e = 0; e < n; ++e) row[p + e] = v[e];
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

inline int grid_for(long items, int cap = 4096) {
  long g = (items + kThreads - 1) / kThreads;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

// ---- global average / max pooling: one block per (b,c) plane -----------------------------------
__global__ __launch_bounds__(kThreads) void global_pool_kernel(const float* __restrict__ x, float* __restrict__ avg,
                                                               float* __restrict__ mx, int* __restrict__ amax, long HW) {
  __shared__ float red[kThreads / 64];
  __shared__ float smax[kThreads];
  __shared__ int sidx[kThreads];
  const float* pl = x + (long)blockIdx.x * HW;
  float s = 0.f, m = -INFINITY;
  int mi = 0;
  for (long p = threadIdx.x; p < HW; p += blockDim.x) {
    const float v = pl[p];
    s += v;
    if (v > m) { m = v; mi = (int)p; }
  }
  smax[threadIdx.x] = m; sidx[threadIdx.x] = mi;
  const float tot = block_sum(s, red);
  __syncthreads();
  if (threadIdx.x == 0) {
    float bm = smax[0];
    int bi = sidx[0];
    for (int t = 1; t < kThreads; ++t)                   // first occurrence wins ties, as ATen's max pool
      if (smax[t] > bm || (smax[t] == bm && sidx[t] < bi)) { bm = smax[t]; bi = sidx[t]; }
    avg[blockIdx.x] = tot / (float)HW;
    mx[blockIdx.x] = bm;
    amax[blockIdx.x] = bi;
  }
}

// gx[b][c][p] = gavg/HW + (p == amax) * gmx
__global__ __launch_bounds__(kThreads) void global_pool_bwd_kernel(const float* __restrict__ gavg, const float* __restrict__ gmx,
                                                                   const int* __restrict__ amax, float* __restrict__ gx,
                                                                   long planes, long HW) {
  const long total = planes * HW;
  const float inv = 1.0f / (float)HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pl = i / HW, p = i - pl * HW;
    gx[i] = gavg[pl] * inv + (p == amax[pl] ? gmx[pl] : 0.f);
  }
}

// ---- DynamicNoiseMap's global branch, one block per sample ---------------------------------------
// h1 = relu(W1 avg), h2 = relu(W1 mx); a = W2 (h1 + h2)  [fc2 is linear, bias-free]; gf = sigmoid(a)
// vrow[k] = sum_c wf[c] * gf[c] * Wn[c][k]
__global__ __launch_bounds__(kThreads) void noise_global_fwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                                    const float* __restrict__ W1, const float* __restrict__ W2,
                                                                    const float* __restrict__ Wn, const float* __restrict__ wf,
                                                                    float* __restrict__ hsum, float* __restrict__ gf,
                                                                    float* __restrict__ vrow, int C, int R) {
  extern __shared__ float sm[];          // hs[R], g[C]
  float* hs = sm;
  float* g = sm + R;
  const int b = blockIdx.x;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    float a = 0.f, m = 0.f;
    for (int c = 0; c < C; ++c) { a += W1[r * C + c] * avg[b * C + c]; m += W1[r * C + c] * mx[b * C + c]; }
    const float h = fmaxf(a, 0.f) + fmaxf(m, 0.f);
    hs[r] = h;
    // keep the two relu masks for the backward in the sign bits of two stored values
    hsum[((long)b * R + r) * 3 + 0] = h;
    hsum[((long)b * R + r) * 3 + 1] = a > 0.f ? 1.f : 0.f;
    hsum[((long)b * R + r) * 3 + 2] = m > 0.f ? 1.f : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += W2[c * R + r] * hs[r];
    const float s = sigmoidf(a);
    g[c] = s;
    gf[b * C + c] = s;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < C; k += blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += wf[c] * g[c] * Wn[c * C + k];
    vrow[b * C + k] = a;
  }
}

// per-sample partial gradients of W1 (R,C), W2 (C,R), Wn (C,C), wf (C) and gavg, gmx (C)
__global__ __launch_bounds__(kThreads) void noise_global_bwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                                    const float* __restrict__ W1, const float* __restrict__ W2,
                                                                    const float* __restrict__ Wn, const float* __restrict__ wf,
                                                                    const float* __restrict__ hsum, const float* __restrict__ gf,
                                                                    const float* __restrict__ gvrow, float* __restrict__ gW1_b,
                                                                    float* __restrict__ gW2_b, float* __restrict__ gWn_b,
                                                                    float* __restrict__ gwf_b, float* __restrict__ gavg,
                                                                    float* __restrict__ gmx, int C, int R) {
  extern __shared__ float sm[];          // ga[C] (grad at pre-sigmoid a), gh[R]
  float* ga = sm;
  float* gh = sm + C;
  const int b = blockIdx.x;
  const float* gv = gvrow + (long)b * C;
  // vrow[k] = sum_c wf[c] gf[c] Wn[c][k]
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float dot = 0.f;
    for (int k = 0; k < C; ++k) {
      dot += gv[k] * Wn[c * C + k];
      gWn_b[((long)b * C + c) * C + k] = gv[k] * wf[c] * gf[b * C + c];
    }
    gwf_b[(long)b * C + c] = dot * gf[b * C + c];
    const float g_gf = dot * wf[c];
    const float s = gf[b * C + c];
    ga[c] = g_gf * s * (1.f - s);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    float t = 0.f;
    for (int c = 0; c < C; ++c) t += ga[c] * W2[c * R + r];
    gh[r] = t;
  }
  for (int i = threadIdx.x; i < C * R; i += blockDim.x) {
    const int c = i / R, r = i - c * R;
    gW2_b[(long)b * C * R + i] = ga[c] * hsum[((long)b * R + r) * 3];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R * C; i += blockDim.x) {
    const int r = i / C, c = i - r * C;
    const float m1 = hsum[((long)b * R + r) * 3 + 1], m2 = hsum[((long)b * R + r) * 3 + 2];
    gW1_b[(long)b * R * C + i] = gh[r] * (m1 * avg[b * C + c] + m2 * mx[b * C + c]);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, m = 0.f;
    for (int r = 0; r < R; ++r) {
      a += gh[r] * hsum[((long)b * R + r) * 3 + 1] * W1[r * C + c];
      m += gh[r] * hsum[((long)b * R + r) * 3 + 2] * W1[r * C + c];
    }
    gavg[b * C + c] = a;
    gmx[b * C + c] = m;
  }
}

// ---- elementwise: MODE 0 leaky fwd, 1 leaky bwd (g, y -> gx), 2 sigmoid fwd, 3 sigmoid bwd (g, y -> gx)
template <int MODE>
__global__ __launch_bounds__(kThreads) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i <= n4; i += (long)gridDim.x * blockDim.x) {
    const int cnt = i < n4 ? 4 : (int)(n & 3);
    if (cnt == 0) continue;
    const f32x4 u = ld4(a, 4 * i, cnt);
    f32x4 v = {0.f, 0.f, 0.f, 0.f}, o;
    if (MODE == 1 || MODE == 3) v = ld4(b, 4 * i, cnt);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (MODE == 0) o[e] = u[e] > 0.f ? u[e] : kLeaky * u[e];
      if (MODE == 1) o[e] = v[e] > 0.f ? u[e] : kLeaky * u[e];
      if (MODE == 2) o[e] = sigmoidf(u[e]);
      if (MODE == 3) o[e] = u[e] * v[e] * (1.f - v[e]);
    }
    st4(y, 4 * i, cnt, o);
  }
}

// ---- noise_map[b][p] = sigmoid(sum_c v[b][c] t[b][c][p]) ---------------------------------------------
__global__ __launch_bounds__(kThreads) void rowdot_sigmoid_kernel(const float* __restrict__ t, const float* __restrict__ v,
                                                                  float* __restrict__ nm, int B, int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const float* tb = t + b * C * HW;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) s += ld4(tb + (long)c * HW, p, n) * v[b * C + c];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = sigmoidf(s[e]);
    st4(nm + b * HW, p, n, o);
  }
}

// gt[b][c][p] = v[b][c] * gl[p],  gl = gnm * nm * (1 - nm)
__global__ __launch_bounds__(kThreads) void rowdot_sigmoid_bwd_t_kernel(const float* __restrict__ gnm, const float* __restrict__ nm,
                                                                        const float* __restrict__ v, float* __restrict__ gt, int B,
                                                                        int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long b = it / nq, p = (it - b * nq) << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const f32x4 s = ld4(nm + b * HW, p, n);
    const f32x4 gl = ld4(gnm + b * HW, p, n) * s * (1.f - s);
    float* gb = gt + b * C * HW;
    for (int c = 0; c < C; ++c) st4(gb + (long)c * HW, p, n, gl * v[b * C + c]);
  }
}

// gv[b][c] = sum_p gl[p] * t[b][c][p]   -- one block per (b,c)
__global__ __launch_bounds__(kThreads) void rowdot_sigmoid_bwd_v_kernel(const float* __restrict__ gnm, const float* __restrict__ nm,
                                                                        const float* __restrict__ t, float* __restrict__ gv, int C,
                                                                        long HW) {
  __shared__ float red[kThreads / 64];
  const long bc = blockIdx.x, b = bc / C;
  const float* tp = t + bc * HW;
  float a = 0.f;
  for (long p = threadIdx.x; p < HW; p += blockDim.x) {
    const float s = nm[b * HW + p];
    a += gnm[b * HW + p] * s * (1.f - s) * tp[p];
  }
  const float r = block_sum(a, red);
  if (threadIdx.x == 0) gv[bc] = r;
}

// ---- v' = v * sigmoid(ws[c] * nm) ------------------------------------------------------------------
// forward: vin at vin + b*vin_bs (channel slice of qkv), vout contiguous (B,C,HW)
__global__ __launch_bounds__(kThreads) void modulate_fwd_kernel(const float* __restrict__ vin, long vin_bs,
                                                                const float* __restrict__ nm, const float* __restrict__ ws,
                                                                float* __restrict__ vout, int B, int C, long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * C * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long q = it % nq, bc = it / nq, b = bc / C;
    const int c = (int)(bc - b * C);
    const long p = q << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const f32x4 s = ld4(nm + b * HW, p, n), v = ld4(vin + b * vin_bs + (long)c * HW, p, n);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * sigmoidf(ws[c] * s[e]);
    st4(vout + bc * HW, p, n, o);
  }
}

// backward: gvin = gvout * keep;  gk = gvout * v * keep * (1 - keep);  gnm[b][p] = sum_c gk * ws[c];
// gws[c] partial per (b, pixel chunk) = sum_p gk * nm.
// A block is 32 pixel quads x 8 channel lanes (thread = quad + 32 * channel lane; channel lane cl walks channels cl, cl + 8, ...):
// at the 50x75 level a lane per quad looping over all 144 channels, with a block reduction (two barriers) per channel, left
// 64 blocks on the chip for 306 us.  The per-channel sum over the block's 32 quads is a 5-step shuffle inside a half wave (no
// LDS, no barrier); the sum over the 8 channel lanes for gnm goes through LDS once, in fixed order.
constexpr int kModQ = 32, kModCL = kThreads / kModQ;
__global__ __launch_bounds__(kThreads) void modulate_bwd_kernel(const float* __restrict__ vin, long vin_bs,
                                                                const float* __restrict__ nm, const float* __restrict__ ws,
                                                                const float* __restrict__ gvout, float* __restrict__ gvin,
                                                                long gvin_bs, float* __restrict__ gnm, float* __restrict__ gws_part,
                                                                int B, int C, long HW) {
  __shared__ f32x4 red[kModCL][kModQ];
  const int b = blockIdx.y;
  const int ql = threadIdx.x & (kModQ - 1), cl = threadIdx.x / kModQ;
  const long nq = (HW + 3) >> 2;
  const long q = (long)blockIdx.x * kModQ + ql;
  const bool live = q < nq;
  const long p = q << 2;
  const int n = live ? ((HW - p >= 4) ? 4 : (int)(HW - p)) : 0;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, gn = s;
  if (live) s = ld4(nm + b * HW, p, n);
  for (int c = cl; c < C; c += kModCL) {
    float part = 0.f;
    if (live) {
      const f32x4 v = ld4(vin + b * vin_bs + (long)c * HW, p, n);
      const f32x4 g = ld4(gvout + ((long)b * C + c) * HW, p, n);
      const float wc = ws[c];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float k = sigmoidf(wc * s[e]);
        o[e] = g[e] * k;
        const float gk = g[e] * v[e] * k * (1.f - k);
        gn[e] += gk * wc;
        if (e < n) part += gk * s[e];
      }
      st4(gvin + b * gvin_bs + (long)c * HW, p, n, o);
    }
#pragma unroll
    for (int m = 1; m < kModQ; m <<= 1) part += __shfl_xor(part, m, 64);      // the 32 quads of this channel lane: one half wave
    if (ql == 0) gws_part[((long)b * gridDim.x + blockIdx.x) * C + c] = part;
  }
  red[cl][ql] = gn;
  __syncthreads();
  if (cl == 0 && live) {
    f32x4 t = red[0][ql];
#pragma unroll
    for (int i = 1; i < kModCL; ++i) t += red[i][ql];
    st4(gnm + b * HW, p, n, t);
  }
}

// ---- out = nm * a + (1 - nm) * d ;  backward: ga = g*nm, gd = g*(1-nm), gnm = sum_c g*(a - d) -------
__global__ __launch_bounds__(kThreads) void blend_fwd_kernel(const float* __restrict__ a, const float* __restrict__ d,
                                                             const float* __restrict__ nm, float* __restrict__ out, int B, int C,
                                                             long HW) {
  const long nq = (HW + 3) >> 2;
  const long total = (long)B * C * nq;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const long q = it % nq, bc = it / nq, b = bc / C;
    const long p = q << 2;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    const f32x4 s = ld4(nm + b * HW, p, n);
    st4(out + bc * HW, p, n, s * ld4(a + bc * HW, p, n) + (1.f - s) * ld4(d + bc * HW, p, n));
  }
}

// block = 32 pixel quads x 8 channel lanes, as modulate_bwd_kernel (a lane per quad looping over 144 channels took 353 us at 50x75)
__global__ __launch_bounds__(kThreads) void blend_bwd_kernel(const float* __restrict__ a, const float* __restrict__ d,
                                                             const float* __restrict__ nm, const float* __restrict__ g,
                                                             float* __restrict__ ga, float* __restrict__ gd, float* __restrict__ gnm,
                                                             int B, int C, long HW) {
  __shared__ f32x4 red[kModCL][kModQ];
  const int b = blockIdx.y;
  const int ql = threadIdx.x & (kModQ - 1), cl = threadIdx.x / kModQ;
  const long nq = (HW + 3) >> 2;
  const long q = (long)blockIdx.x * kModQ + ql;
  const bool live = q < nq;
  const long p = q << 2;
  const int n = live ? ((HW - p >= 4) ? 4 : (int)(HW - p)) : 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const f32x4 s = ld4(nm + b * HW, p, n);
    for (int c = cl; c < C; c += kModCL) {
      const long o = ((long)b * C + c) * HW;
      const f32x4 gg = ld4(g + o, p, n);
      acc += gg * (ld4(a + o, p, n) - ld4(d + o, p, n));
      st4(ga + o, p, n, gg * s);
      st4(gd + o, p, n, gg * (1.f - s));
    }
  }
  red[cl][ql] = acc;
  __syncthreads();
  if (cl == 0 && live) {
    f32x4 t = red[0][ql];
#pragma unroll
    for (int i = 1; i < kModCL; ++i) t += red[i][ql];
    st4(gnm + b * HW, p, n, t);
  }
}

// ---- bilinear resize, align_corners=False (ATen area_pixel_compute_source_index) ---------------------
struct TapF {
  int i0, i1;
  float l1;
};
__device__ __forceinline__ TapF src_tap_f(int o, float scale, int in) {
  float f = scale * ((float)o + 0.5f) - 0.5f;
  if (f < 0.f) f = 0.f;
  TapF t;
  t.i0 = (int)f;
  if (t.i0 > in - 1) t.i0 = in - 1;
  t.l1 = fminf(fmaxf(f - (float)t.i0, 0.f), 1.f);
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  return t;
}

// dst planes may be a channel slice of a wider tensor: plane pl of sample b at dst + b*dst_bs + c*Ho*Wo
__global__ __launch_bounds__(kThreads) void resize_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, long dst_bs,
                                                              int B, int C, int Hi, int Wi, int Ho, int Wo) {
  const long total = (long)B * C * Ho * Wo;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wo);
    const int y = (int)((i / Wo) % Ho);
    const long bc = i / ((long)Wo * Ho), b = bc / C;
    const int c = (int)(bc - b * C);
    const TapF ty = src_tap_f(y, sh, Hi), tx = src_tap_f(x, sw, Wi);
    const float* p = src + bc * (long)Hi * Wi;
    const float top = (1.f - tx.l1) * p[(long)ty.i0 * Wi + tx.i0] + tx.l1 * p[(long)ty.i0 * Wi + tx.i1];
    const float bot = (1.f - tx.l1) * p[(long)ty.i1 * Wi + tx.i0] + tx.l1 * p[(long)ty.i1 * Wi + tx.i1];
    dst[b * dst_bs + ((long)c * Ho + y) * Wo + x] = (1.f - ty.l1) * top + ty.l1 * bot;
  }
}

// adjoint (gather): gsrc[b][c][yi][xi] = sum over outputs whose footprint contains (yi, xi)
__global__ __launch_bounds__(kThreads) void resize_bwd_kernel(const float* __restrict__ gdst, long gdst_bs, float* __restrict__ gsrc,
                                                              int B, int C, int Hi, int Wi, int Ho, int Wo) {
  const long total = (long)B * C * Hi * Wi;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const float ish = 1.f / sh, isw = 1.f / sw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xi = (int)(i % Wi);
    const int yi = (int)((i / Wi) % Hi);
    const long bc = i / ((long)Wi * Hi), b = bc / C;
    const int c = (int)(bc - b * C);
    int ylo = (int)floorf(((float)yi - 1.f + 0.5f) * ish - 0.5f) - 1, yhi = (int)ceilf(((float)yi + 1.f + 0.5f) * ish - 0.5f) + 1;
    int xlo = (int)floorf(((float)xi - 1.f + 0.5f) * isw - 0.5f) - 1, xhi = (int)ceilf(((float)xi + 1.f + 0.5f) * isw - 0.5f) + 1;
    if (yi == 0) ylo = 0;
    if (xi == 0) xlo = 0;
    ylo = max(ylo, 0); xlo = max(xlo, 0); yhi = min(yhi, Ho - 1); xhi = min(xhi, Wo - 1);
    const float* g = gdst + b * gdst_bs + (long)c * Ho * Wo;
    float s = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
      const TapF ty = src_tap_f(yo, sh, Hi);
      float wy = 0.f;
      if (ty.i0 == yi) wy += 1.f - ty.l1;
      if (ty.i1 == yi) wy += ty.l1;
      if (wy == 0.f) continue;
      float rs = 0.f;
      for (int xo = xlo; xo <= xhi; ++xo) {
        const TapF tx = src_tap_f(xo, sw, Wi);
        float wx = 0.f;
        if (tx.i0 == xi) wx += 1.f - tx.l1;
        if (tx.i1 == xi) wx += tx.l1;
        if (wx != 0.f) rs += wx * g[(long)yo * Wo + xo];
      }
      s += wy * rs;
    }
    gsrc[i] = s;
  }
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_global_pool_fwd(const float* x, float* avg, float* mx, int* amax, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(x && avg && mx && amax && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL(global_pool_kernel, dim3((unsigned)(B * C)), dim3(kThreads), 0, (hipStream_t)stream, x, avg, mx, amax, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_global_pool_bwd(const float* gavg, const float* gmx, const int* amax, float* gx, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(gavg && gmx && amax && gx && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL(global_pool_bwd_kernel, dim3(grid_for((long)B * C * HW)), dim3(kThreads), 0, (hipStream_t)stream, gavg, gmx,
                     amax, gx, (long)B * C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_noise_global_fwd(const float* avg, const float* mx, const float* W1, const float* W2, const float* Wn, const float* wf,
                            float* hsum, float* gf, float* vrow, int B, int C, int R, void* stream) {
  CIDNET_CHECK_ARG(avg && mx && W1 && W2 && Wn && wf && hsum && gf && vrow && B > 0 && C > 0 && R > 0);
  hipLaunchKernelGGL(noise_global_fwd_kernel, dim3((unsigned)B), dim3(kThreads), (size_t)(R + C) * sizeof(float),
                     (hipStream_t)stream, avg, mx, W1, W2, Wn, wf, hsum, gf, vrow, C, R);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_noise_global_bwd(const float* avg, const float* mx, const float* W1, const float* W2, const float* Wn, const float* wf,
                            const float* hsum, const float* gf, const float* gvrow, float* gW1_b, float* gW2_b, float* gWn_b,
                            float* gwf_b, float* gavg, float* gmx, int B, int C, int R, void* stream) {
  CIDNET_CHECK_ARG(avg && mx && W1 && W2 && Wn && wf && hsum && gf && gvrow && gW1_b && gW2_b && gWn_b && gwf_b && gavg && gmx);
  CIDNET_CHECK_ARG(B > 0 && C > 0 && R > 0);
  hipLaunchKernelGGL(noise_global_bwd_kernel, dim3((unsigned)B), dim3(kThreads), (size_t)(R + C) * sizeof(float),
                     (hipStream_t)stream, avg, mx, W1, W2, Wn, wf, hsum, gf, gvrow, gW1_b, gW2_b, gWn_b, gwf_b, gavg, gmx, C, R);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

/* mode: 0 leaky_relu(0.2) fwd (a -> y); 1 its bwd (a = g, b = y_fwd -> gx); 2 sigmoid fwd; 3 sigmoid bwd (a = g, b = y_fwd) */
int cidnet_elementwise(int mode, const float* a, const float* b, float* y, long n, void* stream) {
  CIDNET_CHECK_ARG(a && y && n > 0 && mode >= 0 && mode <= 3);
  CIDNET_CHECK_ARG(!(mode & 1) || b);
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for(n / 4 + 1);
  switch (mode) {
    case 0: hipLaunchKernelGGL((ew_kernel<0>), dim3(grid), dim3(kThreads), 0, s, a, b, y, n); break;
    case 1: hipLaunchKernelGGL((ew_kernel<1>), dim3(grid), dim3(kThreads), 0, s, a, b, y, n); break;
    case 2: hipLaunchKernelGGL((ew_kernel<2>), dim3(grid), dim3(kThreads), 0, s, a, b, y, n); break;
    default: hipLaunchKernelGGL((ew_kernel<3>), dim3(grid), dim3(kThreads), 0, s, a, b, y, n); break;
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_rowdot_sigmoid_fwd(const float* t, const float* v, float* nm, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(t && v && nm && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3(grid_for((long)B * ((HW + 3) / 4))), dim3(kThreads), 0, (hipStream_t)stream, t, v,
                     nm, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_rowdot_sigmoid_bwd(const float* gnm, const float* nm, const float* t, const float* v, float* gt, float* gv, int B, int C,
                              long HW, void* stream) {
  CIDNET_CHECK_ARG(gnm && nm && t && v && gt && gv && B > 0 && C > 0 && HW > 0);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rowdot_sigmoid_bwd_t_kernel, dim3(grid_for((long)B * ((HW + 3) / 4))), dim3(kThreads), 0, s, gnm, nm, v, gt, B,
                     C, HW);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(rowdot_sigmoid_bwd_v_kernel, dim3((unsigned)(B * C)), dim3(kThreads), 0, s, gnm, nm, t, gv, C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_modulate_fwd(const float* vin, long vin_bs, const float* nm, const float* ws, float* vout, int B, int C, long HW,
                        void* stream) {
  CIDNET_CHECK_ARG(vin && nm && ws && vout && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL(modulate_fwd_kernel, dim3(grid_for((long)B * C * ((HW + 3) / 4))), dim3(kThreads), 0, (hipStream_t)stream, vin,
                     vin_bs, nm, ws, vout, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_modulate_bwd_ws_floats(int B, int C, long HW) { return (long)B * (((HW + 3) / 4 + kModQ - 1) / kModQ) * C; }

/* gws_part: (B * nchunk, C) partials, nchunk = ceil(ceil(HW/4)/32); the caller sums rows (cidnet_sum_rows) */
int cidnet_modulate_bwd(const float* vin, long vin_bs, const float* nm, const float* ws, const float* gvout, float* gvin,
                        long gvin_bs, float* gnm, float* gws_part, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(vin && nm && ws && gvout && gvin && gnm && gws_part && B > 0 && C > 0 && HW > 0);
  const unsigned nchunk = (unsigned)(((HW + 3) / 4 + kModQ - 1) / kModQ);
  hipLaunchKernelGGL(modulate_bwd_kernel, dim3(nchunk, (unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, vin, vin_bs, nm, ws,
                     gvout, gvin, gvin_bs, gnm, gws_part, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_blend_fwd(const float* a, const float* d, const float* nm, float* out, int B, int C, long HW, void* stream) {
  CIDNET_CHECK_ARG(a && d && nm && out && B > 0 && C > 0 && HW > 0);
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(grid_for((long)B * C * ((HW + 3) / 4))), dim3(kThreads), 0, (hipStream_t)stream, a, d, nm,
                     out, B, C, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_blend_bwd(const float* a, const float* d, const float* nm, const float* g, float* ga, float* gd, float* gnm, int B, int C,
                     long HW, void* stream) {
  CIDNET_CHECK_ARG(a && d && nm && g && ga && gd && gnm && B > 0 && C > 0 && HW > 0);
  const unsigned nchunk = (unsigned)(((HW + 3) / 4 + kModQ - 1) / kModQ);
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(nchunk, (unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, a, d, nm, g, ga, gd, gnm, B, C,
                     HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_resize_bilinear_fwd(const float* src, float* dst, long dst_bs, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
  CIDNET_CHECK_ARG(src && dst && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(grid_for((long)B * C * Ho * Wo, 16384)), dim3(kThreads), 0, (hipStream_t)stream, src, dst,
                     dst_bs, B, C, Hi, Wi, Ho, Wo);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_resize_bilinear_bwd(const float* gdst, long gdst_bs, float* gsrc, int B, int C, int Hi, int Wi, int Ho, int Wo,
                               void* stream) {
  CIDNET_CHECK_ARG(gdst && gsrc && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(grid_for((long)B * C * Hi * Wi, 16384)), dim3(kThreads), 0, (hipStream_t)stream, gdst,
                     gdst_bs, gsrc, B, C, Hi, Wi, Ho, Wo);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
