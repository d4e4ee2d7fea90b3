// K1/K2: RGB <-> HVI colour transform, forward and backward, for NCHW fp32 images on gfx950.
//
// Semantics follow the reference's RGB_HVI.HVIT / PHVIT (net/HVI_transform.py:16-47, 49-122) as
// restated in oracle/cidnet_oracle.py: all comparisons, divisions and the python-style modulo are
// the same IEEE fp32 operations (this file is compiled with -ffp-contract=off, fused multiply-adds
// appear only where written as fmaf), so mask / arg-max / sextant decisions are bit-exact with the
// CPU path; sin/cos come from an LDS-staged 257-entry table + angle addition (abs err < 1.5e-7),
// pow/atan2/log from the device math library.
//
// Layout: planes are contiguous (b, c, p) with p = y*W+x; each lane moves 4 consecutive pixels per
// plane (16 B per lane, 1 KiB per wave-instruction).  HBM-bound: 24 B/px forward (3 planes in, 3
// out), 36 B/px backward (+ the incoming gradient planes).
#include "common.h"
#include "trig_table.inc"

namespace cidnet {
namespace {

constexpr float kEps = 1e-8f;
constexpr float kPi = 3.14159274101257324f;      // float(3.141592653589793)
constexpr float kTwoPi = 6.28318548202514648f;   // float(2.0 * pi)
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 16384;     // HVIT forward at 32x3x1024x1024: 4.4 TB/s with 2048 blocks, 4.7-5.0 with 16384 (tools/hvi_grid_probe.py); PHVIT unchanged

// base^k for base in [1e-8, 1+1e-8] through the hardware log2 / exp2 (v_log_f32, v_exp_f32: <= 1 ulp each).
// |k log2(base)| <= 27 k, so the absolute error of the exponent is <= 3e-6 * k and the relative error of the
// result <= 2e-6 * k (4e-7 at k = 0.2) -- an order of magnitude cheaper than the library powf, which was the
// single largest VALU cost of these HBM-bound kernels.
__device__ __forceinline__ float pow_fast(float base, float k) { return __builtin_amdgcn_exp2f(k * __builtin_amdgcn_logf(base)); }

// atan2(y, x) in [-pi, pi] without the library call (which costs more than the rest of PHVIT's pixel together): the ratio
// t = min(|x|, |y|) / max(|x|, |y|) by one IEEE division, an odd minimax polynomial of atan on [0, 1] (8 coefficients in
// t^2, <= 3.5 ulp: the coefficients of SLEEF's atanf), then the octant.  What PHVIT needs beyond accuracy: for a tiny
// negative angle (y < 0 << x) the reference's h = atan2(y, x) / (2 pi) % 1 rounds to exactly 1.0f and the pixel comes out
// BLACK (hi == 6, net/HVI_transform.py:65-66,79-90).  There t^2 is below half an ulp of 1, the polynomial returns t itself,
// i.e. the correctly rounded quotient |y| / |x| -- what atan2f returns for such arguments -- so the black pixels are the
// reference's (tests/test_hvi_gpu.py: adversarial fixture, hi == 6 sets compared exactly).
__device__ __forceinline__ float atan2_poly(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mx == 0.f ? 0.f : mn / mx;
  const float t2 = t * t;
  float u = 0.00282363896258175373077393f;
  u = fmaf(u, t2, -0.0159569028764963150024414f);
  u = fmaf(u, t2, 0.0425049886107444763183594f);
  u = fmaf(u, t2, -0.0748900920152664184570312f);
  u = fmaf(u, t2, 0.106347933411598205566406f);
  u = fmaf(u, t2, -0.142027363181114196777344f);
  u = fmaf(u, t2, 0.199926957488059997558594f);
  u = fmaf(u, t2, -0.333331018686294555664062f);
  float r = fmaf(t, t2 * u, t);
  r = ay > ax ? 1.57079632679489661923f - r : r;
  r = x < 0.f ? 3.14159265358979323846f - r : r;
  return y < 0.f ? -r : r;
}

__device__ __forceinline__ void stage_trig(float2* T) {
  for (int i = threadIdx.x; i <= CIDNET_TRIG_N; i += blockDim.x) T[i] = g_trig_table[i];
  __syncthreads();
}

// sin/cos of a float angle x in [0, 2*pi] (radians): nearest table node + Cody-Waite residual.
__device__ __forceinline__ void sincos_tab(const float2* T, float x, float& s, float& c) {
  float fi = rintf(x * CIDNET_TRIG_INV_STEP);
  fi = fminf(fmaxf(fi, 0.f), (float)CIDNET_TRIG_N);
  float d = fmaf(-fi, CIDNET_TRIG_STEP_HI, x);
  d = fmaf(-fi, CIDNET_TRIG_STEP_LO, d);
  const float d2 = d * d;
  const float cd = fmaf(d2, fmaf(d2, 4.16666679e-2f, -0.5f), 1.0f);
  const float sd = fmaf(d * d2, fmaf(d2, 8.33333377e-3f, -1.66666672e-1f), d);
  const float2 t = T[(int)fi];
  c = fmaf(t.x, cd, -(t.y * sd));
  s = fmaf(t.y, cd, t.x * sd);
}

// a / b for a normal positive b: hardware reciprocal + one Newton step (<= 1.5 ulp of the quotient, sign exact).  The three
// quotients of HVIT feed VALUES only (hue, saturation: abs error < 1e-7 of a quantity in [0, 1]); every mask / arg-max
// decision of net/HVI_transform.py:21-31 is a comparison of the inputs themselves and stays bit-exact.  An IEEE division is
// ~10 VALU instructions, and at ~150 per pixel the kernel is VALU-bound, not HBM-bound.
__device__ __forceinline__ float div_fast(float a, float b) {
  float x = __builtin_amdgcn_rcpf(b);
  x = fmaf(fmaf(-b, x, 1.0f), x, x);
  return a * x;
}

struct HvitPx {
  float value, mn, d, num, hue, sat, sn, csn, base, cs, ch, cv;
  int branch;   // 0 gray, 1 R, 2 G, 3 B   (net/HVI_transform.py:23-27)
  int amax, amin;
};

// forward quantities of one pixel; everything the backward needs is recomputed from (r,g,b,k)
__device__ __forceinline__ HvitPx hvit_px(float r, float g, float b, float k, const float2* T) {
  HvitPx o;
  o.value = fmaxf(fmaxf(r, g), b);
  o.mn = fminf(fminf(r, g), b);
  o.amax = (r == o.value) ? 0 : ((g == o.value) ? 1 : 2);     // first arg-max, as torch.max(dim)
  o.amin = (r == o.mn) ? 0 : ((g == o.mn) ? 1 : 2);
  o.d = (o.value - o.mn) + kEps;
  float hue6;
  if (r == o.value) {                 // R mask is assigned last in the reference => wins ties
    o.branch = 1; o.num = g - b;
  } else if (g == o.value) {
    o.branch = 2; o.num = b - r;
  } else {
    o.branch = 3; o.num = r - g;
  }
  const float t = div_fast(o.num, o.d);
  if (o.branch == 1) hue6 = (t < 0.f) ? t + 6.0f : t;          // torch.remainder(t, 6) for |t| < 6
  else hue6 = (o.branch == 2 ? 2.0f : 4.0f) + t;
  if (o.mn == o.value) { o.branch = 0; hue6 = 0.f; }
  o.hue = hue6 * 0.166666672f;                                   // / 6 (value only: <= 1 ulp)
  o.sat = div_fast(o.value - o.mn, o.value + kEps);
  if (o.value == 0.f) o.sat = 0.f;
  const float xs = (o.value * 0.5f) * kPi;
  sincos_tab(T, xs, o.sn, o.csn);
  o.base = o.sn + kEps;
  o.cs = pow_fast(o.base, k);
  sincos_tab(T, kTwoPi * o.hue, o.cv, o.ch);
  return o;
}

__global__ __launch_bounds__(kThreads) void hvit_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ kptr,
                                                            float* __restrict__ hvi, uint8_t* __restrict__ code,
                                                            int B, long HW) {
  __shared__ float2 T[CIDNET_TRIG_N + 1];
  stage_trig(T);
  const float k = kptr[0];
  const unsigned nq = (unsigned)((HW + 3) >> 2);            // pixel quads per plane; B * nq < 2^32 (checked by the launcher)
  const unsigned total = (unsigned)B * nq;
  for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const unsigned bq = it / nq;                                // 32-bit: a 64-bit division per pixel quad cost as much as the trigonometry
    const long b = bq, p = (long)(it - bq * nq) << 2;
    const float* src = rgb + b * 3 * HW + p;
    float* dst = hvi + b * 3 * HW + p;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    f32x4 r, g, bl, oh, ov, oi;
    if (n == 4) {
      r = load4u(src); g = load4u(src + HW); bl = load4u(src + 2 * HW);
    } else {
      for (int e = 0; e < 4; ++e) {
        r[e] = e < n ? src[e] : 0.f; g[e] = e < n ? src[HW + e] : 0.f; bl[e] = e < n ? src[2 * HW + e] : 0.f;
      }
    }
    unsigned cc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      HvitPx o = hvit_px(r[e], g[e], bl[e], k, T);
      const float m = o.cs * o.sat;
      oh[e] = m * o.ch; ov[e] = m * o.cv; oi[e] = o.value;
      cc[e] = (unsigned)o.branch | ((unsigned)o.amax << 2) | ((unsigned)o.amin << 4) | ((o.value == 0.f) ? 64u : 0u);
    }
    if (n == 4) {
      store4u(dst, oh); store4u(dst + HW, ov); store4u(dst + 2 * HW, oi);
    } else {
      for (int e = 0; e < n; ++e) { dst[e] = oh[e]; dst[HW + e] = ov[e]; dst[2 * HW + e] = oi[e]; }
    }
    if (code) for (int e = 0; e < n; ++e) code[b * HW + p + e] = (uint8_t)cc[e];
  }
}

// d(loss)/d(rgb) and per-block partial of d(loss)/d(density_k).
__global__ __launch_bounds__(kThreads) void hvit_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ kptr,
                                                            const float* __restrict__ ghvi, float* __restrict__ grgb,
                                                            float* __restrict__ gk_part, int B, long HW) {
  __shared__ float2 T[CIDNET_TRIG_N + 1];
  __shared__ float red[kThreads / 64];
  stage_trig(T);
  const float k = kptr[0];
  const unsigned nq = (unsigned)((HW + 3) >> 2);            // pixel quads per plane; B * nq < 2^32 (checked by the launcher)
  const unsigned total = (unsigned)B * nq;
  float gk_acc = 0.f;
  for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const unsigned bq = it / nq;                                // 32-bit: a 64-bit division per pixel quad cost as much as the trigonometry
    const long b = bq, p = (long)(it - bq * nq) << 2;
    const long off = b * 3 * HW + p;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    f32x4 r, g, bl, gh, gv, gi, or_, og, ob;
    if (n == 4) {
      r = load4u(rgb + off); g = load4u(rgb + off + HW); bl = load4u(rgb + off + 2 * HW);
      gh = load4u(ghvi + off); gv = load4u(ghvi + off + HW); gi = load4u(ghvi + off + 2 * HW);
    } else {
      for (int e = 0; e < 4; ++e) {
        const bool ok = e < n;
        r[e] = ok ? rgb[off + e] : 0.f; g[e] = ok ? rgb[off + HW + e] : 0.f; bl[e] = ok ? rgb[off + 2 * HW + e] : 0.f;
        gh[e] = ok ? ghvi[off + e] : 0.f; gv[e] = ok ? ghvi[off + HW + e] : 0.f; gi[e] = ok ? ghvi[off + 2 * HW + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const HvitPx o = hvit_px(r[e], g[e], bl[e], k, T);
      const float A = gh[e] * o.ch + gv[e] * o.cv;
      const float g_cs = o.sat * A;
      const float g_sat = o.cs * A;
      const float g_hue6 = (o.cs * o.sat) * (gv[e] * o.ch - gh[e] * o.cv) * (kTwoPi / 6.0f);
      if (e < n) gk_acc += g_cs * o.cs * (0.693147180559945f * __builtin_amdgcn_logf(o.base));
      // cs = base^k, base = sin(value*pi/2) + eps
      float g_value = gi[e] + g_cs * k * (o.cs / o.base) * o.csn * (0.5f * kPi);
      float g_delta = 0.f;
      if (o.value != 0.f) {
        const float iv = 1.0f / (o.value + kEps);
        g_delta += g_sat * iv;
        g_value -= g_sat * (o.value - o.mn) * iv * iv;
      }
      float cr = 0.f, cg = 0.f, cb = 0.f;
      if (o.branch != 0) {
        const float id = 1.0f / o.d;
        const float gn = g_hue6 * id;
        g_delta -= g_hue6 * o.num * id * id;
        if (o.branch == 1) { cg += gn; cb -= gn; }
        else if (o.branch == 2) { cb += gn; cr -= gn; }
        else { cr += gn; cg -= gn; }
      }
      g_value += g_delta;
      const float g_mn = -g_delta;
      cr += (o.amax == 0 ? g_value : 0.f) + (o.amin == 0 ? g_mn : 0.f);
      cg += (o.amax == 1 ? g_value : 0.f) + (o.amin == 1 ? g_mn : 0.f);
      cb += (o.amax == 2 ? g_value : 0.f) + (o.amin == 2 ? g_mn : 0.f);
      or_[e] = cr; og[e] = cg; ob[e] = cb;
    }
    if (grgb) {
      if (n == 4) {
        store4u(grgb + off, or_); store4u(grgb + off + HW, og); store4u(grgb + off + 2 * HW, ob);
      } else {
        for (int e = 0; e < n; ++e) { grgb[off + e] = or_[e]; grgb[off + HW + e] = og[e]; grgb[off + 2 * HW + e] = ob[e]; }
      }
    }
  }
  const float s = block_sum(gk_acc, red);
  if (threadIdx.x == 0 && gk_part) gk_part[blockIdx.x] = s;
}

// fixed-order sum of the per-block partials: bitwise reproducible d/dk
__global__ void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = s;
}

// ---------------------------------------------------------------------------------------------
struct PhvitCfg {
  float k_host;
  int gated, gated2;
  float alpha_s, alpha;
};

struct PhvitPx {
  float H0, V0, I0;      // raw inputs (after the fused residual add)
  float I1, sn, csn, base, cs, den, H1, V1, H2, V2, H3, V3, x, y, h, u, s_pre, s, v, f, p, q, t;
  int hi;
};

__device__ __forceinline__ PhvitPx phvit_px(float H0, float V0, float I0, float k, const PhvitCfg& cfg, const float2* T) {
  PhvitPx o;
  o.H0 = H0; o.V0 = V0; o.I0 = I0;
  o.H1 = fminf(fmaxf(H0, -1.f), 1.f);
  o.V1 = fminf(fmaxf(V0, -1.f), 1.f);
  o.I1 = fminf(fmaxf(I0, 0.f), 1.f);
  sincos_tab(T, (o.I1 * 0.5f) * kPi, o.sn, o.csn);
  o.base = o.sn + kEps;
  o.cs = (k == 0.f) ? 1.0f : pow_fast(o.base, k);
  o.den = o.cs + kEps;
  // one IEEE reciprocal and two products instead of two IEEE divisions (each ~10 VALU instructions in a kernel that is as
  // VALU-bound as it is HBM-bound): the quotients differ from the reference's by at most one ulp, far inside PHVIT's 5e-6
  const float rden = 1.0f / o.den;
  o.H2 = o.H1 * rden;
  o.V2 = o.V1 * rden;
  o.H3 = fminf(fmaxf(o.H2, -1.f), 1.f);
  o.V3 = fminf(fmaxf(o.V2, -1.f), 1.f);
  o.y = o.V3 + kEps;
  o.x = o.H3 + kEps;
  float h = atan2_poly(o.y, o.x) * (1.0f / kTwoPi);  // |h| <= 0.5: fmod(h, 1) is h itself (product with the rounded reciprocal: <= 1 ulp from the quotient)
  if (h < 0.f) h += 1.0f;                           // python-style h % 1 (may round to exactly 1.0)
  o.h = h;
  o.u = (o.H3 * o.H3 + o.V3 * o.V3) + kEps;
  o.s_pre = __builtin_amdgcn_sqrtf(o.u);            // u >= 1e-8: the hardware square root (1 ulp) is enough
  if (cfg.gated) o.s_pre = o.s_pre * cfg.alpha_s;
  o.s = fminf(fmaxf(o.s_pre, 0.f), 1.f);
  o.v = fminf(fmaxf(o.I1, 0.f), 1.f);
  const float h6 = h * 6.0f;
  const float hif = floorf(h6);
  o.hi = (int)hif;
  o.f = h6 - hif;
  o.p = o.v * (1.f - o.s);
  o.q = o.v * (1.f - (o.f * o.s));
  o.t = o.v * (1.f - ((1.f - o.f) * o.s));
  return o;
}

__device__ __forceinline__ void phvit_pick(const PhvitPx& o, float& r, float& g, float& b) {
  switch (o.hi) {                       // net/HVI_transform.py:92-114; hi == 6 (or NaN) stays black
    case 0: r = o.v; g = o.t; b = o.p; break;
    case 1: r = o.q; g = o.v; b = o.p; break;
    case 2: r = o.p; g = o.v; b = o.t; break;
    case 3: r = o.p; g = o.q; b = o.v; break;
    case 4: r = o.t; g = o.p; b = o.v; break;
    case 5: r = o.v; g = o.p; b = o.q; break;
    default: r = 0.f; g = 0.f; b = 0.f; break;
  }
}

// in = hvi (+ [hv ; iv] when the residual form of net/CIDNet.py:119 is fused)
__global__ __launch_bounds__(kThreads) void phvit_fwd_kernel(const float* __restrict__ hv, const float* __restrict__ iv,
                                                             const float* __restrict__ hvi, const float* __restrict__ kdev,
                                                             PhvitCfg cfg, float* __restrict__ rgb, uint8_t* __restrict__ sext,
                                                             int B, long HW) {
  __shared__ float2 T[CIDNET_TRIG_N + 1];
  stage_trig(T);
  const float k = kdev ? kdev[0] : cfg.k_host;
  const unsigned nq = (unsigned)((HW + 3) >> 2);            // pixel quads per plane; B * nq < 2^32 (checked by the launcher)
  const unsigned total = (unsigned)B * nq;
  for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const unsigned bq = it / nq;                                // 32-bit: a 64-bit division per pixel quad cost as much as the trigonometry
    const long b = bq, p = (long)(it - bq * nq) << 2;
    const long off = b * 3 * HW + p;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    f32x4 a0, a1, a2, o0, o1, o2;
    if (n == 4) {
      a0 = load4u(hvi + off); a1 = load4u(hvi + off + HW); a2 = load4u(hvi + off + 2 * HW);
      if (hv) {
        a0 += load4u(hv + b * 2 * HW + p); a1 += load4u(hv + b * 2 * HW + HW + p); a2 += load4u(iv + b * HW + p);
      }
    } else {
      for (int e = 0; e < 4; ++e) {
        const bool ok = e < n;
        a0[e] = ok ? hvi[off + e] : 0.f; a1[e] = ok ? hvi[off + HW + e] : 0.f; a2[e] = ok ? hvi[off + 2 * HW + e] : 0.f;
        if (hv && ok) { a0[e] += hv[b * 2 * HW + p + e]; a1[e] += hv[b * 2 * HW + HW + p + e]; a2[e] += iv[b * HW + p + e]; }
      }
    }
    unsigned hh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const PhvitPx o = phvit_px(a0[e], a1[e], a2[e], k, cfg, T);
      float r, g, bl;
      phvit_pick(o, r, g, bl);
      if (cfg.gated2) { r *= cfg.alpha; g *= cfg.alpha; bl *= cfg.alpha; }
      o0[e] = r; o1[e] = g; o2[e] = bl;
      hh[e] = (unsigned)o.hi;
    }
    if (n == 4) {
      store4u(rgb + off, o0); store4u(rgb + off + HW, o1); store4u(rgb + off + 2 * HW, o2);
    } else {
      for (int e = 0; e < n; ++e) { rgb[off + e] = o0[e]; rgb[off + HW + e] = o1[e]; rgb[off + 2 * HW + e] = o2[e]; }
    }
    if (sext) for (int e = 0; e < n; ++e) sext[b * HW + p + e] = (uint8_t)hh[e];
  }
}

// gradient wrt the (summed) HVI input; optionally also written in the split (hv | iv) form that the
// two decoder heads consume (the residual add fans the same gradient out to all of them).
__global__ __launch_bounds__(kThreads) void phvit_bwd_kernel(const float* __restrict__ hv, const float* __restrict__ iv,
                                                             const float* __restrict__ hvi, const float* __restrict__ kdev,
                                                             PhvitCfg cfg, const float* __restrict__ grgb,
                                                             float* __restrict__ ghvi, float* __restrict__ ghv,
                                                             float* __restrict__ giv, int B, long HW) {
  __shared__ float2 T[CIDNET_TRIG_N + 1];
  stage_trig(T);
  const float k = kdev ? kdev[0] : cfg.k_host;
  const unsigned nq = (unsigned)((HW + 3) >> 2);            // pixel quads per plane; B * nq < 2^32 (checked by the launcher)
  const unsigned total = (unsigned)B * nq;
  for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const unsigned bq = it / nq;                                // 32-bit: a 64-bit division per pixel quad cost as much as the trigonometry
    const long b = bq, p = (long)(it - bq * nq) << 2;
    const long off = b * 3 * HW + p;
    const int n = (HW - p >= 4) ? 4 : (int)(HW - p);
    f32x4 a0, a1, a2, g0, g1, g2, d0, d1, d2;
    if (n == 4) {
      a0 = load4u(hvi + off); a1 = load4u(hvi + off + HW); a2 = load4u(hvi + off + 2 * HW);
      if (hv) {
        a0 += load4u(hv + b * 2 * HW + p); a1 += load4u(hv + b * 2 * HW + HW + p); a2 += load4u(iv + b * HW + p);
      }
      g0 = load4u(grgb + off); g1 = load4u(grgb + off + HW); g2 = load4u(grgb + off + 2 * HW);
    } else {
      for (int e = 0; e < 4; ++e) {
        const bool ok = e < n;
        a0[e] = ok ? hvi[off + e] : 0.f; a1[e] = ok ? hvi[off + HW + e] : 0.f; a2[e] = ok ? hvi[off + 2 * HW + e] : 0.f;
        if (hv && ok) { a0[e] += hv[b * 2 * HW + p + e]; a1[e] += hv[b * 2 * HW + HW + p + e]; a2[e] += iv[b * HW + p + e]; }
        g0[e] = ok ? grgb[off + e] : 0.f; g1[e] = ok ? grgb[off + HW + e] : 0.f; g2[e] = ok ? grgb[off + 2 * HW + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const PhvitPx o = phvit_px(a0[e], a1[e], a2[e], k, cfg, T);
      float gr = g0[e], gg = g1[e], gb = g2[e];
      if (cfg.gated2) { gr *= cfg.alpha; gg *= cfg.alpha; gb *= cfg.alpha; }
      float gv = 0.f, gp = 0.f, gq = 0.f, gt = 0.f;
      switch (o.hi) {
        case 0: gv = gr; gt = gg; gp = gb; break;
        case 1: gq = gr; gv = gg; gp = gb; break;
        case 2: gp = gr; gv = gg; gt = gb; break;
        case 3: gp = gr; gq = gg; gv = gb; break;
        case 4: gt = gr; gp = gg; gv = gb; break;
        case 5: gv = gr; gp = gg; gq = gb; break;
        default: break;
      }
      const float omf = 1.f - o.f;
      gv += gp * (1.f - o.s) + gq * (1.f - o.f * o.s) + gt * (1.f - omf * o.s);
      float gs = -(gp * o.v) - gq * o.v * o.f - gt * o.v * omf;
      const float gf = (gt - gq) * o.v * o.s;
      // f = 6h - floor(6h);  h = (atan2(y,x) / 2pi) % 1
      const float ga = gf * 6.0f / kTwoPi;
      const float r2 = o.x * o.x + o.y * o.y;
      float gH3 = -ga * o.y / r2;
      float gV3 = ga * o.x / r2;
      // s = clamp(s_pre, 0, 1); s_pre = sqrt(u) [* alpha_s]
      if (!(o.s_pre >= 0.f && o.s_pre <= 1.f)) gs = 0.f;
      if (cfg.gated) gs *= cfg.alpha_s;
      const float gu = gs * 0.5f / sqrtf(o.u);
      gH3 += 2.f * o.H3 * gu;
      gV3 += 2.f * o.V3 * gu;
      const float gH2 = (o.H2 >= -1.f && o.H2 <= 1.f) ? gH3 : 0.f;
      const float gV2 = (o.V2 >= -1.f && o.V2 <= 1.f) ? gV3 : 0.f;
      const float iden = 1.f / o.den;
      const float gH1 = gH2 * iden, gV1 = gV2 * iden;
      const float gcs = -(gH2 * o.H1 + gV2 * o.V1) * iden * iden;
      // v = clamp(I1,0,1) (always inside), cs = base^k with a python-float k (no grad to k)
      float gI1 = gv;
      if (k != 0.f) gI1 += gcs * k * (o.cs / o.base) * o.csn * (0.5f * kPi);
      d0[e] = (o.H0 >= -1.f && o.H0 <= 1.f) ? gH1 : 0.f;
      d1[e] = (o.V0 >= -1.f && o.V0 <= 1.f) ? gV1 : 0.f;
      d2[e] = (o.I0 >= 0.f && o.I0 <= 1.f) ? gI1 : 0.f;
    }
    if (n == 4) {
      if (ghvi) { store4u(ghvi + off, d0); store4u(ghvi + off + HW, d1); store4u(ghvi + off + 2 * HW, d2); }
      if (ghv) { store4u(ghv + b * 2 * HW + p, d0); store4u(ghv + b * 2 * HW + HW + p, d1); store4u(giv + b * HW + p, d2); }
    } else {
      for (int e = 0; e < n; ++e) {
        if (ghvi) { ghvi[off + e] = d0[e]; ghvi[off + HW + e] = d1[e]; ghvi[off + 2 * HW + e] = d2[e]; }
        if (ghv) { ghv[b * 2 * HW + p + e] = d0[e]; ghv[b * 2 * HW + HW + p + e] = d1[e]; giv[b * HW + p + e] = d2[e]; }
      }
    }
  }
}

inline int grid_for(int B, long HW) {
  const long quads = (long)B * ((HW + 3) >> 2);
  long g = (quads + kThreads - 1) / kThreads;
  if (g > kMaxBlocks) g = kMaxBlocks;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_hvit_fwd(const float* rgb, const float* density_k, float* hvi, uint8_t* branch_code, int B, int H, int W,
                    void* stream) {
  CIDNET_CHECK_ARG(rgb && density_k && hvi && B > 0 && H > 0 && W > 0);
  if ((long)B * (((long)H * W + 3) >> 2) >= (1L << 32) - 65536L * 256) return CIDNET_ERR_SHAPE;   // 32-bit quad index in the kernels
  const long HW = (long)H * W;
  hipLaunchKernelGGL(hvit_fwd_kernel, dim3(grid_for(B, HW)), dim3(kThreads), 0, (hipStream_t)stream, rgb, density_k, hvi,
                     branch_code, B, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_hvit_bwd_ws_floats(void) { return kMaxBlocks; }

int cidnet_hvit_bwd(const float* rgb, const float* density_k, const float* g_hvi, float* g_rgb, float* g_k, float* ws,
                    long ws_floats, int B, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(rgb && density_k && g_hvi && B > 0 && H > 0 && W > 0);
  if ((long)B * (((long)H * W + 3) >> 2) >= (1L << 32) - 65536L * 256) return CIDNET_ERR_SHAPE;   // 32-bit quad index in the kernels
  CIDNET_CHECK_ARG(g_rgb || g_k);
  if (g_k && (!ws || ws_floats < kMaxBlocks)) return CIDNET_ERR_WS;
  const long HW = (long)H * W;
  const int grid = grid_for(B, HW);
  hipLaunchKernelGGL(hvit_bwd_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, rgb, density_k, g_hvi, g_rgb,
                     g_k ? ws : nullptr, B, HW);
  CIDNET_LAUNCH_STATUS();
  if (g_k) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, grid, g_k);
    CIDNET_LAUNCH_STATUS();
  }
  return CIDNET_OK;
}

int cidnet_phvit_fwd(const float* hv, const float* iv, const float* hvi, const float* k_dev, float k_host, int gated,
                     float alpha_s, int gated2, float alpha, float* rgb, uint8_t* sextant, int B, int H, int W,
                     void* stream) {
  CIDNET_CHECK_ARG(hvi && rgb && B > 0 && H > 0 && W > 0);
  if ((long)B * (((long)H * W + 3) >> 2) >= (1L << 32) - 65536L * 256) return CIDNET_ERR_SHAPE;   // 32-bit quad index in the kernels
  CIDNET_CHECK_ARG((hv == nullptr) == (iv == nullptr));
  const long HW = (long)H * W;
  PhvitCfg cfg{k_host, gated, gated2, alpha_s, alpha};
  hipLaunchKernelGGL(phvit_fwd_kernel, dim3(grid_for(B, HW)), dim3(kThreads), 0, (hipStream_t)stream, hv, iv, hvi, k_dev,
                     cfg, rgb, sextant, B, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_phvit_bwd(const float* hv, const float* iv, const float* hvi, const float* k_dev, float k_host, int gated,
                     float alpha_s, int gated2, float alpha, const float* g_rgb, float* g_hvi, float* g_hv, float* g_iv,
                     int B, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(hvi && g_rgb && B > 0 && H > 0 && W > 0);
  if ((long)B * (((long)H * W + 3) >> 2) >= (1L << 32) - 65536L * 256) return CIDNET_ERR_SHAPE;   // 32-bit quad index in the kernels
  CIDNET_CHECK_ARG((hv == nullptr) == (iv == nullptr));
  CIDNET_CHECK_ARG((g_hv == nullptr) == (g_iv == nullptr));
  CIDNET_CHECK_ARG(g_hvi || g_hv);
  const long HW = (long)H * W;
  PhvitCfg cfg{k_host, gated, gated2, alpha_s, alpha};
  hipLaunchKernelGGL(phvit_bwd_kernel, dim3(grid_for(B, HW)), dim3(kThreads), 0, (hipStream_t)stream, hv, iv, hvi, k_dev,
                     cfg, g_rgb, g_hvi, g_hv, g_iv, B, HW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
