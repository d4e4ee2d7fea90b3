// Bilinear resampling (align_corners=True, nn.UpsamplingBilinear2d) and PReLU pieces of
// NormDownsample / NormUpsample (net/transformer_utils.py:38-43, 62-70):
//   down_prelu_fwd : pre = bilinear(t -> floor(H/2) x floor(W/2)),  out = PReLU(pre)
//   prelu_bwd      : d_pre = d_out * (pre > 0 ? 1 : a),  d_a = sum d_out * min-side(pre)
//   bilinear_bwd   : adjoint of the bilinear map in GATHER form (no atomics, bitwise reproducible):
//                    every input pixel sums the output pixels whose 2x2 footprint contains it.
// All HBM-bound elementwise / small-stencil kernels: lanes run along x for coalesced rows.
#include "common.h"
#include "cidnet_hip.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

struct Tap1 {
  int i0, i1;
  float l1;   // weight of i1; weight of i0 is 1 - l1
};

__device__ __forceinline__ Tap1 src_tap(int o, float scale, int in) {
  const float f = scale * (float)o;
  Tap1 t;
  t.i0 = (int)f;
  t.l1 = fminf(fmaxf(f - (float)t.i0, 0.f), 1.f);
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  return t;
}

__global__ __launch_bounds__(kThreads) void down_prelu_kernel(const float* __restrict__ t, const float* __restrict__ slope,
                                                              float* __restrict__ pre, float* __restrict__ out, long planes,
                                                              int H, int W, int h, int w) {
  const long total = planes * h * w;
  const float a = slope[0];
  const float sh = ac_scale(H, h), sw = ac_scale(W, w);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const long pl = i / ((long)w * h);
    const Tap1 ty = src_tap(y, sh, H), tx = src_tap(x, sw, W);
    const float* p = t + pl * (long)H * W;
    const float top = (1.f - tx.l1) * p[(long)ty.i0 * W + tx.i0] + tx.l1 * p[(long)ty.i0 * W + tx.i1];
    const float bot = (1.f - tx.l1) * p[(long)ty.i1 * W + tx.i0] + tx.l1 * p[(long)ty.i1 * W + tx.i1];
    const float v = (1.f - ty.l1) * top + ty.l1 * bot;
    if (pre) pre[i] = v;
    out[i] = v > 0.f ? v : a * v;
  }
}

__global__ __launch_bounds__(kThreads) void prelu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pre,
                                                             const float* __restrict__ slope, float* __restrict__ dpre,
                                                             float* __restrict__ part, long n) {
  __shared__ float red[kThreads / 64];
  const float a = slope[0];
  float acc = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g = load4u(dout + 4 * i), p = load4u(pre + 4 * i);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = p[e] > 0.f ? g[e] : a * g[e];
      acc += p[e] > 0.f ? 0.f : g[e] * p[e];
    }
    store4u(dpre + 4 * i, o);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float g = dout[i], p = pre[i];
    dpre[i] = p > 0.f ? g : a * g;
    acc += p > 0.f ? 0.f : g * p;
  }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void sum_parts_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += part[i];
  const float s = block_sum(a, red);
  if (threadIdx.x == 0) out[0] = s;
}

// Per-dimension adjoint tables: for input index i the (at most kTabK) output indices whose 2-tap
// footprint contains i, with their weights.  Membership is decided with the forward's own float
// arithmetic (src_tap), so the adjoint matches the forward exactly.
constexpr int kTabK = 6;

__global__ void bilinear_tab_kernel(int* __restrict__ idx, float* __restrict__ wgt, int in, int out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in) return;
  const float s = ac_scale(in, out);
  const float is = s > 0.f ? 1.f / s : 0.f;
  int lo = s > 0.f ? (int)floorf((float)(i - 1) * is) - 1 : 0;
  int hi = s > 0.f ? (int)ceilf((float)(i + 1) * is) + 1 : out - 1;
  lo = max(lo, 0); hi = min(hi, out - 1);
  int n = 0;
  for (int o = lo; o <= hi && n < kTabK; ++o) {
    const Tap1 t = src_tap(o, s, in);
    float w = 0.f;
    if (t.i0 == i) w += 1.f - t.l1;
    if (t.i1 == i) w += t.l1;
    if (w != 0.f) { idx[i * kTabK + n] = o; wgt[i * kTabK + n] = w; ++n; }
  }
  for (; n < kTabK; ++n) { idx[i * kTabK + n] = 0; wgt[i * kTabK + n] = 0.f; }
}

// din[pl][yi][xi] = sum over table taps of wy * wx * dout[pl][yo][xo]   (gather: no atomics)
// A lane owns one input column xi: its x taps are read once into registers and reused for every row the wave
// visits; a wave owns whole input rows (grid-stride over planes * Hi), so the y taps are wave-uniform (scalar
// loads, uniform early exit) and a row's KX gathers hit neighbouring addresses across the wave.  KX bounds the
// x taps of this launch (2 for the x0.5 adjoint, up to kTabK for x2); unused table slots hold (index 0, weight 0).
template <int KX, int KY, int R>
__global__ __launch_bounds__(kThreads) void bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                const int* __restrict__ yidx, const float* __restrict__ ywgt,
                                                                const int* __restrict__ xidx, const float* __restrict__ xwgt,
                                                                long nrows, int Hi, int Wi, int Ho, int Wo) {
  const int xi = blockIdx.x * 64 + (threadIdx.x & 63);
  const bool live = xi < Wi;
  int xo[KX];
  float xw[KX];
#pragma unroll
  for (int k = 0; k < KX; ++k) {
    xo[k] = live ? xidx[xi * kTabK + k] : 0;
    xw[k] = live ? xwgt[xi * kTabK + k] : 0.f;
  }
  // R rows per wave and trip, every tap unrolled (slots past a row's taps carry weight 0 and index 0): the R * KY * KX
  // gathers of a trip are independent of one another, where a loop with an early exit paid the table load and the data
  // load of each tap one after the other
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long stride = (long)gridDim.y * (kThreads / 64) * R;
  for (long r0 = ((long)blockIdx.y * (kThreads / 64) + wave) * R; r0 < nrows; r0 += stride) {
    float s[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long r = r0 + j < nrows ? r0 + j : nrows - 1;
      const int yi = (int)(r % Hi);
      const float* g = dout + (r / Hi) * (long)Ho * Wo;
      float acc = 0.f;
#pragma unroll
      for (int ky = 0; ky < KY; ++ky) {
        const float wy = ywgt[yi * kTabK + ky];
        const float* row = g + (long)yidx[yi * kTabK + ky] * Wo;
        float rs = 0.f;
#pragma unroll
        for (int k = 0; k < KX; ++k) rs += xw[k] * row[xo[k]];
        acc += wy * rs;
      }
      s[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
      if (live && r0 + j < nrows) din[(r0 + j) * Wi + xi] = s[j];
  }
}

// The x2 adjoint (NormUpsample backward: din is half the size of dout) reads 4x more than it writes and every output
// row feeds two input rows, every output column two input columns: gathered from global memory that is ~25 dword loads per
// result through the L1 (245 us at 8x36x400x600 -> 200x300, 1.4 TB/s).  Here a block stages the contiguous band of dout
// rows that its TR input rows need into LDS with coalesced 16-byte loads (each dout element is read from memory exactly
// once per band, bands overlap by ~3 rows) and gathers from LDS.  A thread owns up to NC fixed columns (x taps in
// registers); the y taps of a row are block-uniform.  Same taps, same summation order as bilinear_bwd_kernel.
template <int KX, int KY, int NC, int TR>
__global__ __launch_bounds__(kThreads) void bilinear_bwd_band_kernel(const float* __restrict__ dout, float* __restrict__ din,
                                                                     const int* __restrict__ yidx, const float* __restrict__ ywgt,
                                                                     const int* __restrict__ xidx, const float* __restrict__ xwgt,
                                                                     int Hi, int Wi, int Ho, int Wo, int tiles_per_plane, int max_rows) {
  extern __shared__ float band[];                 // [rows][Wo]
  const long pl = blockIdx.x / tiles_per_plane;
  const int r0 = (blockIdx.x - (int)pl * tiles_per_plane) * TR, r1 = min(r0 + TR, Hi);
  int xo[NC][KX];
  float xw[NC][KX];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int xi = threadIdx.x + c * kThreads;
#pragma unroll
    for (int k = 0; k < KX; ++k) {
      xo[c][k] = xi < Wi ? xidx[xi * kTabK + k] : 0;
      xw[c][k] = xi < Wi ? xwgt[xi * kTabK + k] : 0.f;
    }
  }
  // band of output rows [ylo, yhi]: taps ascend within a row (packed at the front of its table row) and from row to row,
  // so the band runs from the first tap of the first row that has any to the last tap of the last row that has any (a
  // row has none when the map skips inputs)
  int ylo = 0, yhi = -1;
  for (int yi = r0; yi < r1; ++yi)
    if (ywgt[yi * kTabK] != 0.f) { ylo = yidx[yi * kTabK]; break; }
  for (int yi = r1 - 1; yi >= r0 && yhi < 0; --yi)
    for (int k = KY - 1; k >= 0; --k)
      if (ywgt[yi * kTabK + k] != 0.f) { yhi = yidx[yi * kTabK + k]; break; }
  if (yhi < ylo) yhi = ylo;
  const int nrow = min(yhi - ylo + 1, max_rows);
  const float* src = dout + (pl * Ho + ylo) * (long)Wo;
  const int n = nrow * Wo, n4 = n >> 2;
  for (int i = threadIdx.x; i < n4; i += kThreads) {
    const f32x4 v = load4u(src + 4 * i);
    band[4 * i] = v[0]; band[4 * i + 1] = v[1]; band[4 * i + 2] = v[2]; band[4 * i + 3] = v[3];
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) band[i] = src[i];
  __syncthreads();
  for (int yi = r0; yi < r1; ++yi) {
    float s[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = 0.f;
#pragma unroll
    for (int ky = 0; ky < KY; ++ky) {
      const float wy = ywgt[yi * kTabK + ky];
      // rows past the band can only belong to zero-weight slots (index 0): keep the address inside the band
      const int rr = min(max(yidx[yi * kTabK + ky] - ylo, 0), nrow - 1);
      const float* row = band + rr * Wo;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float rs = 0.f;
#pragma unroll
        for (int k = 0; k < KX; ++k) rs += xw[c][k] * row[xo[c][k]];
        s[c] += wy * rs;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int xi = threadIdx.x + c * kThreads;
      if (xi < Wi) din[(pl * Hi + yi) * (long)Wi + xi] = s[c];
    }
  }
}

// y = a + b (residual sums that no GEMM epilogue absorbs)
__global__ __launch_bounds__(kThreads) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ y, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    store4u(y + 4 * i, load4u(a + 4 * i) + load4u(b + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    y[i] = a[i] + b[i];
  }
}

inline int grid_for(long n, int cap = 4096) {
  long g = (n + kThreads - 1) / kThreads;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

constexpr int kPreluBlocks = 1024;

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

int cidnet_down_prelu_fwd(const float* t, const float* slope, float* pre, float* out, int B, int C, int H, int W,
                          void* stream) {
  CIDNET_CHECK_ARG(t && slope && out && B > 0 && C > 0 && H > 1 && W > 1);
  const int h = H / 2, w = W / 2;
  const long planes = (long)B * C;
  hipLaunchKernelGGL(down_prelu_kernel, dim3(grid_for(planes * h * w)), dim3(kThreads), 0, (hipStream_t)stream, t, slope, pre,
                     out, planes, H, W, h, w);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_prelu_bwd_ws_floats(void) { return kPreluBlocks; }

int cidnet_prelu_bwd(const float* dout, const float* pre, const float* slope, float* dpre, float* dslope, float* ws,
                     long ws_floats, long n, void* stream) {
  CIDNET_CHECK_ARG(dout && pre && slope && dpre && dslope && ws && n > 0);
  if (ws_floats < kPreluBlocks) return CIDNET_ERR_WS;
  const int grid = grid_for((n + 3) / 4, kPreluBlocks);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, dout, pre, slope, dpre, ws, n);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, grid, dslope);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_bilinear_bwd_ws_floats(int Hi, int Wi) { return 2L * kTabK * ((long)Hi + Wi); }

/* the per-axis tap tables alone (they depend on the four sizes only: a caller may compute them once per shape) */
int cidnet_bilinear_bwd_tabs(float* tabs, long tabs_floats, int Hi, int Wi, int Ho, int Wo, void* stream) {
  CIDNET_CHECK_ARG(tabs && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  if (tabs_floats < cidnet_bilinear_bwd_ws_floats(Hi, Wi)) return CIDNET_ERR_WS;
  hipStream_t s = (hipStream_t)stream;
  int* yidx = reinterpret_cast<int*>(tabs);
  float* ywgt = tabs + (long)kTabK * Hi;
  int* xidx = reinterpret_cast<int*>(tabs + 2L * kTabK * Hi);
  float* xwgt = tabs + 2L * kTabK * Hi + (long)kTabK * Wi;
  hipLaunchKernelGGL(bilinear_tab_kernel, dim3((Hi + 255) / 256), dim3(256), 0, s, yidx, ywgt, Hi, Ho);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(bilinear_tab_kernel, dim3((Wi + 255) / 256), dim3(256), 0, s, xidx, xwgt, Wi, Wo);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_bilinear_bwd(const float* dout, float* din, float* ws, long ws_floats, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        void* stream) {
  CIDNET_CHECK_ARG(dout && din && ws && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  const int rc = cidnet_bilinear_bwd_tabs(ws, ws_floats, Hi, Wi, Ho, Wo, stream);
  if (rc != CIDNET_OK) return rc;
  return cidnet_bilinear_bwd_pre(dout, din, ws, B, C, Hi, Wi, Ho, Wo, stream);
}

/* the adjoint with tables already computed by cidnet_bilinear_bwd_tabs for the same four sizes */
int cidnet_bilinear_bwd_pre(const float* dout, float* din, const float* tabs, int B, int C, int Hi, int Wi, int Ho, int Wo,
                            void* stream) {
  CIDNET_CHECK_ARG(dout && din && tabs && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
  hipStream_t s = (hipStream_t)stream;
  const int* yidx = reinterpret_cast<const int*>(tabs);
  const float* ywgt = tabs + (long)kTabK * Hi;
  const int* xidx = reinterpret_cast<const int*>(tabs + 2L * kTabK * Hi);
  const float* xwgt = tabs + 2L * kTabK * Hi + (long)kTabK * Wi;
  // taps of one input index: outputs o with floor(o * s) in {i-1, i}, s = (in-1)/(out-1): at most floor(2/s) + 1
  const double sx = Wo > 1 ? (double)(Wi - 1) / (double)(Wo - 1) : 0.0, sy = Ho > 1 ? (double)(Hi - 1) / (double)(Ho - 1) : 0.0;
  const int kx = sx > 0.0 ? (int)(2.0 / sx + 1e-3) + 1 : kTabK, ky = sy > 0.0 ? (int)(2.0 / sy + 1e-3) + 1 : kTabK;
  const int k = kx > ky ? kx : ky;
  // the two resampling adjoints of the model (x2 and x0.5, up to 5 resp. 2 taps per axis): band kernel when the band of
  // output rows that TR input rows need fits the LDS budget
  if ((k <= 2 || (k > 3 && k <= 5)) && Wi <= 3 * kThreads && sy > 0.0) {
    const int TR = k <= 2 ? 16 : 8;
    const int max_rows = (int)((TR + 1) / sy) + 4;             // outputs per input row = 1 / sy
    const size_t lds = (size_t)max_rows * Wo * sizeof(float);
    if (lds <= 64 * 1024) {
      const int tiles = (Hi + TR - 1) / TR;
      const dim3 grid((unsigned)((long)B * C * tiles));
      const int nc = (Wi + kThreads - 1) / kThreads;
#define CIDNET_BAND(KK, NC, TRR)                                                                                           \
  hipLaunchKernelGGL((bilinear_bwd_band_kernel<KK, KK, NC, TRR>), grid, dim3(kThreads), lds, s, dout, din, yidx, ywgt, xidx, xwgt, Hi, \
                     Wi, Ho, Wo, tiles, max_rows)
      if (k <= 2) {
        if (nc == 1) CIDNET_BAND(2, 1, 16); else if (nc == 2) CIDNET_BAND(2, 2, 16); else CIDNET_BAND(2, 3, 16);
      } else {
        if (nc == 1) CIDNET_BAND(5, 1, 8); else if (nc == 2) CIDNET_BAND(5, 2, 8); else CIDNET_BAND(5, 3, 8);
      }
#undef CIDNET_BAND
      CIDNET_LAUNCH_STATUS();
      return CIDNET_OK;
    }
  }
  const long nrows = (long)B * C * Hi;
  const int gx = (Wi + 63) / 64;
  const int R = k <= 2 ? 4 : 2;
  long gy = (nrows + (kThreads / 64) * R - 1) / ((kThreads / 64) * R);
  const long cap = 16384 / gx > 0 ? 16384 / gx : 1;
  if (gy > cap) gy = cap;
  const dim3 grid((unsigned)gx, (unsigned)gy);
#define CIDNET_BILINEAR_BWD(KK, RR) \
  hipLaunchKernelGGL((bilinear_bwd_kernel<KK, KK, RR>), grid, dim3(kThreads), 0, s, dout, din, yidx, ywgt, xidx, xwgt, nrows, Hi, Wi, Ho, Wo)
  if (k <= 2) CIDNET_BILINEAR_BWD(2, 4);
  else if (k <= 3) CIDNET_BILINEAR_BWD(3, 2);
  else if (k <= 5) CIDNET_BILINEAR_BWD(5, 2);
  else CIDNET_BILINEAR_BWD(kTabK, 2);
#undef CIDNET_BILINEAR_BWD
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_add(const float* a, const float* b, float* y, long n, void* stream) {
  CIDNET_CHECK_ARG(a && b && y && n > 0);
  hipLaunchKernelGGL(add_kernel, dim3(grid_for((n + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, a, b, y, n);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
