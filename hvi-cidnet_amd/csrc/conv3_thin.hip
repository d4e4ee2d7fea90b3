// K9/K11 "thin" 3x3 convolutions: one side has at most 4 channels (the stem convs 1->36 / 3->36 and the head
// convs 36->1 / 36->2 of net/CIDNet.py:21-24,32-35,39-42,50-53, their data gradients and weight gradients).
// On the MFMA implicit GEMM a 1..4-channel side is padded to a 16-wide tile and the launch is starved; these
// shapes are pure streaming (one side is 36 full-resolution planes, the arithmetic is < 2 FLOP/byte), so they
// get plain VALU stencil kernels with the tiling of dw.hip: a lane owns a 4-pixel x `rows` strip and slides a
// 3-row register window down it.  Same conventions as conv3.hip:
//   Y[b][m][y][x] = sum_{k,tap} A[m][k][tap] * Xpad[b][k][y+dy-1][x+dx-1],  A[m][k][tap] = Wt[m*w_ms + k*w_ks + tap'],
//   tap' = 8 - tap when flip (data gradient), zero or replicate padding of X.
#include "common.h"
#include "stencil.h"
#include "conv3_thin.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;

struct Strip {
  int nx4, rows, nstrips;
};

inline Strip make_strip(int H, int W, int rows) {
  Strip s{(W + 3) >> 2, rows, (H + rows - 1) / rows};
  return s;
}

struct ThinArgs {
  const float* X; long x_bs;
  const float* Wt; long w_ms, w_ks;
  float* Y; long y_bs;
  int B, M, K, H, W, flip;
  Strip st;
};

__device__ __forceinline__ void taps12(const float* p, float (&w)[9]) {   // 9 taps padded to three 16 B broadcast reads
  const f32x4* wp = reinterpret_cast<const f32x4*>(p);
  const f32x4 wa = wp[0], wb = wp[1], wc = wp[2];
  w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = wa[3]; w[4] = wb[0]; w[5] = wb[1]; w[6] = wb[2]; w[7] = wb[3]; w[8] = wc[0];
}

__device__ __forceinline__ f32x4 stencil9(const Win6& a, const Win6& b, const Win6& c, const float (&w)[9]) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = w[0] * a.v[e] + w[1] * a.v[e + 1] + w[2] * a.v[e + 2] + w[3] * b.v[e] + w[4] * b.v[e + 1] + w[5] * b.v[e + 2] +
           w[6] * c.v[e] + w[7] * c.v[e + 1] + w[8] * c.v[e + 2];
  return o;
}

// ---- K <= 4 input channels, any M: every output plane is written once, the K input windows stay in registers ----
template <int K, bool REP, bool NARROW>
__global__ __launch_bounds__(kThreads) void c3_thin_k_kernel(ThinArgs a) {
  extern __shared__ float As[];                 // [m][k][12]
  for (int i = threadIdx.x; i < a.M * K * 9; i += kThreads) {
    const int m = i / (K * 9), rem = i - m * (K * 9), k = rem / 9, t = rem - k * 9;
    As[(m * K + k) * 12 + t] = a.Wt[(long)m * a.w_ms + (long)k * a.w_ks + (a.flip ? 8 - t : t)];
  }
  __syncthreads();
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  const int xl = (int)(idx % a.st.nx4);
  const long rest = idx / a.st.nx4;
  const int strip = (int)(rest % a.st.nstrips);
  const long b = rest / a.st.nstrips;
  if (b >= a.B) return;
  const int H = a.H, W = a.W;
  int dup;
  const int x0 = lane_x0<NARROW>(xl, W, dup), y0 = strip * a.st.rows, yend = min(y0 + a.st.rows, H);
  const long HW = (long)H * W;
  const float* Xb = a.X + b * a.x_bs;
  float* Yb = a.Y + b * a.y_bs;
  Win6 w0[K], w1[K], w2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    w0[k] = load_win6<REP, NARROW>(Xb + k * HW, y0 - 1, x0, H, W);
    w1[k] = load_win6<REP, NARROW>(Xb + k * HW, y0, x0, H, W);
  }
  for (int y = y0; y < yend; ++y) {
#pragma unroll
    for (int k = 0; k < K; ++k) w2[k] = load_win6<REP, NARROW>(Xb + k * HW, y + 1, x0, H, W);
    for (int m = 0; m < a.M; ++m) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float w[9];
        taps12(As + (m * K + k) * 12, w);
        o += stencil9(w0[k], w1[k], w2[k], w);
      }
      store_px4<NARROW>(Yb + m * HW, y, x0, W, o);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) { w0[k] = w1[k]; w1[k] = w2[k]; }
  }
}

// ---- M <= 4 output channels, any K.  The outputs are few, so the parallelism has to come from the reduction: the
// four waves of a block take every fourth input plane of the same 64 strips (R rows x 4 pixels per lane), slide a
// 3-row window down each plane with the next row requested one iteration ahead, and waves 1..3 hand their partial
// sums to wave 0 through LDS (fixed order). ----
template <int M, int R, bool REP, bool NARROW>
__global__ __launch_bounds__(kThreads) void c3_thin_m_kernel(ThinArgs a) {
  extern __shared__ float As[];                 // [k][m][12], then the exchange buffer [3][R*M*4][64]
  float* xch = As + (size_t)a.K * M * 12;
  for (int i = threadIdx.x; i < a.K * M * 9; i += kThreads) {
    const int k = i / (M * 9), rem = i - k * (M * 9), m = rem / 9, t = rem - m * 9;
    As[(k * M + m) * 12 + t] = a.Wt[(long)m * a.w_ms + (long)k * a.w_ks + (a.flip ? 8 - t : t)];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  const int xl = (int)(idx % a.st.nx4);
  const long rest = idx / a.st.nx4;
  const int strip = (int)(rest % a.st.nstrips);
  const long b = rest / a.st.nstrips;
  const bool live = b < a.B;
  const int H = a.H, W = a.W;
  int dup;
  const int x0 = lane_x0<NARROW>(xl, W, dup), y0 = strip * R;
  const long HW = (long)H * W;
  f32x4 acc[R][M];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int m = 0; m < M; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const float* Xb = a.X + b * a.x_bs;
#pragma unroll 1
    for (int k = wave; k < a.K; k += 4) {
      const float* plane = Xb + k * HW;
      float w[M][9];
#pragma unroll
      for (int m = 0; m < M; ++m) taps12(As + (k * M + m) * 12, w[m]);
      Win6 w0 = load_win6<REP, NARROW>(plane, y0 - 1, x0, H, W), w1 = load_win6<REP, NARROW>(plane, y0, x0, H, W);
      Win6 nx = load_win6<REP, NARROW>(plane, y0 + 1, x0, H, W);
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const Win6 w2 = nx;
        if (i + 1 < R) nx = load_win6<REP, NARROW>(plane, y0 + i + 2, x0, H, W);
#pragma unroll
        for (int m = 0; m < M; ++m) acc[i][m] += stencil9(w0, w1, w2, w[m]);
        w0 = w1; w1 = w2;
        __builtin_amdgcn_sched_barrier(0);       // keep one row in flight: the scheduler otherwise hoists all R+2 row loads (300 VGPRs)
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) xch[(((wave - 1) * R + i) * M * 4 + m * 4 + e) * 64 + lane] = acc[i][m][e];
  }
  __syncthreads();
  if (wave == 0 && live) {
    float* Yb = a.Y + b * a.y_bs;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (y0 + i >= H) break;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        f32x4 o = acc[i][m];
#pragma unroll
        for (int wv = 0; wv < 3; ++wv)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += xch[((wv * R + i) * M * 4 + m * 4 + e) * 64 + lane];
        store_px4<NARROW>(Yb + m * HW, y0 + i, x0, W, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradients: dW[m][n][tap] = sum_{b,y,x} dY[b][m][y][x] * Xpad[b][n][y+dy-1][x+dx-1]
// Per-block partials go to slabs [b][chunk][M*N*9] (fixed order, no atomics); conv3.hip's reduce sums them.
// ---------------------------------------------------------------------------------------------
struct ThinWgArgs {
  const float* dY; long dy_bs;
  const float* X; long x_bs;
  float* slabs;
  int B, M, N, H, W, nchunk;
  Strip st;
};

// NACC accumulators per lane -> one value per block in out[i] (i < NACC), fixed order
template <int NACC>
__device__ __forceinline__ void block_reduce_store(const float (&acc)[NACC], float* red, float* out_base, const int* out_idx_lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wave * NACC + i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NACC; i += kThreads) {
    const int o = out_idx_lds[i];
    if (o >= 0) out_base[o] = (red[i] + red[NACC + i]) + (red[2 * NACC + i] + red[3 * NACC + i]);
  }
}

// N <= 4 input channels (X thin), dY has M planes; a block takes MG output channels of one chunk (MG * N * 9 <= 72
// accumulators per lane; X is a few planes and stays in L2 across the M / MG blocks that re-read it)
template <int N, bool REP, bool NARROW>
__global__ __launch_bounds__(kThreads) void c3_thin_wgrad_n_kernel(ThinWgArgs a) {
  constexpr int MG = 8 / (N <= 2 ? N : 4), NACC = MG * N * 9;
  __shared__ float red[4 * NACC];
  __shared__ int oidx[NACC];
  const int b = blockIdx.z, m0 = blockIdx.y * MG;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  for (int i = threadIdx.x; i < NACC; i += kThreads) {
    const int mm = i / (N * 9), rem = i - mm * (N * 9);
    oidx[i] = (m0 + mm < a.M) ? (m0 + mm) * (a.N * 9) + rem : -1;   // rem = n*9 + t with n < N == a.N
  }
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  const int strip = idx / a.st.nx4;
  if (strip < a.st.nstrips) {
    int dup;
    const int x0 = lane_x0<NARROW>(idx - strip * a.st.nx4, W, dup), y0 = strip * a.st.rows, yend = min(y0 + a.st.rows, H);
    const float* Xb = a.X + (long)b * a.x_bs;
    const float* Gb = a.dY + (long)b * a.dy_bs;
    Win6 w0[N], w1[N], w2[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      w0[n] = load_win6<REP, NARROW>(Xb + n * HW, y0 - 1, x0, H, W);
      w1[n] = load_win6<REP, NARROW>(Xb + n * HW, y0, x0, H, W);
    }
    for (int y = y0; y < yend; ++y) {
#pragma unroll
      for (int n = 0; n < N; ++n) w2[n] = load_win6<REP, NARROW>(Xb + n * HW, y + 1, x0, H, W);
#pragma unroll
      for (int mm = 0; mm < MG; ++mm) {
        f32x4 g = load_px4<NARROW>(Gb + (long)min(m0 + mm, a.M - 1) * HW, y, x0, W);   // rows past M: never stored
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = e < dup ? 0.f : g[e];
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              acc[(mm * N + n) * 9 + dx] += g[e] * w0[n].v[e + dx];
              acc[(mm * N + n) * 9 + 3 + dx] += g[e] * w1[n].v[e + dx];
              acc[(mm * N + n) * 9 + 6 + dx] += g[e] * w2[n].v[e + dx];
            }
      }
#pragma unroll
      for (int n = 0; n < N; ++n) { w0[n] = w1[n]; w1[n] = w2[n]; }
    }
  }
  float* slab = a.slabs + ((long)b * a.nchunk + blockIdx.x) * ((long)a.M * a.N * 9);
  block_reduce_store<NACC>(acc, red, slab, oidx);
}

// M <= 4 output channels (dY thin), X has N planes; a block takes one input plane n of one chunk
template <int M, bool REP, bool NARROW>
__global__ __launch_bounds__(kThreads) void c3_thin_wgrad_m_kernel(ThinWgArgs a) {
  constexpr int NACC = M * 9;
  __shared__ float red[4 * NACC];
  __shared__ int oidx[NACC];
  const int b = blockIdx.z, n = blockIdx.y;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  for (int i = threadIdx.x; i < NACC; i += kThreads) {
    const int m = i / 9, t = i - m * 9;
    oidx[i] = (m * a.N + n) * 9 + t;
  }
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  const int strip = idx / a.st.nx4;
  if (strip < a.st.nstrips) {
    int dup;
    const int x0 = lane_x0<NARROW>(idx - strip * a.st.nx4, W, dup), y0 = strip * a.st.rows, yend = min(y0 + a.st.rows, H);
    const float* xp = a.X + (long)b * a.x_bs + (long)n * HW;
    const float* Gb = a.dY + (long)b * a.dy_bs;
    Win6 w0 = load_win6<REP, NARROW>(xp, y0 - 1, x0, H, W), w1 = load_win6<REP, NARROW>(xp, y0, x0, H, W);
    for (int y = y0; y < yend; ++y) {
      const Win6 w2 = load_win6<REP, NARROW>(xp, y + 1, x0, H, W);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        f32x4 g = load_px4<NARROW>(Gb + (long)m * HW, y, x0, W);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = e < dup ? 0.f : g[e];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            acc[m * 9 + dx] += g[e] * w0.v[e + dx];
            acc[m * 9 + 3 + dx] += g[e] * w1.v[e + dx];
            acc[m * 9 + 6 + dx] += g[e] * w2.v[e + dx];
          }
      }
      w0 = w1; w1 = w2;
    }
  }
  float* slab = a.slabs + ((long)b * a.nchunk + blockIdx.x) * ((long)a.M * a.N * 9);
  block_reduce_store<NACC>(acc, red, slab, oidx);
}

constexpr int kWgRows = 16;

}  // namespace

bool c3_thin_applies(int M, int K) { return (M <= 4 && K <= 256) || (K <= 4 && M <= 256); }

template <int M, int R>
int launch_thin_m(ThinArgs a, int replicate, hipStream_t s) {
  a.st = make_strip(a.H, a.W, R);
  const long items = (long)a.B * a.st.nstrips * a.st.nx4;
  const dim3 grid((unsigned)((items + 63) / 64));
  const size_t lds = ((size_t)a.K * M * 12 + (size_t)3 * R * M * 4 * 64) * sizeof(float);
  if (a.W < 8) {
    if (replicate) hipLaunchKernelGGL((c3_thin_m_kernel<M, R, true, true>), grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL((c3_thin_m_kernel<M, R, false, true>), grid, dim3(kThreads), lds, s, a);
  } else {
    if (replicate) hipLaunchKernelGGL((c3_thin_m_kernel<M, R, true, false>), grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL((c3_thin_m_kernel<M, R, false, false>), grid, dim3(kThreads), lds, s, a);
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

template <int K>
int launch_thin_k(ThinArgs a, int replicate, hipStream_t s) {
  a.st = make_strip(a.H, a.W, 8);
  const long items = (long)a.B * a.st.nstrips * a.st.nx4;
  const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
  const size_t lds = (size_t)a.M * K * 12 * sizeof(float);
  if (a.W < 8) {
    if (replicate) hipLaunchKernelGGL((c3_thin_k_kernel<K, true, true>), grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL((c3_thin_k_kernel<K, false, true>), grid, dim3(kThreads), lds, s, a);
  } else {
    if (replicate) hipLaunchKernelGGL((c3_thin_k_kernel<K, true, false>), grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL((c3_thin_k_kernel<K, false, false>), grid, dim3(kThreads), lds, s, a);
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int c3_thin_conv(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, int replicate, float* Y, long y_bs,
                 int B, int M, int K, int H, int W, hipStream_t s) {
  const ThinArgs a{X, x_bs, Wt, w_ms, w_ks, Y, y_bs, B, M, K, H, W, flip, Strip{}};
  if (M <= 4) {
    switch (M) {
      case 1: return launch_thin_m<1, 8>(a, replicate, s);
      case 2: return launch_thin_m<2, 4>(a, replicate, s);
      case 3: return launch_thin_m<3, 4>(a, replicate, s);
      default: return launch_thin_m<4, 4>(a, replicate, s);
    }
  }
  switch (K) {
    case 1: return launch_thin_k<1>(a, replicate, s);
    case 2: return launch_thin_k<2>(a, replicate, s);
    case 3: return launch_thin_k<3>(a, replicate, s);
    default: return launch_thin_k<4>(a, replicate, s);
  }
}

int c3_thin_wgrad_chunks(int H, int W) {
  const Strip st = make_strip(H, W, kWgRows);
  return (int)(((long)st.nstrips * st.nx4 + kThreads - 1) / kThreads);
}

template <int N>
void launch_wg_n(const ThinWgArgs& a, int replicate, hipStream_t s) {
  constexpr int MG = 8 / (N <= 2 ? N : 4);
  const dim3 grid((unsigned)a.nchunk, (unsigned)((a.M + MG - 1) / MG), (unsigned)a.B);
  if (a.W < 8) {
    if (replicate) hipLaunchKernelGGL((c3_thin_wgrad_n_kernel<N, true, true>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((c3_thin_wgrad_n_kernel<N, false, true>), grid, dim3(kThreads), 0, s, a);
  } else {
    if (replicate) hipLaunchKernelGGL((c3_thin_wgrad_n_kernel<N, true, false>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((c3_thin_wgrad_n_kernel<N, false, false>), grid, dim3(kThreads), 0, s, a);
  }
}

template <int M>
void launch_wg_m(const ThinWgArgs& a, int replicate, hipStream_t s) {
  const dim3 grid((unsigned)a.nchunk, (unsigned)a.N, (unsigned)a.B);
  if (a.W < 8) {
    if (replicate) hipLaunchKernelGGL((c3_thin_wgrad_m_kernel<M, true, true>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((c3_thin_wgrad_m_kernel<M, false, true>), grid, dim3(kThreads), 0, s, a);
  } else {
    if (replicate) hipLaunchKernelGGL((c3_thin_wgrad_m_kernel<M, true, false>), grid, dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((c3_thin_wgrad_m_kernel<M, false, false>), grid, dim3(kThreads), 0, s, a);
  }
}

int c3_thin_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int replicate, float* slabs, int B, int M, int N, int H,
                  int W, hipStream_t s) {
  const ThinWgArgs a{dY, dy_bs, X, x_bs, slabs, B, M, N, H, W, c3_thin_wgrad_chunks(H, W), make_strip(H, W, kWgRows)};
  if (N <= 4) {
    switch (N) {
      case 1: launch_wg_n<1>(a, replicate, s); break;
      case 2: launch_wg_n<2>(a, replicate, s); break;
      case 3: launch_wg_n<3>(a, replicate, s); break;
      default: launch_wg_n<4>(a, replicate, s); break;
    }
  } else {
    switch (M) {
      case 1: launch_wg_m<1>(a, replicate, s); break;
      case 2: launch_wg_m<2>(a, replicate, s); break;
      case 3: launch_wg_m<3>(a, replicate, s); break;
      default: launch_wg_m<4>(a, replicate, s); break;
    }
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // namespace cidnet
