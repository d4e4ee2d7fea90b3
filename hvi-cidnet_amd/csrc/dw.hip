// K5 / K8 stencil parts: depthwise 3x3 convolution (zero pad 1) forward / data-gradient (flipped
// taps) / weight-gradient, and the IEL gate  g = (tanh(dw1(u1)) + u1) * (tanh(dw2(u2)) + u2)
// forward and backward (reference: net/LCA.py:14,16,53-55,62-65).
//
// All kernels share one tiling: a lane owns a 4-pixel-wide, `rows`-tall column strip of one (b,c)
// plane and slides a 3-row register window down it, so each input row is fetched once per strip
// (+ halo rows per strip) as one 16 B load plus two neighbour scalars that hit L1.  Consecutive lanes
// cover consecutive 16 B pieces of a row (dense: ceil(W/4) lanes per row, no padding to a power of
// two), then the next strip.  The strip height is picked per launch (pick_tiling): as tall as the
// plane count allows, because every strip re-reads its halo rows.  HBM-bound by construction.
#include <type_traits>
#include "common.h"
#include "cidnet_hip.h"
#include "stencil.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kMinRows = 8, kMaxRows = 24;
constexpr int kSub = 4;     // 256-item groups a fused-backward block walks before reducing

__device__ __forceinline__ f32x4 stencil(const Win6& a, const Win6& b, const Win6& c, const float* w) {
  // Tap-major: the four pixels' FMA chains advance together.  Written pixel by pixel, each output compiles to one
  // 9-deep dependent chain after the other, and a wave then issues at the FMA latency instead of the issue rate (these
  // kernels are VALU-bound).  Per pixel the operations and their order are unchanged: results are bit-identical.
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = w[0] * a.v[e];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaf(w[1], a.v[e + 1], v[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaf(w[2], a.v[e + 2], v[e]);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(w[3 + t], b.v[e + t], v[e]);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(w[6 + t], c.v[e + t], v[e]);
  return f32x4{v[0], v[1], v[2], v[3]};
}

struct Item {
  long bc;
  int y0, x0, dup;
  bool live;
};

struct Tiling {
  int nx4;       // lanes per row = ceil(W / 4)
  int rows;      // strip height
  int nstrips;   // ceil(H / rows)
};

// flattened work decomposition: (bc * nstrips + strip) * nx4 + xl
template <bool NARROW>
__device__ __forceinline__ Item decode_item(long planes, Tiling tl, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  Item it;
  const long xl = idx % tl.nx4;
  const long rest = idx / tl.nx4;
  const long strip = rest % tl.nstrips;
  it.bc = rest / tl.nstrips;
  it.y0 = (int)strip * tl.rows;
  it.x0 = lane_x0<NARROW>((int)xl, W, it.dup);
  it.live = it.bc < planes;
  return it;
}

// Strip height.  A strip costs rows + halo row passes, so taller strips re-read less; but these kernels are
// latency-bound per lane (one row per iteration), so the launch must keep several rounds of resident waves
// (`min_lanes`), and in the block-per-plane kernels a plane's items are rounded up to whole waves.  Cost model:
// lane-iterations = lanes (rounded to waves if per_plane) x (rows + halo); heights above kMinRows are only
// taken while the launch keeps min_lanes lanes.
#ifdef CIDNET_DEBUG
int g_dw_force_rows = 0;     // timing studies only (cidnet_debug_dw_rows; -DCIDNET_DEBUG builds)
#else
constexpr int g_dw_force_rows = 0;
#endif

inline Tiling pick_tiling(long planes, int H, int W, int halo, long min_lanes, bool per_plane) {
  const int nx4 = (W + 3) >> 2;
  if (g_dw_force_rows > 0) {
    const int r = g_dw_force_rows < H ? g_dw_force_rows : H;
    return Tiling{nx4, r, (H + r - 1) / r};
  }
  Tiling best{nx4, 0, 0};
  long best_cost = -1;
  const int lo = H < 4 ? H : 4, hi = H < kMaxRows ? H : kMaxRows;
  for (int r = lo; r <= hi; ++r) {
    const int ns = (H + r - 1) / r;
    const long items = (long)ns * nx4;
    if (r > kMinRows && planes * items < min_lanes) break;
    const long lanes = per_plane ? ((items + 63) / 64) * 64 : items;
    const long cost = lanes * (r + halo);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best.rows = r; best.nstrips = ns; }
  }
  return best;
}

inline long n_items(long planes, Tiling tl) { return planes * tl.nstrips * tl.nx4; }

__device__ __forceinline__ void load_w9(const float* w1, const float* w2, int csplit, int c, bool flip, float* w) {
  const float* src = (c < csplit) ? w1 + (long)c * 9 : w2 + (long)(c - csplit) * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = src[flip ? 8 - t : t];
}

// out = dwconv3x3(in) [+ addend]
template <bool NARROW, class T = float>     // T: storage type of in / addend / out (float, or bf16_t in the bf16 mode)
__global__ __launch_bounds__(kThreads) void dw3x3_kernel(const T* __restrict__ in, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, int csplit,
                                                         const T* __restrict__ addend, T* __restrict__ out, int flip,
                                                         int B, int C, int H, int W, Tiling tl) {
  const Item it = decode_item<NARROW>((long)B * C, tl, W);
  if (!it.live) return;
  float w[9];
  load_w9(w1, w2, csplit, (int)(it.bc % C), flip != 0, w);
  const long HW = (long)H * W;
  const T* ip = in + it.bc * HW;
  T* op = out + it.bc * HW;
  const T* ap = addend ? addend + it.bc * HW : nullptr;
  Win6 r0 = load_win6<false, NARROW>(ip, it.y0 - 1, it.x0, H, W);
  Win6 r1 = load_win6<false, NARROW>(ip, it.y0, it.x0, H, W);
  const int yend = min(it.y0 + tl.rows, H);
  for (int y = it.y0; y < yend; ++y) {
    const Win6 r2 = load_win6<false, NARROW>(ip, y + 1, it.x0, H, W);
    f32x4 o = stencil(r0, r1, r2, w);
    if (ap) {
      o += load_px4<NARROW>(ap, y, it.x0, W);
    }
    store_px4<NARROW>(op, y, it.x0, W, o);
    r0 = r1; r1 = r2;
  }
}

// IEL gate.  u: (B, 2h, H, W); channel c pairs u1 = u[c], u2 = u[h + c].
// MODE 0: g = s1*s2.   MODE 1 (backward): da = dg * s_other * (1 - t^2), ds = dg * s_other.
template <int MODE, bool NARROW>
__global__ __launch_bounds__(kThreads) void iel_gate_kernel(const float* __restrict__ u, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, const float* __restrict__ dg,
                                                            float* __restrict__ g, float* __restrict__ da,
                                                            float* __restrict__ ds, int B, int h, int H, int W, Tiling tl) {
  const Item it = decode_item<NARROW>((long)B * h, tl, W);
  if (!it.live) return;
  const int c = (int)(it.bc % h);
  const long b = it.bc / h;
  float wa[9], wb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { wa[t] = w1[(long)c * 9 + t]; wb[t] = w2[(long)c * 9 + t]; }
  const long HW = (long)H * W;
  const float* p1 = u + (b * 2 * h + c) * HW;
  const float* p2 = u + (b * 2 * h + h + c) * HW;
  Win6 a0 = load_win6<false, NARROW>(p1, it.y0 - 1, it.x0, H, W), a1 = load_win6<false, NARROW>(p1, it.y0, it.x0, H, W);
  Win6 b0 = load_win6<false, NARROW>(p2, it.y0 - 1, it.x0, H, W), b1 = load_win6<false, NARROW>(p2, it.y0, it.x0, H, W);
  const int yend = min(it.y0 + tl.rows, H);
  for (int y = it.y0; y < yend; ++y) {
    const Win6 a2 = load_win6<false, NARROW>(p1, y + 1, it.x0, H, W);
    const Win6 b2 = load_win6<false, NARROW>(p2, y + 1, it.x0, H, W);
    const f32x4 c1 = stencil(a0, a1, a2, wa), c2 = stencil(b0, b1, b2, wb);
    f32x4 t1, t2, s1, s2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t1[e] = tanh_fast(c1[e]); t2[e] = tanh_fast(c2[e]);
      s1[e] = t1[e] + a1.v[1 + e]; s2[e] = t2[e] + b1.v[1 + e];
    }
    if (MODE == 0) {
      store_px4<NARROW>(g + it.bc * HW, y, it.x0, W, s1 * s2);
    } else {
      const Win6 gr = load_win6<false, NARROW>(dg + it.bc * HW, y, it.x0, H, W);
      f32x4 d1, d2, e1, e2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d1[e] = gr.v[1 + e] * s2[e]; d2[e] = gr.v[1 + e] * s1[e];
        e1[e] = d1[e] * (1.f - t1[e] * t1[e]); e2[e] = d2[e] * (1.f - t2[e] * t2[e]);
      }
      store_px4<NARROW>(ds + (b * 2 * h + c) * HW, y, it.x0, W, d1);
      store_px4<NARROW>(ds + (b * 2 * h + h + c) * HW, y, it.x0, W, d2);
      store_px4<NARROW>(da + (b * 2 * h + c) * HW, y, it.x0, W, e1);
      store_px4<NARROW>(da + (b * 2 * h + h + c) * HW, y, it.x0, W, e2);
    }
    a0 = a1; a1 = a2; b0 = b1; b1 = b2;
  }
}

// Forward of the IEL middle in one pass (net/LCA.py:61-65): u = dwconv(pin) (2h channels), then the gate
// g = (tanh(dw1(u1)) + u1) * (tanh(dw2(u2)) + u2).  Run separately, u is written by one kernel and read back by the
// next; here a lane slides a 3-row, 8-wide window of pin down its strip, forms each row of u on a 6-wide window
// (own 4 pixels + the 1-pixel halo the gate's convolutions need), stores its own 4 pixels of u (the backward needs u)
// and the gate row.  HBM traffic: read pin (2h), write u (2h) and g (h): 5h instead of 7h planes.
template <bool NARROW, class T>      // T: storage type of the hidden tensors pin, u, g (float, or bf16_t in the bf16 storage mode)
__global__ __launch_bounds__(kThreads) void iel_dw_gate_kernel(const T* __restrict__ pin, const float* __restrict__ wdw,
                                                               const float* __restrict__ w1, const float* __restrict__ w2,
                                                               T* __restrict__ u, T* __restrict__ g, int B, int h, int H,
                                                               int W, Tiling tl) {
  const Item it = decode_item<NARROW>((long)B * h, tl, W);
  if (!it.live) return;
  const int c = (int)(it.bc % h);
  const long b = it.bc / h;
  float wpa[9], wpb[9], wa[9], wb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wpa[t] = wdw[(long)c * 9 + t]; wpb[t] = wdw[(long)(h + c) * 9 + t];
    wa[t] = w1[(long)c * 9 + t]; wb[t] = w2[(long)c * 9 + t];
  }
  const long HW = (long)H * W;
  const T* p1 = pin + (b * 2 * h + c) * HW;
  const T* p2 = pin + (b * 2 * h + h + c) * HW;
  T* u1 = u + (b * 2 * h + c) * HW;
  T* u2 = u + (b * 2 * h + h + c) * HW;
  T* gp = g + it.bc * HW;
  const int x0 = it.x0, y0 = it.y0, yend = min(y0 + tl.rows, H);
  // u on the columns x0-1 .. x0+4 of row r, from pin rows r-1, r, r+1 (8-wide); zero outside the image
  auto u_row = [&](const Win8& a, const Win8& m, const Win8& z, const float (&w)[9], int r) {
    Win6 o;
    float v[6];                               // tap-major over the six columns (see stencil())
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = w[0] * a.v[i];
#pragma unroll
    for (int t = 1; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = fmaf(w[t], a.v[i + t], v[i]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = fmaf(w[3 + t], m.v[i + t], v[i]);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = fmaf(w[6 + t], z.v[i + t], v[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int x = x0 - 1 + i;
      o.v[i] = (r >= 0 && r < H && x >= 0 && x < W) ? v[i] : 0.f;
    }
    return o;
  };
  // Rolling windows live in three slots whose roles (row r-1, r, r+1) rotate with the row phase; the row loop is
  // unrolled by three so every slot index is a compile-time constant and no register moves are spent on the shift
  // (they were 17 % of the loop's VALU instructions in a kernel that is VALU-issue-bound).
  Win8 A[3], Bw[3];
  Win6 UA[3], UB[3];                             // u rows of the two channels
  A[1] = load_win8<NARROW>(p1, y0 - 2, x0, H, W); A[2] = load_win8<NARROW>(p1, y0 - 1, x0, H, W);
  Bw[1] = load_win8<NARROW>(p2, y0 - 2, x0, H, W); Bw[2] = load_win8<NARROW>(p2, y0 - 1, x0, H, W);
#pragma unroll
  for (int i = 0; i < 6; ++i) { UA[1].v[i] = UA[2].v[i] = UB[1].v[i] = UB[2].v[i] = 0.f; }
  auto step = [&](auto PH, int r) __attribute__((always_inline)) {
    constexpr int ph = decltype(PH)::value;
    constexpr int s0 = (ph + 1) % 3, s1 = (ph + 2) % 3, s2 = ph;
    A[s2] = load_win8<NARROW>(p1, r + 1, x0, H, W);
    Bw[s2] = load_win8<NARROW>(p2, r + 1, x0, H, W);
    UA[s2] = u_row(A[s0], A[s1], A[s2], wpa, r);
    UB[s2] = u_row(Bw[s0], Bw[s1], Bw[s2], wpb, r);
    if (u != nullptr && r >= y0 && r < yend) {     // u == nullptr: inference, nothing reads u back
      store_px4<NARROW>(u1, r, x0, W, f32x4{UA[s2].v[1], UA[s2].v[2], UA[s2].v[3], UA[s2].v[4]});
      store_px4<NARROW>(u2, r, x0, W, f32x4{UB[s2].v[1], UB[s2].v[2], UB[s2].v[3], UB[s2].v[4]});
    }
    const int q = r - 1;                         // gate row q needs u rows q-1, q, q+1
    if (q >= y0 && q < yend) {
      const f32x4 c1 = stencil(UA[s0], UA[s1], UA[s2], wa), c2 = stencil(UB[s0], UB[s1], UB[s2], wb);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (tanh_fast(c1[e]) + UA[s1].v[1 + e]) * (tanh_fast(c2[e]) + UB[s1].v[1 + e]);
      store_px4<NARROW>(gp, q, x0, W, o);
    }
  };
  for (int r = y0 - 1; r <= yend; r += 3) {
    step(std::integral_constant<int, 0>{}, r);
    if (r + 1 > yend) break;
    step(std::integral_constant<int, 1>{}, r + 1);
    if (r + 2 > yend) break;
    step(std::integral_constant<int, 2>{}, r + 2);
  }
}

// gw[c][t] partial over one block of strips of plane (b,c):  sum gout[y][x] * in[y+dy-1][x+dx-1]
template <bool NARROW>
__global__ __launch_bounds__(kThreads) void dw3x3_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ gout,
                                                               float* __restrict__ part, int H, int W, int nchunk, Tiling tl) {
  __shared__ float red[kThreads / 64];
  const long bc = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // item inside the plane
  const int strip = idx / tl.nx4;
  int dup;
  const int x0 = lane_x0<NARROW>(idx - strip * tl.nx4, W, dup), y0 = strip * tl.rows;
  const bool live = strip < tl.nstrips;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  if (live) {
    const long HW = (long)H * W;
    const float* ip = in + bc * HW;
    const float* gp = gout + bc * HW;
    Win6 r0 = load_win6<false, NARROW>(ip, y0 - 1, x0, H, W), r1 = load_win6<false, NARROW>(ip, y0, x0, H, W);
    const int yend = min(y0 + tl.rows, H);
    for (int y = y0; y < yend; ++y) {
      const Win6 r2 = load_win6<false, NARROW>(ip, y + 1, x0, H, W);
      const Win6 gr = load_win6<false, NARROW>(gp, y, x0, H, W);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = e < dup ? 0.f : gr.v[1 + e];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          acc[dx] += gv * r0.v[e + dx]; acc[3 + dx] += gv * r1.v[e + dx]; acc[6 + dx] += gv * r2.v[e + dx];
        }
      }
      r0 = r1; r1 = r2;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float s = block_sum(acc[t], red);
    if (threadIdx.x == 0) part[(bc * nchunk + blockIdx.x) * 9 + t] = s;
  }
}

// Fused depthwise backward: one pass over (gout, in) produces BOTH the data gradient
//   gin[y][x] = sum_tap w[8-tap] * gout[y+dy-1][x+dx-1]  (+ addend)
// and the per-block partial of the weight gradient gw[tap] = sum gout[y][x] * in[y+dy-1][x+dx-1]:
// gout is read once instead of twice (3 tensor passes instead of 4).
template <bool NARROW, class T>      // T: storage type of in, gout, addend, gin
__global__ __launch_bounds__(kThreads) void dw3x3_bwd_kernel(const T* __restrict__ in, const T* __restrict__ gout,
                                                             const float* __restrict__ w1, const float* __restrict__ w2, int csplit,
                                                             const T* __restrict__ addend, T* __restrict__ gin,
                                                             float* __restrict__ part, int C, int H, int W, int nchunk, Tiling tl) {
  __shared__ float red[kThreads / 64];
  const long bc = blockIdx.y;
  const long HW = (long)H * W;
  const T* ip = in + bc * HW;
  const T* gp = gout + bc * HW;
  const T* ap = addend ? addend + bc * HW : nullptr;
  T* op = gin + bc * HW;
  float w[9];
  load_w9(w1, w2, csplit, (int)(bc % C), true, w);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  // a block walks kSub consecutive 256-item groups of the plane so the nine block reductions below
  // are paid once per kSub groups
  for (int sub = 0; sub < kSub; ++sub) {
    const int idx = (blockIdx.x * kSub + sub) * blockDim.x + threadIdx.x;
    const int strip = idx / tl.nx4;
    int dup;
    const int x0 = lane_x0<NARROW>(idx - strip * tl.nx4, W, dup), y0 = strip * tl.rows;
    if (strip >= tl.nstrips) continue;
    Win6 i0 = load_win6<false, NARROW>(ip, y0 - 1, x0, H, W), i1 = load_win6<false, NARROW>(ip, y0, x0, H, W);
    Win6 g0 = load_win6<false, NARROW>(gp, y0 - 1, x0, H, W), g1 = load_win6<false, NARROW>(gp, y0, x0, H, W);
    const int yend = min(y0 + tl.rows, H);
    for (int y = y0; y < yend; ++y) {
      const Win6 i2 = load_win6<false, NARROW>(ip, y + 1, x0, H, W);
      const Win6 g2 = load_win6<false, NARROW>(gp, y + 1, x0, H, W);
      f32x4 o = stencil(g0, g1, g2, w);
      if (ap) {
        o += load_px4<NARROW>(ap, y, x0, W);
      }
      store_px4<NARROW>(op, y, x0, W, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = e < dup ? 0.f : g1.v[1 + e];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          acc[dx] += gv * i0.v[e + dx]; acc[3 + dx] += gv * i1.v[e + dx]; acc[6 + dx] += gv * i2.v[e + dx];
        }
      }
      i0 = i1; i1 = i2; g0 = g1; g1 = g2;
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float sres = block_sum(acc[t], red);
    if (threadIdx.x == 0) part[(bc * nchunk + blockIdx.x) * 9 + t] = sres;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused backward of the IEL gate and of dwconv1 / dwconv2 (net/LCA.py:62-64):
//   forward:  a_i = dw_i(u_i), t_i = tanh(a_i), s_i = t_i + u_i, g = s1 * s2        (i = 1, 2)
//   backward: ds_1 = dg * s2, da_1 = ds_1 * (1 - t1^2)   (and symmetrically for 2)
//             du_i = ds_i + dw_i^T(da_i),   gw_i[tap] = sum da_i[p] * u_i[p + tap]
// The unfused path wrote da, ds (4h channels) and read them back (plus u again) in a second kernel; here a
// lane recomputes da on its strip plus a one-pixel halo from an 8-wide register window of u, so the only
// HBM traffic is: read dg (h), read u (2h), write du (2h).  One lane = one channel pair, 4 px x `rows` rows.
struct GateRow {      // da, ds of one row on the 6-wide window x0-1 .. x0+4, for both channels of the pair
  float da1[6], da2[6], ds1[6], ds2[6];
};

__device__ __forceinline__ GateRow gate_bwd_row(const Win8& a0, const Win8& a1, const Win8& a2, const Win8& b0, const Win8& b1,
                                                const Win8& b2, const Win6& dg, const float* wa, const float* wb) {
  GateRow o;
  float c1v[6], c2v[6];                       // tap-major over the twelve chains (see stencil())
#pragma unroll
  for (int jx = 0; jx < 6; ++jx) { c1v[jx] = wa[0] * a0.v[jx]; c2v[jx] = wb[0] * b0.v[jx]; }
#pragma unroll
  for (int t = 1; t < 3; ++t)
#pragma unroll
    for (int jx = 0; jx < 6; ++jx) { c1v[jx] = fmaf(wa[t], a0.v[jx + t], c1v[jx]); c2v[jx] = fmaf(wb[t], b0.v[jx + t], c2v[jx]); }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int jx = 0; jx < 6; ++jx) { c1v[jx] = fmaf(wa[3 + t], a1.v[jx + t], c1v[jx]); c2v[jx] = fmaf(wb[3 + t], b1.v[jx + t], c2v[jx]); }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int jx = 0; jx < 6; ++jx) { c1v[jx] = fmaf(wa[6 + t], a2.v[jx + t], c1v[jx]); c2v[jx] = fmaf(wb[6 + t], b2.v[jx + t], c2v[jx]); }
#pragma unroll
  for (int jx = 0; jx < 6; ++jx) {
    const float c1 = c1v[jx], c2 = c2v[jx];
    const float t1 = tanh_fast(c1), t2 = tanh_fast(c2);
    const float s1 = t1 + a1.v[jx + 1], s2 = t2 + b1.v[jx + 1];
    const float g = dg.v[jx];
    o.ds1[jx] = g * s2; o.ds2[jx] = g * s1;
    o.da1[jx] = o.ds1[jx] * (1.f - t1 * t1); o.da2[jx] = o.ds2[jx] * (1.f - t2 * t2);
  }
  return o;
}

template <bool NARROW, class T>      // T: storage type of u, dg, du
__global__ __launch_bounds__(kThreads) void iel_gate_dw_bwd_kernel(const T* __restrict__ u, const float* __restrict__ w1,
                                                                   const float* __restrict__ w2, const T* __restrict__ dg,
                                                                   T* __restrict__ du, float* __restrict__ part, int h, int H,
                                                                   int W, int nchunk, Tiling tl) {
  __shared__ float red[kThreads / 64];
  const long bc = blockIdx.y;                    // b * h + c
  const int c = (int)(bc % h);
  const long b = bc / h;
  const long HW = (long)H * W;
  const T* p1 = u + (b * 2 * h + c) * HW;
  const T* p2 = u + (b * 2 * h + h + c) * HW;
  const T* gp = dg + bc * HW;
  T* o1 = du + (b * 2 * h + c) * HW;
  T* o2 = du + (b * 2 * h + h + c) * HW;
  float wa[9], wb[9], fa[9], fb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wa[t] = w1[(long)c * 9 + t]; wb[t] = w2[(long)c * 9 + t];
    fa[t] = w1[(long)c * 9 + 8 - t]; fb[t] = w2[(long)c * 9 + 8 - t];
  }
  float acc1[9], acc2[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { acc1[t] = 0.f; acc2[t] = 0.f; }

  for (int sub = 0; sub < kSub; ++sub) {
    const int idx = (blockIdx.x * kSub + sub) * blockDim.x + threadIdx.x;
    const int strip = idx / tl.nx4;
    int dup;
    const int x0 = lane_x0<NARROW>(idx - strip * tl.nx4, W, dup), y0 = strip * tl.rows;
    if (strip >= tl.nstrips) continue;
    const int yend = min(y0 + tl.rows, H);
    // u windows hold rows r-1, r, r+1 while row r of (da, ds) is being formed; the rows the NEXT iteration needs
    // (u row r+2, dg row r+1) are requested before this iteration's arithmetic, so with only two waves per SIMD
    // (222 VGPRs) the HBM latency still hides behind ~500 VALU ops
    // Three window slots per channel and three gate-row slots whose roles rotate with the row phase; the row loop is
    // unrolled by three so the slot indices are compile-time constants (the shifts of the rolling windows were 22 %
    // of the loop's instructions, and the kernel is VALU-issue-bound).  Only the prefetched rows (na, nb, ng: requested
    // one iteration ahead, before the arithmetic) are copied into their slot.
    Win8 A[3], Bw[3];
    GateRow G[3];
    A[1] = load_win8<NARROW>(p1, y0 - 2, x0, H, W); A[2] = load_win8<NARROW>(p1, y0 - 1, x0, H, W);
    Bw[1] = load_win8<NARROW>(p2, y0 - 2, x0, H, W); Bw[2] = load_win8<NARROW>(p2, y0 - 1, x0, H, W);
    Win8 na = load_win8<NARROW>(p1, y0, x0, H, W), nb = load_win8<NARROW>(p2, y0, x0, H, W);
    Win6 ng = load_win6<false, NARROW>(gp, y0 - 1, x0, H, W);
#pragma unroll
    for (int i = 0; i < 6; ++i) { G[1].da1[i] = G[1].da2[i] = G[1].ds1[i] = G[1].ds2[i] = 0.f; }
    G[2] = G[1];
    auto step = [&](auto PH, int r) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH)::value;
      constexpr int s0 = (ph + 1) % 3, s1 = (ph + 2) % 3, s2 = ph;     // rows r-1, r, r+1 of u; rows r-2, r-1, r of (da, ds)
      A[s2] = na; Bw[s2] = nb;
      const Win6 dgr = ng;                                     // zero outside the image => da = ds = 0 there
      na = load_win8<NARROW>(p1, r + 2, x0, H, W);
      nb = load_win8<NARROW>(p2, r + 2, x0, H, W);
      ng = load_win6<false, NARROW>(gp, r + 1, x0, H, W);
      G[s2] = gate_bwd_row(A[s0], A[s1], A[s2], Bw[s0], Bw[s1], Bw[s2], dgr, wa, wb);
      const GateRow &gm = G[s0], &gc = G[s1], &gn = G[s2];
      if (r >= y0 && r < yend) {                               // weight gradients: this lane's own pixels of row r
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d1 = e < dup ? 0.f : gn.da1[e + 1], d2 = e < dup ? 0.f : gn.da2[e + 1];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            acc1[dx] += d1 * A[s0].v[e + dx + 1]; acc1[3 + dx] += d1 * A[s1].v[e + dx + 1]; acc1[6 + dx] += d1 * A[s2].v[e + dx + 1];
            acc2[dx] += d2 * Bw[s0].v[e + dx + 1]; acc2[3 + dx] += d2 * Bw[s1].v[e + dx + 1]; acc2[6 + dx] += d2 * Bw[s2].v[e + dx + 1];
          }
        }
      }
      const int q = r - 1;                                     // du row q needs da rows q-1 (gm), q (gc), q+1 (gn)
      if (q >= y0 && q < yend) {
        f32x4 d1, d2;                         // tap-major over the eight chains (see stencil())
#pragma unroll
        for (int e = 0; e < 4; ++e) { d1[e] = fmaf(fa[0], gm.da1[e], gc.ds1[e + 1]); d2[e] = fmaf(fb[0], gm.da2[e], gc.ds2[e + 1]); }
#pragma unroll
        for (int t = 1; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { d1[e] = fmaf(fa[t], gm.da1[e + t], d1[e]); d2[e] = fmaf(fb[t], gm.da2[e + t], d2[e]); }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { d1[e] = fmaf(fa[3 + t], gc.da1[e + t], d1[e]); d2[e] = fmaf(fb[3 + t], gc.da2[e + t], d2[e]); }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { d1[e] = fmaf(fa[6 + t], gn.da1[e + t], d1[e]); d2[e] = fmaf(fb[6 + t], gn.da2[e + t], d2[e]); }
        store_px4<NARROW>(o1, q, x0, W, d1);
        store_px4<NARROW>(o2, q, x0, W, d2);
      }
    };
    for (int r = y0 - 1; r <= yend; r += 3) {
      step(std::integral_constant<int, 0>{}, r);
      if (r + 1 > yend) break;
      step(std::integral_constant<int, 1>{}, r + 1);
      if (r + 2 > yend) break;
      step(std::integral_constant<int, 2>{}, r + 2);
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float s1 = block_sum(acc1[t], red);
    const float s2 = block_sum(acc2[t], red);
    if (threadIdx.x == 0) {
      part[((bc * nchunk + blockIdx.x) * 2) * 9 + t] = s1;
      part[((bc * nchunk + blockIdx.x) * 2 + 1) * 9 + t] = s2;
    }
  }
}

// gw1[c][t], gw2[c][t] = sum_b sum_chunk part[((b*h + c)*nchunk + chunk)*2 + {0,1}][t]
__global__ void gate_wgrad_reduce_kernel(const float* __restrict__ part, int B, int h, int nchunk, float* __restrict__ gw1,
                                         float* __restrict__ gw2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * 18) return;
  const int c = i / 18, r = i - c * 18, which = r / 9, t = r - which * 9;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < nchunk; ++k) s += part[((((long)b * h + c) * nchunk + k) * 2 + which) * 9 + t];
  (which ? gw2 : gw1)[(long)c * 9 + t] = s;
}

// gw[c][t] (+)= sum_b sum_chunk part[((b*C + c)*nchunk + chunk)*9 + t]
__global__ void dw_wgrad_reduce_kernel(const float* __restrict__ part, int B, int C, int nchunk, float* __restrict__ gw1,
                                       float* __restrict__ gw2, int csplit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 9) return;
  const int c = i / 9, t = i - c * 9;
  float s = 0.f;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < nchunk; ++k) s += part[(((long)b * C + c) * nchunk + k) * 9 + t];
  if (c < csplit) gw1[(long)c * 9 + t] = s; else gw2[(long)(c - csplit) * 9 + t] = s;
}

// tilings of the kernel families (halo rows re-read per strip: 2 for a stencil, 3 for the fused gate pass;
// min_lanes = two rounds of the lanes 256 CUs keep resident at each kernel's register footprint)
inline Tiling fwd_tiling(long planes, int H, int W) { return pick_tiling(planes, H, W, 2, 256L * 1792 * 2, false); }
inline Tiling dw_gate_tiling(long planes, int H, int W) { return pick_tiling(planes, H, W, 4, 256L * 1024 * 2, false); }
inline Tiling wgrad_tiling(long planes, int H, int W) { return pick_tiling(planes, H, W, 2, 256L * 1280 * 2, true); }
inline Tiling gate_bwd_tiling(long planes, int H, int W) { return pick_tiling(planes, H, W, 3, 256L * 512 * 2, true); }

// block size of a block-per-plane kernel: whole waves, no more than the plane has items
inline int plane_threads(Tiling tl) {
  const long items = (long)tl.nstrips * tl.nx4;
  return items >= kThreads ? kThreads : (int)(((items + 63) / 64) * 64);
}

inline int chunks_of(Tiling tl) {
  const int th = plane_threads(tl);
  return (int)(((long)tl.nstrips * tl.nx4 + th - 1) / th);
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

// launch KERNEL<..., NARROW> with NARROW = (W < 8): the generic per-element loaders only exist in that instantiation
#define CIDNET_LAUNCH_NW(W_, KERNEL_T, KERNEL_F, ...)                    \
  do {                                                                   \
    if ((W_) < 8) hipLaunchKernelGGL(KERNEL_T, __VA_ARGS__);             \
    else hipLaunchKernelGGL(KERNEL_F, __VA_ARGS__);                      \
  } while (0)

extern "C" {

int cidnet_dw3x3_bwd_t(const void* in, const void* gout, const float* w1, const float* w2, int csplit, const void* addend,
                       void* gin, int dt, float* gw1, float* gw2, float* ws, long ws_floats, int B, int C, int H, int W,
                       void* stream);
int cidnet_iel_gate_dw_bwd_t(const void* u, const float* w1, const float* w2, const void* dg, void* du, int dt, float* gw1,
                             float* gw2, float* ws, long ws_floats, int B, int h, int H, int W, void* stream);

#ifdef CIDNET_DEBUG
void cidnet_debug_dw_rows(int rows) { g_dw_force_rows = rows; }
#endif

int cidnet_dw3x3(const float* in, const float* w1, const float* w2, int csplit, const float* addend, float* out, int flip,
                 int B, int C, int H, int W, void* stream) {
  return cidnet_dw3x3_t(in, w1, w2, csplit, addend, out, CIDNET_F32, flip, B, C, H, W, stream);
}

/* in / addend / out stored as fp32 or bf16 (one type, dt) */
int cidnet_dw3x3_t(const void* in, const float* w1, const float* w2, int csplit, const void* addend, void* out, int dt, int flip,
                   int B, int C, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(in && w1 && out && B > 0 && C > 0 && H > 0 && W > 0 && (dt | 1) == 1);
  CIDNET_CHECK_ARG(csplit >= C || w2);
  const Tiling tl = fwd_tiling((long)B * C, H, W);
  const long items = n_items((long)B * C, tl);
  const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
  if (dt)
    CIDNET_LAUNCH_NW(W, (dw3x3_kernel<true, bf16_t>), (dw3x3_kernel<false, bf16_t>), grid, dim3(kThreads), 0, (hipStream_t)stream,
                     (const bf16_t*)in, w1, w2, csplit, (const bf16_t*)addend, (bf16_t*)out, flip, B, C, H, W, tl);
  else
    CIDNET_LAUNCH_NW(W, (dw3x3_kernel<true, float>), (dw3x3_kernel<false, float>), grid, dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)in, w1, w2, csplit, (const float*)addend, (float*)out, flip, B, C, H, W, tl);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_iel_gate_fwd(const float* u, const float* w1, const float* w2, float* g, int B, int h, int H, int W,
                        void* stream) {
  CIDNET_CHECK_ARG(u && w1 && w2 && g && B > 0 && h > 0 && H > 0 && W > 0);
  const Tiling tl = fwd_tiling((long)B * h, H, W);
  const long items = n_items((long)B * h, tl);
  CIDNET_LAUNCH_NW(W, (iel_gate_kernel<0, true>), (iel_gate_kernel<0, false>), dim3((unsigned)((items + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, u, w1, w2, (const float*)nullptr, g, (float*)nullptr, (float*)nullptr, B, h, H, W, tl);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_iel_dw_gate_fwd_t(const void* pin, const float* wdw, const float* w1, const float* w2, void* u, void* g, int dt, int B,
                             int h, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(pin && wdw && w1 && w2 && g && B > 0 && h > 0 && H > 0 && W > 0 && (dt | 1) == 1);   // u may be NULL (not stored)
  const Tiling tl = dw_gate_tiling((long)B * h, H, W);
  const long items = n_items((long)B * h, tl);
  const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
  if (dt)
    CIDNET_LAUNCH_NW(W, (iel_dw_gate_kernel<true, bf16_t>), (iel_dw_gate_kernel<false, bf16_t>), grid, dim3(kThreads), 0,
                     (hipStream_t)stream, (const bf16_t*)pin, wdw, w1, w2, (bf16_t*)u, (bf16_t*)g, B, h, H, W, tl);
  else
    CIDNET_LAUNCH_NW(W, (iel_dw_gate_kernel<true, float>), (iel_dw_gate_kernel<false, float>), grid, dim3(kThreads), 0,
                     (hipStream_t)stream, (const float*)pin, wdw, w1, w2, (float*)u, (float*)g, B, h, H, W, tl);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_iel_dw_gate_fwd(const float* pin, const float* wdw, const float* w1, const float* w2, float* u, float* g, int B, int h,
                           int H, int W, void* stream) {
  return cidnet_iel_dw_gate_fwd_t(pin, wdw, w1, w2, u, g, 0, B, h, H, W, stream);
}

int cidnet_iel_gate_bwd(const float* u, const float* w1, const float* w2, const float* dg, float* da, float* ds, int B,
                        int h, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(u && w1 && w2 && dg && da && ds && B > 0 && h > 0 && H > 0 && W > 0);
  const Tiling tl = fwd_tiling((long)B * h, H, W);
  const long items = n_items((long)B * h, tl);
  CIDNET_LAUNCH_NW(W, (iel_gate_kernel<1, true>), (iel_gate_kernel<1, false>), dim3((unsigned)((items + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, u, w1, w2, dg, (float*)nullptr, da, ds, B, h, H, W, tl);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_dw3x3_wgrad_ws_floats(int B, int C, int H, int W) {
  return (long)B * C * chunks_of(wgrad_tiling((long)B * C, H, W)) * 9;
}

int cidnet_dw3x3_wgrad(const float* in, const float* gout, float* gw1, float* gw2, int csplit, float* ws, long ws_floats,
                       int B, int C, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(in && gout && gw1 && ws && B > 0 && C > 0 && H > 0 && W > 0);
  CIDNET_CHECK_ARG(csplit >= C || gw2);
  if (ws_floats < cidnet_dw3x3_wgrad_ws_floats(B, C, H, W)) return CIDNET_ERR_WS;
  const Tiling tl = wgrad_tiling((long)B * C, H, W);
  const int nchunk = chunks_of(tl);
  CIDNET_LAUNCH_NW(W, (dw3x3_wgrad_kernel<true>), (dw3x3_wgrad_kernel<false>), dim3((unsigned)nchunk, (unsigned)(B * C)), dim3(plane_threads(tl)), 0, (hipStream_t)stream,
                     in, gout, ws, H, W, nchunk, tl);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((unsigned)((C * 9 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, B,
                     C, nchunk, gw1, gw2, csplit);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

/* data gradient and weight gradient of a depthwise 3x3 in one pass: gin = dw3x3(gout, flipped taps)
 * [+ addend]; gw = sum gout * shifted(in).  ws as cidnet_dw3x3_wgrad. */
int cidnet_dw3x3_bwd(const float* in, const float* gout, const float* w1, const float* w2, int csplit, const float* addend,
                     float* gin, float* gw1, float* gw2, float* ws, long ws_floats, int B, int C, int H, int W, void* stream) {
  return cidnet_dw3x3_bwd_t(in, gout, w1, w2, csplit, addend, gin, 0, gw1, gw2, ws, ws_floats, B, C, H, W, stream);
}

int cidnet_dw3x3_bwd_t(const void* in, const void* gout, const float* w1, const float* w2, int csplit, const void* addend,
                       void* gin, int dt, float* gw1, float* gw2, float* ws, long ws_floats, int B, int C, int H, int W,
                       void* stream) {
  CIDNET_CHECK_ARG(in && gout && w1 && gin && gw1 && ws && B > 0 && C > 0 && H > 0 && W > 0 && (dt | 1) == 1);
  CIDNET_CHECK_ARG(csplit >= C || (w2 && gw2));
  if (ws_floats < cidnet_dw3x3_wgrad_ws_floats(B, C, H, W)) return CIDNET_ERR_WS;
  const Tiling tl = wgrad_tiling((long)B * C, H, W);
  const int nchunk = (chunks_of(tl) + kSub - 1) / kSub;
  if (dt)
    CIDNET_LAUNCH_NW(W, (dw3x3_bwd_kernel<true, bf16_t>), (dw3x3_bwd_kernel<false, bf16_t>), dim3((unsigned)nchunk, (unsigned)(B * C)),
                     dim3(plane_threads(tl)), 0, (hipStream_t)stream, (const bf16_t*)in, (const bf16_t*)gout, w1, w2, csplit,
                     (const bf16_t*)addend, (bf16_t*)gin, ws, C, H, W, nchunk, tl);
  else
    CIDNET_LAUNCH_NW(W, (dw3x3_bwd_kernel<true, float>), (dw3x3_bwd_kernel<false, float>), dim3((unsigned)nchunk, (unsigned)(B * C)),
                     dim3(plane_threads(tl)), 0, (hipStream_t)stream, (const float*)in, (const float*)gout, w1, w2, csplit,
                     (const float*)addend, (float*)gin, ws, C, H, W, nchunk, tl);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3((unsigned)((C * 9 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, B, C,
                     nchunk, gw1, gw2, csplit);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_iel_gate_dw_bwd_ws_floats(int B, int h, int H, int W) {
  return (long)B * h * ((chunks_of(gate_bwd_tiling((long)B * h, H, W)) + kSub - 1) / kSub) * 18;
}

/* du = d(loss)/d(u) of the gate INCLUDING the backward of dwconv1/dwconv2, and their weight gradients, in one
 * pass: reads dg (B,h,H,W) and u (B,2h,H,W), writes du (B,2h,H,W), gw1/gw2 (h,1,3,3). */
int cidnet_iel_gate_dw_bwd(const float* u, const float* w1, const float* w2, const float* dg, float* du, float* gw1, float* gw2,
                           float* ws, long ws_floats, int B, int h, int H, int W, void* stream) {
  return cidnet_iel_gate_dw_bwd_t(u, w1, w2, dg, du, 0, gw1, gw2, ws, ws_floats, B, h, H, W, stream);
}

int cidnet_iel_gate_dw_bwd_t(const void* u, const float* w1, const float* w2, const void* dg, void* du, int dt, float* gw1,
                             float* gw2, float* ws, long ws_floats, int B, int h, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(u && w1 && w2 && dg && du && gw1 && gw2 && ws && B > 0 && h > 0 && H > 0 && W > 0 && (dt | 1) == 1);
  if (ws_floats < cidnet_iel_gate_dw_bwd_ws_floats(B, h, H, W)) return CIDNET_ERR_WS;
  const Tiling tl = gate_bwd_tiling((long)B * h, H, W);
  const int nchunk = (chunks_of(tl) + kSub - 1) / kSub;
  if (dt)
    CIDNET_LAUNCH_NW(W, (iel_gate_dw_bwd_kernel<true, bf16_t>), (iel_gate_dw_bwd_kernel<false, bf16_t>), dim3((unsigned)nchunk, (unsigned)(B * h)),
                     dim3(plane_threads(tl)), 0, (hipStream_t)stream, (const bf16_t*)u, w1, w2, (const bf16_t*)dg, (bf16_t*)du, ws, h, H,
                     W, nchunk, tl);
  else
    CIDNET_LAUNCH_NW(W, (iel_gate_dw_bwd_kernel<true, float>), (iel_gate_dw_bwd_kernel<false, float>), dim3((unsigned)nchunk, (unsigned)(B * h)),
                     dim3(plane_threads(tl)), 0, (hipStream_t)stream, (const float*)u, w1, w2, (const float*)dg, (float*)du, ws, h, H,
                     W, nchunk, tl);
  CIDNET_LAUNCH_STATUS();
  hipLaunchKernelGGL(gate_wgrad_reduce_kernel, dim3((unsigned)((h * 18 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, B, h,
                     nchunk, gw1, gw2);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
