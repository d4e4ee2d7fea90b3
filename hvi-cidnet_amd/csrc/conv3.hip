// K9/K10/K11: dense 3x3 convolution (pad 1: zero or replicate) as an implicit GEMM on the fp32
// MFMA (v_mfma_f32_16x16x4_f32) -- forward, data gradient (same kernel: flipped taps, transposed
// weight strides), weight gradient, and the border correction of the replicate-pad data gradient.
// Reference call sites: net/transformer_utils.py:39,58 (zero pad) and net/CIDNet.py:21-24,32-35,
// 39-42,50-53 (ReplicationPad2d(1) + valid conv).
//
// Forward mapping (no activation goes through LDS):
//   lane l = (c = l&15, j = l>>4).  For a k-group of 4 input channels (channel = group*4 + j) and an
//   input row, the lane loads the float4 X[ch][row][x0+4c .. +3]; its left/right neighbours come from
//   lanes c-1 / c+1 by DPP row shifts (lanes 0 / 15 fetch the tile halo).  Element e+dx of that
//   6-wide window is the B operand (k = j, col = c) of the MFMA for tap (dy,dx) and pixel slot e, so
//   the accumulators e = 0..3 of an output channel hold 4 consecutive pixels: float4 stores.
//   One loaded input row feeds up to 3 output rows x 3 dx x 4 slots x MT channel tiles of MFMAs.
//   Output channels that do not fill a 16-row tile: up to two groups of 4 rows per block go through
//   v_mfma_f32_4x4x1_16b_f32 instead of a padded 16x16x4 tile (M = 36 = 2 tiles + 1 group: 72 instead of 96 MFMA
//   cycles per tap).  Its 16 blocks are (k-slot j) x (4 lanes): lane l feeds its own window element as B and
//   A[32 + (l&3)][k = j], so block l>>2 accumulates the k = j (mod 4) share of 4 rows x the lane's own pixels;
//   the four j-shares are added across lanes (xor 16, 32) once per tile, and lane j stores row 32 + j.
//   Weights (A operand) are staged per 16-channel chunk in LDS as [ci][tap][co] with a leading
//   dimension == 16 (mod 32): conflict-free ds_read_b32.
#include <type_traits>
#include "common.h"
#include "conv3_thin.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kR = 2;      // output rows per wave
constexpr int kKC = 36;    // input channels staged per chunk (9 k-groups)
constexpr int kStageBatch = 6;   // 16-byte weight loads a thread keeps in flight while staging the panel
#ifdef CIDNET_DEBUG
int g_c3_dbg = 0;          // timing-study switches (cidnet_debug_c3_flags); the shipped library has no mutable state
#else
constexpr int g_c3_dbg = 0;
#endif

struct C3Args {
  const float* X; long x_bs;
  const float* Wt; long w_ms, w_ks;   // A[m][k][tap] = Wt[m*w_ms + k*w_ks + tap']
  float* Y; long y_bs;
  const float* R; long r_bs;          // optional addend of the output's shape (Y = conv + R), or nullptr
  int M, K, H, W;
  int flip;        // tap' = 8 - tap (data gradient)
  int replicate;   // border mode of the input
  int nmb;         // number of m-blocks (blockIdx.z = b*nmb + mb)
  int tpb;         // row tiles per block (launcher)
  int dbg;         // timing-study switches (0 in production): 1 no stores, 2 no window loads, 4 no LDS weight reads
};

struct Win6 {
  float v[6];
};
struct __attribute__((packed, aligned(4))) f2u { float x, y; };

// window x0-1 .. x0+4 of input row yy (zero or replicate outside the image).  Interior lanes issue
// one 16 B and one 8 B load; only lanes touching the left/right image border take the scalar path.
__device__ __forceinline__ Win6 load_win(const float* __restrict__ plane, int yy, int x0, int H, int W, bool replicate) {
  Win6 r;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = 0.f;
  if (yy < 0 || yy >= H) {
    if (!replicate) return r;
    yy = yy < 0 ? 0 : H - 1;
  }
  const float* row = plane + (long)yy * W;
  if (x0 >= 1 && x0 + 5 <= W) {
    const f4u a = *reinterpret_cast<const f4u*>(row + x0 - 1);
    const f2u b = *reinterpret_cast<const f2u*>(row + x0 + 3);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y;
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      int x = x0 - 1 + i;
      if (x < 0) { if (!replicate) continue; x = 0; }
      if (x >= W) { if (!replicate) continue; x = W - 1; }
      r.v[i] = row[x];
    }
  }
  return r;
}

// Straight-line window loads for W >= 8 in two steps (see the weight-gradient kernel below for why): raw_win only
// ISSUES the loads, at addresses clamped into the row; finish_win rebuilds the window of a border lane with selects.
// A lane whose 4 pixels cross the right border is pulled back to W-4 (it recomputes pixels its neighbour also owns
// and stores identical values), so every lane loads and stores whole vectors.
struct RawWin6 {
  f4u a;
  f2u b;
};

__device__ __forceinline__ RawWin6 raw_win(const float* __restrict__ plane, int yy, int xq, int H, int W) {
  const float* row = plane + (long)min(max(yy, 0), H - 1) * W;
  RawWin6 r;
  r.a = *reinterpret_cast<const f4u*>(row + max(xq - 1, 0));
  r.b = *reinterpret_cast<const f2u*>(row + min(xq + 3, W - 2));
  return r;
}

__device__ __forceinline__ Win6 finish_win(const RawWin6& w, int yy, int xq, int H, int W, bool replicate) {
  const bool left = xq == 0, right = xq + 4 >= W;
  Win6 r;
  r.v[0] = left ? (replicate ? w.a.x : 0.f) : w.a.x;
  r.v[1] = left ? w.a.x : w.a.y;
  r.v[2] = left ? w.a.y : w.a.z;
  r.v[3] = left ? w.a.z : w.a.w;
  r.v[4] = right ? w.b.y : w.b.x;
  r.v[5] = right ? (replicate ? w.b.y : 0.f) : w.b.y;
  const bool ok = replicate || (yy >= 0 && yy < H);
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = ok ? r.v[i] : 0.f;
  return r;
}

// LOGX: log2 of the lanes (x 4 pixels) a 16-lane group spends on x; the other 16>>LOGX lanes take
// further rows, so narrow images (W = 75, 150) do not waste most of a 64-pixel-wide tile.
// The weight panel sits in LDS in the ORDER OF ITS SOURCE ROWS, so staging is a straight 16-byte copy (a scatter into a
// [k][tap][m] image cost 48 kcycles per block: every lane of a store hit the same bank).  Element (m, k, tap) is at
// m*sm + k*sk + tap with
//   forward layout  Wt[m][k][tap] (w_ks == 9): one LDS row per output channel, sm = c3_ldm(k-extent*9), sk = 9;
//   data-gradient   Wt[k][m][tap] (w_ms == 9): one LDS row per input channel,  sm = 9, sk = c3_ldk(MB*9).
// The row strides make the A-operand reads (16 lanes over m, 4 lane groups over k) conflict-free: banks 2c + 9j with
// the stride 2 (mod 32), banks 9c + 16j with the stride 16 (mod 32), distinct inside each 32-lane half.
constexpr int c3_ldm(int rowlen) { return ((rowlen + 29) / 32) * 32 + 2; }    // smallest >= rowlen that is 2 (mod 32)
constexpr int c3_ldk(int rowlen) { return ((rowlen + 15) / 32) * 32 + 16; }   // smallest >= rowlen that is 16 (mod 32)
constexpr int c3_lds_floats(int mb, int kq) {
  const int f = mb * c3_ldm(kq * 9), d = kq * c3_ldk(mb * 9);
  return f > d ? f : d;
}

__device__ __forceinline__ float pick4(f32x4 v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }

// Phase timing study (build with -DC3_TIMING, tools/c3_phases.py): wave 0 of every block adds up the shader-clock
// cycles it spends staging weights, in the k loops and in the epilogues.  Not compiled into the product library.
#ifdef C3_TIMING
__device__ unsigned long long g_c3_phase[4 * 8192];
#define C3_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define C3_T(var)
#endif

// ADD: the epilogue adds a.R (a separate instantiation: folded into the plain kernel, the addend loads cost the
// 48-row variant its second wave per SIMD)
template <int MT, int LEFT, int LOGX, bool NARROW, bool ADD>
__global__ __launch_bounds__(kThreads) void conv3_kernel(C3Args a) {
  extern __shared__ float As[];                 // weight panel of one k chunk, source-row order (see c3_ldm)
  constexpr int MB = 16 * MT + 4 * LEFT;        // output channels of one block
  constexpr int LG = LEFT > 0 ? LEFT : 1;       // array extent (LEFT = 0: unused)
  constexpr int XL = 1 << LOGX, NY = 16 >> LOGX;
  constexpr int TROWS = 4 * NY * kR;            // output rows of one block tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, j = lane >> 4;
  const int b = blockIdx.z / a.nmb, mb = blockIdx.z - b * a.nmb;
  const int m0 = mb * MB;
  const int x0 = blockIdx.x * (XL * 4) + 4 * (c & (XL - 1));
  const int H = a.H, W = a.W;
  const int xq = NARROW ? x0 : min(x0, W - 4);    // NARROW (W < 8): per-element border path, its own instantiation
  const long HW = (long)H * W;
  const float* Xb = a.X + (long)b * a.x_bs;
  const bool rep = a.replicate != 0;
  const bool single = a.K <= kKC;               // whole weight panel resident: walk `tpb` row tiles

  // Weight panel -> LDS.  Consecutive threads walk the contiguous axis of the weight tensor ((k,tap) for the forward
  // layout, (m,tap) for the data-gradient layout) with 16 B loads: a thread takes 4 consecutive elements of a row,
  // decodes them with constant divisors and scatters them to [k][tap][m].  (One 4 B load + two run-time integer
  // divisions per element made this staging cost half a tile's MFMA time.)
  const bool fwd_layout = a.w_ks == 9 || a.w_ms != 9;      // generic strides are staged element-wise into the forward image
  const int kq_max = ((a.K < kKC ? a.K : kKC) + 3) & ~3;
  const int sm = fwd_layout ? c3_ldm(kq_max * 9) : 9;
  const int sk = fwd_layout ? 9 : c3_ldk(MB * 9);
  auto stage = [&](int kc0, int kcn, int ng) {
    const int kq = ng * 4;
    if (a.w_ks != 9 && a.w_ms != 9) {             // generic strides: element-wise
      for (int i = tid; i < kq * MB * 9; i += kThreads) {
        const int kk = i / (MB * 9), rem = i - kk * (MB * 9);
        const int mm = rem / 9, t = rem - mm * 9;
        float v = 0.f;
        if (kk < kcn && m0 + mm < a.M) v = a.Wt[(long)(m0 + mm) * a.w_ms + (long)(kc0 + kk) * a.w_ks + t];
        As[mm * sm + kk * sk + t] = v;
      }
      return;
    }
    // nrows rows of rowlen floats: a valid prefix copied from global, the rest (and whole invalid rows) zero
    const int nrows = fwd_layout ? MB : kq;
    const int rowlen = fwd_layout ? kq * 9 : MB * 9;
    const int valid = fwd_layout ? kcn * 9 : (a.M - m0 < MB ? a.M - m0 : MB) * 9;
    const int rows_ok = fwd_layout ? (a.M - m0 < MB ? a.M - m0 : MB) : kcn;
    const long gs = fwd_layout ? a.w_ms : a.w_ks;
    const int ls = fwd_layout ? sm : sk;
    const float* base = fwd_layout ? a.Wt + (long)m0 * a.w_ms + (long)kc0 * 9 : a.Wt + (long)m0 * 9 + (long)kc0 * a.w_ks;
    // 256 threads = 8 rows x 32 float4 lanes per pass (no run-time division anywhere: the two per element of a flat
    // index were most of what was left of the staging time); kStageBatch row groups are loaded before the first LDS
    // write, so a pass over the panel pays the L2 round trip once
    // Every resident block wants the same panel at the same moment: read in the same order, all CUs queue on one L2
    // channel at a time (13-27 kcycles for 51 KB).  Each block therefore starts at its own row.
    const int nv = (rowlen + 3) >> 2;
    const int rsub = tid >> 5, vl = tid & 31;
    const int rot = (int)((blockIdx.x * 5u + blockIdx.y * 3u + blockIdx.z * 7u) % (unsigned)nrows);
    for (int v = vl; v < nv; v += 32) {
      const int e0 = 4 * v;
      for (int r0 = rsub; r0 < nrows; r0 += 8 * kStageBatch) {
        f32x4 q[kStageBatch];
        int rows[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) {
          int row = r0 + 8 * u + rot;
          if (row >= nrows) row -= nrows;
          if (r0 + 8 * u >= nrows) row = nrows;   // past the panel: neither loaded nor stored
          rows[u] = row;
          q[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (row < rows_ok) {
            const float* src = base + (long)row * gs + e0;
            if (e0 + 3 < valid) {
              q[u] = load4u(src);
            } else {
#pragma unroll
              for (int z = 0; z < 4; ++z)
                if (e0 + z < valid) q[u][z] = src[z];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) {
          const int row = rows[u];
          if (row < nrows) {
            float* dst = As + row * ls + e0;      // 8-byte aligned (both strides are even); rows are padded past rowlen
            *reinterpret_cast<float2*>(dst) = float2{q[u][0], q[u][1]};
            *reinterpret_cast<float2*>(dst + 2) = float2{q[u][2], q[u][3]};
          }
        }
      }
    }
  };
  C3_T(ts0);
  if (single) {
    stage(0, a.K, (a.K + 3) >> 2);
    __syncthreads();
  }
  C3_T(ts1);
#ifdef C3_TIMING
  unsigned long long t_k = 0, t_e = 0;
#endif

  const int ntiles = (H + TROWS - 1) / TROWS;
  const int tile_end = min((int)(blockIdx.y + 1) * a.tpb, ntiles);
  for (int tile = blockIdx.y * a.tpb; tile < tile_end; ++tile) {
    const int ywave = tile * TROWS + wave * (NY * kR);
    const int yw = ywave + (c >> LOGX) * kR;    // first output row of this lane
    const bool wave_live = ywave < H;

    C3_T(tt0);
    f32x4 acc[kR][MT][4];
    f32x4 accl[kR][LG][4];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int lg = 0; lg < LG; ++lg)
#pragma unroll
        for (int e = 0; e < 4; ++e) accl[r][lg][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kc0 = 0; kc0 < a.K; kc0 += kKC) {
      const int kcn = min(kKC, a.K - kc0);
      const int ng = (kcn + 3) >> 2;
      if (!single) {
        __syncthreads();
        stage(kc0, kcn, ng);
        __syncthreads();
      }
      if (!wave_live) continue;                  // wave-uniform; the wave still joins the barriers above
      // channels past K are clamped to a valid plane: their weights are zero in LDS
      Win6 nxt[kR + 2];                          // NARROW: finished windows of the next k-group
      RawWin6 raw[kR + 2];                       // otherwise: its loads, as issued
      {
        const float* plane = Xb + (long)min(kc0 + j, a.K - 1) * HW;
#pragma unroll
        for (int iy = 0; iy < kR + 2; ++iy) {
          if (NARROW) nxt[iy] = load_win(plane, yw - 1 + iy, x0, H, W, rep);
          else raw[iy] = raw_win(plane, yw - 1 + iy, xq, H, W);
        }
      }
      for (int g = 0; g < ng; ++g) {
        Win6 win[kR + 2];
#pragma unroll
        for (int iy = 0; iy < kR + 2; ++iy) win[iy] = NARROW ? nxt[iy] : finish_win(raw[iy], yw - 1 + iy, xq, H, W, rep);
        if (g + 1 < ng && !(a.dbg & 2)) {
          const float* plane = Xb + (long)min(kc0 + 4 * (g + 1) + j, a.K - 1) * HW;
#pragma unroll
          for (int iy = 0; iy < kR + 2; ++iy) {
            if (NARROW) nxt[iy] = load_win(plane, yw - 1 + iy, x0, H, W, rep);
            else raw[iy] = raw_win(plane, yw - 1 + iy, xq, H, W);
          }
        }
        float av[9][MT];
        float al[9][LG];
        {
          const float* Ak = As + ((a.dbg & 4) ? 0 : g * 4 + j) * sk;
          auto fetch = [&](auto FLIP) __attribute__((always_inline)) {
            constexpr bool flip = decltype(FLIP)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const float* pa = Ak + (mt * 16 + c) * sm;
#pragma unroll
              for (int t = 0; t < 9; ++t) av[t][mt] = pa[flip ? 8 - t : t];
            }
            if (LEFT > 0) {
#pragma unroll
              for (int lg = 0; lg < LEFT; ++lg) {
                const float* pl = Ak + (MT * 16 + lg * 4 + (lane & 3)) * sm;
#pragma unroll
                for (int t = 0; t < 9; ++t) al[t][lg] = pl[flip ? 8 - t : t];
              }
            }
          };
          if (a.flip) fetch(std::true_type{}); else fetch(std::false_type{});
        }
        if (!NARROW) __builtin_amdgcn_sched_barrier(0);     // keep the next group's loads ahead of this group's MFMA burst
#pragma unroll
        for (int iy = 0; iy < kR + 2; ++iy)
#pragma unroll
          for (int r = 0; r < kR; ++r) {
            const int dy = iy - r;
            if (dy < 0 || dy > 2) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  acc[r][mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[dy * 3 + dx][mt], win[iy].v[e + dx], acc[r][mt][e], 0, 0, 0);
              if (LEFT > 0) {
#pragma unroll
                for (int lg = 0; lg < LEFT; ++lg)
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    accl[r][lg][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(al[dy * 3 + dx][lg], win[iy].v[e + dx], accl[r][lg][e], 0, 0, 0);
              }
            }
          }
        // ... and keep the NEXT group's window finishing (which waits for those loads) behind the burst: without this
        // fence the scheduler pulls the selects, and with them the s_waitcnt, up between the first MFMAs
        if (!NARROW) __builtin_amdgcn_sched_barrier(0);
      }
    }

#ifdef C3_TIMING
    const unsigned long long tt1 = __builtin_amdgcn_s_memtime();
    t_k += tt1 - tt0;
    struct C3TileEnd {                            // runs on every way out of the tile body (continue included)
      unsigned long long t1; unsigned long long& acc;
      __device__ ~C3TileEnd() { acc += __builtin_amdgcn_s_memtime() - t1; }
    } c3_tile_end{tt1, t_e};
#endif
    if (LEFT > 0 && wave_live) {                  // add the four k-slot shares of the 4-row groups (all lanes take part)
#pragma unroll
      for (int r = 0; r < kR; ++r)
#pragma unroll
        for (int lg = 0; lg < LEFT; ++lg)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float v = accl[r][lg][e][q];
              v += __shfl_xor(v, 16);
              v += __shfl_xor(v, 32);
              accl[r][lg][e][q] = v;
            }
    }
    if (x0 >= W || !wave_live) continue;
    if (a.dbg & 1) {
#pragma unroll
      for (int r = 0; r < kR; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(acc[r][mt][e]));
#pragma unroll
      for (int r = 0; r < kR; ++r)
#pragma unroll
        for (int lg = 0; lg < LEFT; ++lg)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(accl[r][lg][e]));
      continue;
    }
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const int y = yw + r;
      if (y >= H) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int m = m0 + mt * 16 + j * 4 + reg;
          if (m >= a.M) continue;
          const long off = (long)m * HW + (long)y * W;
          float* row = a.Y + (long)b * a.y_bs + off;
          f32x4 v = {acc[r][mt][0][reg], acc[r][mt][1][reg], acc[r][mt][2][reg], acc[r][mt][3][reg]};
          if (!NARROW) {
            if (ADD) v += load4u(a.R + (long)b * a.r_bs + off + xq);
            store4u(row + xq, v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (x0 + e < W) row[x0 + e] = v[e] + (ADD ? a.R[(long)b * a.r_bs + off + x0 + e] : 0.f);
          }
        }
#pragma unroll
      for (int lg = 0; lg < LEFT; ++lg) {          // lane (c, j) stores row 16*MT + 4*lg + j of the 4-row group
        const int m = m0 + MT * 16 + lg * 4 + j;
        if (m >= a.M) continue;
        const long off = (long)m * HW + (long)y * W;
        float* row = a.Y + (long)b * a.y_bs + off;
        f32x4 v = {pick4(accl[r][lg][0], j), pick4(accl[r][lg][1], j), pick4(accl[r][lg][2], j), pick4(accl[r][lg][3], j)};
        if (!NARROW) {
          if (ADD) v += load4u(a.R + (long)b * a.r_bs + off + xq);
          store4u(row + xq, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (x0 + e < W) row[x0 + e] = v[e] + (ADD ? a.R[(long)b * a.r_bs + off + x0 + e] : 0.f);
        }
      }
    }
  }
#ifdef C3_TIMING
  if (tid == 0) {
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (blk < 8192) {
      g_c3_phase[4 * blk + 0] = ts1 - ts0; g_c3_phase[4 * blk + 1] = t_k; g_c3_phase[4 * blk + 2] = t_e; g_c3_phase[4 * blk + 3] = te - ts0;
    }
  }
#endif
}

template <int MT, int LEFT, int LOGX, bool NARROW>
int launch_c3x(C3Args a, int B, hipStream_t s) {
  constexpr int MB = 16 * MT + 4 * LEFT;
  constexpr int XL = 1 << LOGX, NY = 16 >> LOGX;
  const int kc = ((a.K < kKC ? a.K : kKC) + 3) & ~3;
  const size_t lds = (size_t)c3_lds_floats(MB, kc) * sizeof(float);
  const int xt = (a.W + XL * 4 - 1) / (XL * 4);
  const int ntiles = (a.H + 4 * NY * kR - 1) / (4 * NY * kR);
  long tpb = a.tpb > 0 ? a.tpb : 1;             // chosen together with LOGX by launch_c3's cost model
  if (a.K > kKC) tpb = 1;                        // the panel is re-staged per tile anyway
  if (((g_c3_dbg >> 8) & 0xFF) && a.K <= kKC) tpb = (g_c3_dbg >> 8) & 0xFF;     // timing study: forced tiles per block
  a.tpb = (int)tpb;
  dim3 grid((unsigned)xt, (unsigned)((ntiles + tpb - 1) / tpb), (unsigned)(B * a.nmb));
  if (a.R)
    hipLaunchKernelGGL((conv3_kernel<MT, LEFT, LOGX, NARROW, true>), grid, dim3(kThreads), lds, s, a);
  else
    hipLaunchKernelGGL((conv3_kernel<MT, LEFT, LOGX, NARROW, false>), grid, dim3(kThreads), lds, s, a);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

template <int MT, int LEFT>
int launch_c3(const C3Args& a0, int B, hipStream_t s) {
  // Tile shape (64 / 32 / 16 pixels wide; every tile is 512 pixels) and row tiles per block, by a cost model in units of
  // one tile's time.  The blocks of a launch run in lockstep rounds of the 512 resident blocks (2 per CU), so
  //   cost = rounds * (stage + tiles per block),  rounds = ceil(blocks / 512),  stage ~ 0.15 (tools/c3_phases.py);
  // a grid of 1064 blocks (36->36 at 200x300 with 16-pixel-wide tiles) pays three rounds for two rounds of work.
  // Ties go to the shape with less padded work, then to fewer tiles per block.
  C3Args a = a0;
  int best = 4, best_tpb = 1;
  double best_cost = 1e30;
  long best_work = 1L << 60;
  for (int lx = 4; lx >= 2; --lx) {
    const int xl4 = 4 << lx, trows = 4 * (16 >> lx) * kR;
    const long xt = (a.W + xl4 - 1) / xl4, nt = (a.H + trows - 1) / trows;
    for (int tpb = 1; tpb <= (a.K <= kKC ? 4 : 1); ++tpb) {
      const long blocks = xt * ((nt + tpb - 1) / tpb) * B * a.nmb;
      const long rounds = (blocks + 511) / 512;
      const double stage = a.K <= kKC ? 0.15 : 0.0;          // deeper layers stage inside every tile: same for all shapes
      const double cost = (double)rounds * (stage + tpb);
      const long work = xt * nt;
      if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && work < best_work)) {
        best_cost = cost; best_work = work; best = lx; best_tpb = tpb;
      }
    }
  }
  a.tpb = best_tpb;
  if (a.W < 8) return launch_c3x<MT, LEFT, 2, true>(a, B, s);
  if (best == 4) return launch_c3x<MT, LEFT, 4, false>(a, B, s);
  if (best == 3) return launch_c3x<MT, LEFT, 3, false>(a, B, s);
  return launch_c3x<MT, LEFT, 2, false>(a, B, s);
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[m][n][dy][dx] = sum_{b,y,x} dY[b][m][y][x] * Xpad[b][n][y+dy-1][x+dx-1]
// A operand = dY rows (16 output channels x 4 k-slots), B operand = X rows shifted by the tap; the
// k-slot (j, e) <-> pixel xs + 8j + e of image row y.  A wave sweeps a 32-pixel-wide column of rows
// with a 3-row sliding window of X, holding all 9 taps x MT x 1 accumulator tiles.
// ---------------------------------------------------------------------------------------------
struct C3WgArgs {
  const float* dY; long dy_bs;
  const float* X; long x_bs;
  float* slabs;     // [B][chunks][M*N*9]
  int M, N, H, W;
  int replicate;
  int rr;           // rows per chunk
  int ncol, nitems; // 32-pixel column tiles per row; work items = ncol * row chunks (a wave owns one item)
  int nfull, ngrp;  // 16-wide n tiles, 4-wide groups of the input-channel remainder
  int nby;          // (m block, n block) pairs per pixel chunk in this launch
  int nchunk;       // pixel chunks per sample
  int dbg;          // timing-study switches: 32 no main loop, 64 no epilogue
};

struct Win10 {
  float v[10];
};

// Window xq-1 .. xq+8 of row yy in two steps, both straight-line code (W >= 8, 0 <= xq <= W-8).  raw_win10 only
// ISSUES the loads: two 16 B loads that are always inside the row and the two halo scalars at clamped addresses
// (which is the replicate value).  finish_win10 applies the border selects (zero padding, rows outside the image).
// The split matters: the loads for row y+1 are issued before row y's MFMA burst and finished after it, so their
// latency hides behind ~5000 MFMA cycles.  Any arithmetic on the loaded values placed next to the loads (selects,
// or the joins of a branchy per-element border path) makes the compiler wait for the data BEFORE the burst.
struct RawWin10 {
  f32x4 a, b;
  float lft, rgt;
};

__device__ __forceinline__ RawWin10 raw_win10(const float* __restrict__ plane, int yy, int xq, int H, int W) {
  const float* row = plane + (long)min(max(yy, 0), H - 1) * W;
  RawWin10 r;
  r.a = load4u(row + xq);
  r.b = load4u(row + xq + 4);
  r.lft = row[max(xq - 1, 0)];
  r.rgt = row[min(xq + 8, W - 1)];
  return r;
}

__device__ __forceinline__ Win10 finish_win10(const RawWin10& w, int yy, int xq, int H, int W, bool replicate) {
  Win10 r;
  r.v[0] = (xq == 0 && !replicate) ? 0.f : w.lft;
  r.v[1] = w.a[0]; r.v[2] = w.a[1]; r.v[3] = w.a[2]; r.v[4] = w.a[3];
  r.v[5] = w.b[0]; r.v[6] = w.b[1]; r.v[7] = w.b[2]; r.v[8] = w.b[3];
  r.v[9] = (xq + 8 >= W && !replicate) ? 0.f : w.rgt;
  const bool ok = replicate || (yy >= 0 && yy < H);
#pragma unroll
  for (int i = 0; i < 10; ++i) r.v[i] = ok ? r.v[i] : 0.f;
  return r;
}

// Fallback for images narrower than 8 pixels (bottom levels of toy-sized inputs): one thread per weight element.
__global__ void conv3_wgrad_narrow_kernel(const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs,
                                          int replicate, float* __restrict__ dW, int B, int M, int N, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N * 9) return;
  const int m = i / (N * 9), rem = i - m * (N * 9), n = rem / 9, t = rem - n * 9;
  const int dy = t / 3 - 1, dx = t % 3 - 1;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* g = dY + b * dy_bs + (long)m * H * W;
    const float* x = X + b * x_bs + (long)n * H * W;
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx) {
        int yy = y + dy, xc = xx + dx;
        if (replicate) { yy = min(max(yy, 0), H - 1); xc = min(max(xc, 0), W - 1); }
        else if (yy < 0 || yy >= H || xc < 0 || xc >= W) continue;
        s += g[(long)y * W + xx] * x[(long)yy * W + xc];
      }
  }
  dW[i] = s;
}

// MT 16-row tiles + LEFT 4-row groups of output channels per block (as in the forward kernel).  NLEFT = false: the
// block owns a full 16-column n tile (16x16x4 MFMAs for the tiles, 4x4x1 for the row groups).  NLEFT = true: the
// block owns one 4-column group of the input-channel remainder (N = 36: columns 32..35) and runs everything on
// 4x4x1: its 16 blocks are (k-slot j) x (4 row- or column-groups), lane l feeding its own dY / X values, so no
// operand is padded to 16.  4x4x1 results hold one k-slot's share per 16-lane group; the shares are added in the
// LDS pass that also adds the four waves.
template <int MT, int LEFT, bool NLEFT>
__global__ __launch_bounds__(kThreads) void conv3_wgrad_kernel(C3WgArgs a) {
  extern __shared__ float red[];                 // [4 waves][(MT+LEFT)*4][64]
  constexpr int LG = LEFT > 0 ? LEFT : 1;
  constexpr int MB = 16 * MT + 4 * LEFT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, j = lane >> 4;
  const int b = blockIdx.z;
  const int chunk = blockIdx.x, by = blockIdx.y;
  const int nny = NLEFT ? a.ngrp : a.nfull;       // n blocks of this launch
  const int mb = by / nny, nt = by - mb * nny;
  const int m0 = mb * MB, n0 = NLEFT ? a.nfull * 16 + nt * 4 : nt * 16;
  // work items (column tile, row chunk) are dealt to the waves of consecutive blocks in one flat sequence, so a plane whose
  // width is not a multiple of 128 leaves no wave idle (150 px = 5 column tiles: 8 wave slots per row chunk before)
  const int item = chunk * 4 + wave;
  const bool seg_ok = item < a.nitems;
  const int col = seg_ok ? item % a.ncol : 0, rchunk = seg_ok ? item / a.ncol : 0;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  const int xs = col * 32 + 8 * j;                         // this lane's first pixel of the segment
  // the lane that crosses the right border is pulled back to W-8 and its first `dup` pixels (owned by its left
  // neighbour) are masked out of the A operand; lanes entirely past the border mask everything (W >= 8)
  const int xq = min(xs, W - 8), dup = xs - xq;
  const int ya = rchunk * a.rr, yb = min(ya + a.rr, H);
  const bool rep = a.replicate != 0;

  C3_T(tw0);
#ifdef C3_TIMING
  unsigned long long t_wait = 0;
#endif
  f32x4 acc[9][MT];
  f32x4 accl[9][LG];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int lg = 0; lg < LG; ++lg) accl[t][lg] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (seg_ok && ya < yb && !(a.dbg & 32)) {
    // rows / columns past M / N are clamped to a valid plane: they only reach accumulator entries
    // that are never stored
    const float* xpl = a.X + (long)b * a.x_bs + (long)min(n0 + (NLEFT ? (lane & 3) : r), a.N - 1) * HW;
    const float* ypl[MT + LG];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ypl[mt] = a.dY + (long)b * a.dy_bs + (long)min(m0 + mt * 16 + r, a.M - 1) * HW;
#pragma unroll
    for (int lg = 0; lg < LG; ++lg)
      ypl[MT + lg] = a.dY + (long)b * a.dy_bs + (long)min(m0 + MT * 16 + lg * 4 + (lane & 3), a.M - 1) * HW;
    f32x4 dyraw[MT + LG][2];                       // next row's dY, as loaded
    auto issue_dy = [&](int y) {
#pragma unroll
      for (int mt = 0; mt < MT + LEFT; ++mt) {
        const float* row = ypl[mt] + (long)y * W;
        dyraw[mt][0] = load4u(row + xq);
        dyraw[mt][1] = load4u(row + xq + 4);
      }
    };
    Win10 w0 = finish_win10(raw_win10(xpl, ya - 1, xq, H, W), ya - 1, xq, H, W, rep);
    Win10 w1 = finish_win10(raw_win10(xpl, ya, xq, H, W), ya, xq, H, W, rep);
    RawWin10 w2raw = raw_win10(xpl, ya + 1, xq, H, W);
    issue_dy(ya);
    for (int y = ya; y < yb; ++y) {
      // consume what was requested one iteration (one MFMA burst) ago ...
#ifdef C3_TIMING
      const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // how long do the prefetched rows still take to arrive?
      t_wait += __builtin_amdgcn_s_memtime() - tq0;
#endif
      const Win10 w2 = finish_win10(w2raw, y + 1, xq, H, W, rep);
      float av[MT + LG][8];
#pragma unroll
      for (int mt = 0; mt < MT + LEFT; ++mt)
#pragma unroll
        for (int e = 0; e < 8; ++e) av[mt][e] = e < dup ? 0.f : dyraw[mt][e >> 2][e & 3];
      // ... and request the next row's operands before this row's burst
      if (y + 1 < yb) {
        w2raw = raw_win10(xpl, y + 2, xq, H, W);
        issue_dy(y + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (NLEFT) {
              acc[dx][mt] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[mt][e], w0.v[e + dx], acc[dx][mt], 0, 0, 0);
              acc[3 + dx][mt] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[mt][e], w1.v[e + dx], acc[3 + dx][mt], 0, 0, 0);
              acc[6 + dx][mt] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[mt][e], w2.v[e + dx], acc[6 + dx][mt], 0, 0, 0);
            } else {
              acc[dx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w0.v[e + dx], acc[dx][mt], 0, 0, 0);
              acc[3 + dx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w1.v[e + dx], acc[3 + dx][mt], 0, 0, 0);
              acc[6 + dx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w2.v[e + dx], acc[6 + dx][mt], 0, 0, 0);
            }
          }
#pragma unroll
          for (int lg = 0; lg < LEFT; ++lg) {
            accl[dx][lg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[MT + lg][e], w0.v[e + dx], accl[dx][lg], 0, 0, 0);
            accl[3 + dx][lg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[MT + lg][e], w1.v[e + dx], accl[3 + dx][lg], 0, 0, 0);
            accl[6 + dx][lg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[MT + lg][e], w2.v[e + dx], accl[6 + dx][lg], 0, 0, 0);
          }
        }
      // (no fence behind the burst here: letting the scheduler sink the window rotation into its tail measured 1-3 % faster)
      w0 = w1; w1 = w2;
    }
  }

  C3_T(tw1);
#ifdef C3_TIMING
  struct C3WgEnd {                               // [wait for prefetched rows, main loop, epilogue, total] of wave 0
    unsigned long long t0, t1; unsigned long long& wait;
    __device__ ~C3WgEnd() {
      if (threadIdx.x == 0) {
        const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (blk < 8192) {
          const unsigned long long te = __builtin_amdgcn_s_memtime();
          g_c3_phase[4 * blk + 0] = wait; g_c3_phase[4 * blk + 1] = t1 - t0; g_c3_phase[4 * blk + 2] = te - t1; g_c3_phase[4 * blk + 3] = te - t0;
        }
      }
    }
  } c3_wg_end{tw0, tw1, t_wait};
#endif
  // sum the four waves' tiles (and, for 4x4x1 results, the four k-slot shares) through LDS, one tap per round
  float* slab = a.slabs + (((long)b * a.nchunk + chunk) * (long)a.M) * a.N * 9;
  constexpr int TILE = (MT + LEFT) * 4;
  if (a.dbg & 64) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(acc[t][mt]));
#pragma unroll
      for (int lg = 0; lg < LEFT; ++lg) asm volatile("" ::"v"(accl[t][lg]));
    }
    return;
  }
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT + LEFT; ++mt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float v = 0.f;
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) {                                   // static register indexing
          if (mt < MT) v = (tt == t) ? acc[tt][mt < MT ? mt : 0][reg] : v;
          else v = (tt == t) ? accl[tt][mt >= MT ? mt - MT : 0][reg] : v;
        }
        red[(wave * TILE + mt * 4 + reg) * 64 + lane] = v;
      }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TILE * 64; idx += kThreads) {
      const int l = idx & 63, q = idx >> 6;
      const int reg = q & 3, mt = q >> 2;
      const bool x4 = NLEFT || mt >= MT;                                   // produced by 4x4x1: k-slot shares in l, l+16, l+32, l+48
      if (x4 && l >= 16) continue;
      float v = 0.f;
      if (x4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int k = idx + 16 * jj;
          v += (red[k] + red[TILE * 64 + k]) + (red[2 * TILE * 64 + k] + red[3 * TILE * 64 + k]);
        }
      } else {
        v = (red[idx] + red[TILE * 64 + idx]) + (red[2 * TILE * 64 + idx] + red[3 * TILE * 64 + idx]);
      }
      int m, n;
      if (!NLEFT) {
        m = m0 + (mt < MT ? mt * 16 + (l >> 4) * 4 + reg : MT * 16 + (mt - MT) * 4 + reg);
        n = n0 + (l & 15);
      } else {
        if (mt >= MT && l >= 4) continue;                                  // the 4 row-group blocks hold identical values
        m = m0 + (mt < MT ? mt * 16 + 4 * (l >> 2) + reg : MT * 16 + (mt - MT) * 4 + reg);
        n = n0 + (l & 3);
      }
      if (m < a.M && m < m0 + MB && n < a.N) slab[((long)m * a.N + n) * 9 + t] = v;
    }
  }
}

// out[i] = sum_k slabs[k][i]: 8 elements per block, 32 lanes per element striding over the slabs with two
// independent partial sums (short latency-bound launch), fixed-order fold through LDS
constexpr int kC3RedElems = 8;

__global__ __launch_bounds__(256) void c3_reduce_kernel(const float* __restrict__ slabs, int n_red, long ne, float* __restrict__ out) {
  __shared__ float part[32][kC3RedElems + 1];
  const int ex = threadIdx.x & (kC3RedElems - 1), sl = threadIdx.x / kC3RedElems;
  const long i = (long)blockIdx.x * kC3RedElems + ex;
  float t0 = 0.f, t1 = 0.f;
  if (i < ne) {
    int k = sl;
    for (; k + 32 < n_red; k += 64) { t0 += slabs[(long)k * ne + i]; t1 += slabs[(long)(k + 32) * ne + i]; }
    if (k < n_red) t0 += slabs[(long)k * ne + i];
  }
  part[sl][ex] = t0 + t1;
  __syncthreads();
  if (sl == 0 && i < ne) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (part[4 * q][ex] + part[4 * q + 1][ex]) + (part[4 * q + 2][ex] + part[4 * q + 3][ex]);
    out[i] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
}

// Decomposition of the (M x N) weight-gradient into blocks.  Output channels: equal blocks of <= 48 rows = MT 16-row
// tiles + LEFT 4-row groups (MT + LEFT <= 3 accumulator sets; see cidnet_conv3x3).  Input channels: full 16-column
// tiles, and a remainder of <= 8 columns as 4-column groups run by the NLEFT kernel (else one more padded tile).
struct WgSplit {
  int MT, LEFT, nmb, nfull, ngrp;
};

inline WgSplit wg_split(int M, int N) {
  WgSplit w;
  const int nblk = (M + 47) / 48;
  const int rows = (((M + nblk - 1) / nblk) + 3) & ~3;
  w.MT = rows / 16; w.LEFT = (rows % 16) / 4;
  if (w.MT == 0 || w.MT + w.LEFT > 3 || (g_c3_dbg & 16)) { w.MT = (rows + 15) / 16; w.LEFT = 0; }
  w.nmb = (M + 16 * w.MT + 4 * w.LEFT - 1) / (16 * w.MT + 4 * w.LEFT);
  w.nfull = N / 16; w.ngrp = ((N % 16) + 3) / 4;
  if (w.nfull == 0 || w.ngrp > 2 || (g_c3_dbg & (16 | 128))) { w.nfull = (N + 15) / 16; w.ngrp = 0; }   // 128: timing study
  return w;
}

// Rows per pixel chunk.  256 CUs hold 512 of these blocks at a time (2 waves per SIMD), and a block's run time is
// (rows + ~6 rows' worth of prologue / 9-round epilogue), so the launch costs about ceil(blocks / 512) * (rows + 6):
// a grid just above a multiple of 512 pays a whole extra round for a few blocks.  Pick the cheapest row count >= 8.
#ifndef C3_WG_OVH
#define C3_WG_OVH 6      // fixed cost of a block (window prologue, LDS sum, slab store) in units of one row's burst
#endif
inline int wg_rows(int B, int M, int N, int H, int W) {
  const WgSplit w = wg_split(M, N);
  const int ncol = (W + 31) / 32;
  const long per_chunk = (long)w.nmb * w.nfull * B;                               // blocks of the main launch per pixel chunk
  int best_rr = H;
  long best_cost = -1;
  for (int nrc = 1; nrc <= H; ++nrc) {
    const int rr = (H + nrc - 1) / nrc;
    if (rr < 8 && nrc > 1) break;
    const long chunks = ((long)ncol * ((H + rr - 1) / rr) + 3) / 4;
    const long rounds = (per_chunk * chunks + 511) / 512;
    const long cost = rounds * (rr + C3_WG_OVH);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rr = rr; }
  }
  return best_rr;
}

// Border correction of the data gradient of (ReplicationPad2d(1) + valid 3x3 conv): the zero-pad
// data gradient misses the taps that read a replicated border pixel; add them in place.
// gX[b][ci][u][v] += sum_co sum_{(y,dy),(x,dx) with at least one clamped coordinate} W[co][ci][dy][dx] * gY[b][co][y][x]
__global__ void c3_replicate_fix_kernel(const float* __restrict__ gY, const float* __restrict__ Wt, float* __restrict__ gX,
                                        int B, int Co, int Ci, int H, int W) {
  const int nb = 2 * W + 2 * (H - 2 > 0 ? H - 2 : 0);     // border pixels
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * Ci * nb) return;
  const int k = (int)(idx % nb);
  const int ci = (int)((idx / nb) % Ci);
  const int b = (int)(idx / ((long)nb * Ci));
  int u, v;
  if (k < W) { u = 0; v = k; }
  else if (k < 2 * W) { u = H - 1; v = k - W; }
  else { const int t = k - 2 * W; u = 1 + (t >> 1); v = (t & 1) ? W - 1 : 0; }
  if (H == 1 && k >= W) return;
  if (W == 1 && k >= 2 * W && (k & 1)) return;
  const long HW = (long)H * W;
  float s = 0.f;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      // all (y, x) whose padded tap (y+dy-1, x+dx-1) clamps onto (u, v)
      for (int cy = 0; cy < 2; ++cy) {          // cy = 1: clamped row coordinate
        int y;
        if (cy == 0) { y = u - dy + 1; if (y < 0 || y >= H) continue; }
        else { if (u == 0 && dy == 0) y = 0; else if (u == H - 1 && dy == 2) y = H - 1; else continue; }
        for (int cx = 0; cx < 2; ++cx) {
          int x;
          if (cx == 0) { x = v - dx + 1; if (x < 0 || x >= W) continue; }
          else { if (v == 0 && dx == 0) x = 0; else if (v == W - 1 && dx == 2) x = W - 1; else continue; }
          if (cy == 0 && cx == 0) continue;     // the unclamped tap is in the zero-pad gradient already
          for (int co = 0; co < Co; ++co)
            s += Wt[((long)co * Ci + ci) * 9 + dy * 3 + dx] * gY[((long)b * Co + co) * HW + (long)y * W + x];
        }
      }
    }
  gX[((long)b * Ci + ci) * HW + (long)u * W + v] += s;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

#ifdef CIDNET_DEBUG
void cidnet_debug_c3_flags(int flags) { g_c3_dbg = flags; }
#endif

#ifdef C3_TIMING
// copies the per-block phase cycles [stage, k loops, epilogues, total] of the last conv3_kernel launch to the host
int cidnet_debug_c3_phases(unsigned long long* host, int nblocks) {
  (void)hipDeviceSynchronize();
  const size_t n = sizeof(unsigned long long) * 4 * (nblocks < 8192 ? nblocks : 8192);
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c3_phase), n);
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_c3_phase)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 4 * 8192);
  return rc;
}
#endif

int cidnet_conv3x3_add(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, int replicate, const float* R,
                       long r_bs, float* Y, long y_bs, int B, int M, int K, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0);
  // the addend lives in the MFMA kernel's epilogue; the streaming kernels of <= 4-channel layers do not take one
  CIDNET_CHECK_ARG(!R || !(c3_thin_applies(M, K) && !(g_c3_dbg & 8)));
  C3Args a{};
  a.X = X; a.x_bs = x_bs; a.Wt = Wt; a.w_ms = w_ms; a.w_ks = w_ks; a.Y = Y; a.y_bs = y_bs; a.R = R; a.r_bs = r_bs;
  a.M = M; a.K = K; a.H = H; a.W = W; a.flip = flip; a.replicate = replicate; a.dbg = g_c3_dbg;
  if (c3_thin_applies(M, K) && !(g_c3_dbg & 8))
    return c3_thin_conv(X, x_bs, Wt, w_ms, w_ks, flip, replicate, Y, y_bs, B, M, K, H, W, (hipStream_t)stream);
  // split M into equal blocks of at most 48 rows; a block is MT 16-row tiles plus LEFT 4-row groups
  // (MT + LEFT <= 3 accumulator sets), e.g. 36 = 2 tiles + 1 group, 72 = 2 x 36, 144 = 3 x 48, 24 = 1 tile + 2 groups
  const int nblk = (M + 47) / 48;
  const int rows = (((M + nblk - 1) / nblk) + 3) & ~3;
  int MT = rows / 16, LEFT = (rows % 16) / 4;
  if (MT == 0 || MT + LEFT > 3 || (g_c3_dbg & 16)) { MT = (rows + 15) / 16; LEFT = 0; }
  a.nmb = (M + 16 * MT + 4 * LEFT - 1) / (16 * MT + 4 * LEFT);
  hipStream_t s = (hipStream_t)stream;
  switch (MT * 4 + LEFT) {
    case 4: return launch_c3<1, 0>(a, B, s);
    case 5: return launch_c3<1, 1>(a, B, s);
    case 6: return launch_c3<1, 2>(a, B, s);
    case 8: return launch_c3<2, 0>(a, B, s);
    case 9: return launch_c3<2, 1>(a, B, s);
    default: return launch_c3<3, 0>(a, B, s);
  }
}

int cidnet_conv3x3(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, int replicate, float* Y,
                   long y_bs, int B, int M, int K, int H, int W, void* stream) {
  return cidnet_conv3x3_add(X, x_bs, Wt, w_ms, w_ks, flip, replicate, nullptr, 0, Y, y_bs, B, M, K, H, W, stream);
}

long cidnet_conv3x3_wgrad_ws_floats(int B, int M, int N, int H, int W) {
  if (c3_thin_applies(M, N) && !(g_c3_dbg & 8)) return (long)B * c3_thin_wgrad_chunks(H, W) * M * N * 9;
  const int rr = wg_rows(B, M, N, H, W);
  const long chunks = ((long)((W + 31) / 32) * ((H + rr - 1) / rr) + 3) / 4;
  return (long)B * chunks * M * N * 9;
}

int cidnet_conv3x3_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int replicate, float* dW, float* ws,
                         long ws_floats, int B, int M, int N, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(dY && X && dW && ws && B > 0 && M > 0 && N > 0 && H > 0 && W > 0);
  if (ws_floats < cidnet_conv3x3_wgrad_ws_floats(B, M, N, H, W)) return CIDNET_ERR_WS;
  if (c3_thin_applies(M, N) && !(g_c3_dbg & 8)) {
    const int rc = c3_thin_wgrad(dY, dy_bs, X, x_bs, replicate, ws, B, M, N, H, W, (hipStream_t)stream);
    if (rc != CIDNET_OK) return rc;
    const long ne = (long)M * N * 9;
    hipLaunchKernelGGL(c3_reduce_kernel, dim3((unsigned)((ne + kC3RedElems - 1) / kC3RedElems)), dim3(256), 0, (hipStream_t)stream, ws,
                       B * c3_thin_wgrad_chunks(H, W), ne, dW);
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  if (W < 8) {
    hipLaunchKernelGGL(conv3_wgrad_narrow_kernel, dim3((unsigned)((M * N * 9 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dY,
                       dy_bs, X, x_bs, replicate, dW, B, M, N, H, W);
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  C3WgArgs a{};
  a.dY = dY; a.dy_bs = dy_bs; a.X = X; a.x_bs = x_bs; a.slabs = ws; a.M = M; a.N = N; a.H = H; a.W = W;
  a.replicate = replicate; a.rr = wg_rows(B, M, N, H, W);
  a.ncol = (W + 31) / 32;
  a.nitems = a.ncol * ((H + a.rr - 1) / a.rr);
  const int chunks = (a.nitems + 3) / 4;
  const WgSplit sp = wg_split(M, N);
  const int MT = sp.MT, LEFT = sp.LEFT, nmb = sp.nmb, nfull = sp.nfull, ngrp = sp.ngrp;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)4 * (MT + LEFT) * 4 * 64 * sizeof(float);
  a.nfull = nfull; a.ngrp = ngrp; a.nchunk = chunks; a.dbg = g_c3_dbg;
  const int code = MT * 4 + LEFT;
#define CIDNET_WG_LAUNCH(MT_, LEFT_, NL_) hipLaunchKernelGGL((conv3_wgrad_kernel<MT_, LEFT_, NL_>), grid, dim3(kThreads), lds, s, a)
#define CIDNET_WG_DISPATCH(NL_)                                                  \
  switch (code) {                                                                \
    case 4: CIDNET_WG_LAUNCH(1, 0, NL_); break;                                  \
    case 5: CIDNET_WG_LAUNCH(1, 1, NL_); break;                                  \
    case 6: CIDNET_WG_LAUNCH(1, 2, NL_); break;                                  \
    case 8: CIDNET_WG_LAUNCH(2, 0, NL_); break;                                  \
    case 9: CIDNET_WG_LAUNCH(2, 1, NL_); break;                                  \
    default: CIDNET_WG_LAUNCH(3, 0, NL_); break;                                 \
  }
  {
    a.nby = nmb * nfull;
    const dim3 grid((unsigned)chunks, (unsigned)a.nby, (unsigned)B);
    CIDNET_WG_DISPATCH(false)
    CIDNET_LAUNCH_STATUS();
  }
  if (ngrp > 0) {
    a.nby = nmb * ngrp;
    const dim3 grid((unsigned)chunks, (unsigned)a.nby, (unsigned)B);
    CIDNET_WG_DISPATCH(true)
    CIDNET_LAUNCH_STATUS();
  }
#undef CIDNET_WG_DISPATCH
#undef CIDNET_WG_LAUNCH
  const long ne = (long)M * N * 9;
  hipLaunchKernelGGL(c3_reduce_kernel, dim3((unsigned)((ne + kC3RedElems - 1) / kC3RedElems)), dim3(256), 0, s, ws, B * chunks, ne, dW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_conv3x3_replicate_dgrad_fix(const float* gY, const float* Wt, float* gX, int B, int Co, int Ci, int H, int W,
                                       void* stream) {
  CIDNET_CHECK_ARG(gY && Wt && gX && B > 0 && Co > 0 && Ci > 0 && H > 0 && W > 0);
  const long nb = 2L * W + 2L * (H - 2 > 0 ? H - 2 : 0);
  const long total = (long)B * Ci * nb;
  hipLaunchKernelGGL(c3_replicate_fix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gY, Wt,
                     gX, B, Co, Ci, H, W);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
