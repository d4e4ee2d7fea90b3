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
#include "common.h"
#include "conv3_thin.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kR = 2;      // output rows per wave
constexpr int kKC = 36;    // input channels staged per chunk (9 k-groups)
int g_c3_dbg = 0;

struct C3Args {
  const float* X; long x_bs;
  const float* Wt; long w_ms, w_ks;   // A[m][k][tap] = Wt[m*w_ms + k*w_ks + tap']
  float* Y; long y_bs;
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

// LOGX: log2 of the lanes (x 4 pixels) a 16-lane group spends on x; the other 16>>LOGX lanes take
// further rows, so narrow images (W = 75, 150) do not waste most of a 64-pixel-wide tile.
constexpr int c3_lda(int mb) { return ((mb + 15) / 32) * 32 + 16; }   // smallest >= mb that is 16 (mod 32)

__device__ __forceinline__ float pick4(f32x4 v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); }

template <int MT, int LEFT, int LOGX>
__global__ __launch_bounds__(kThreads) void conv3_kernel(C3Args a) {
  extern __shared__ float As[];                 // [kKC][9][ldA]
  constexpr int MB = 16 * MT + 4 * LEFT;        // output channels of one block
  constexpr int ldA = c3_lda(MB);
  constexpr int LG = LEFT > 0 ? LEFT : 1;       // array extent (LEFT = 0: unused)
  constexpr int XL = 1 << LOGX, NY = 16 >> LOGX;
  constexpr int TROWS = 4 * NY * kR;            // output rows of one block tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, j = lane >> 4;
  const int b = blockIdx.z / a.nmb, mb = blockIdx.z - b * a.nmb;
  const int m0 = mb * MB;
  const int x0 = blockIdx.x * (XL * 4) + 4 * (c & (XL - 1));
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  const float* Xb = a.X + (long)b * a.x_bs;
  const bool rep = a.replicate != 0;
  const bool single = a.K <= kKC;               // whole weight panel resident: walk `tpb` row tiles

  // consecutive threads walk the contiguous axis of the weight tensor ((k,tap) for the forward
  // layout, (m,tap) for the data-gradient layout): full-line fetches instead of 4-byte gathers
  auto stage = [&](int kc0, int kcn, int ng) {
    const int kq = ng * 4;
    if (a.w_ks == 9) {
      for (int i = tid; i < MB * kq * 9; i += kThreads) {
        const int mm = i / (kq * 9), rem = i - mm * (kq * 9);
        const int kk = rem / 9, t = rem - kk * 9;
        float v = 0.f;
        if (kk < kcn && m0 + mm < a.M)
          v = a.Wt[(long)(m0 + mm) * a.w_ms + (long)(kc0 + kk) * 9 + (a.flip ? 8 - t : t)];
        As[(kk * 9 + t) * ldA + mm] = v;
      }
    } else {
      for (int i = tid; i < kq * MB * 9; i += kThreads) {
        const int kk = i / (MB * 9), rem = i - kk * (MB * 9);
        const int mm = rem / 9, t = rem - mm * 9;
        float v = 0.f;
        if (kk < kcn && m0 + mm < a.M)
          v = a.Wt[(long)(m0 + mm) * a.w_ms + (long)(kc0 + kk) * a.w_ks + (a.flip ? 8 - t : t)];
        As[(kk * 9 + t) * ldA + mm] = v;
      }
    }
  };
  if (single) {
    stage(0, a.K, (a.K + 3) >> 2);
    __syncthreads();
  }

  const int ntiles = (H + TROWS - 1) / TROWS;
  const int tile_end = min((int)(blockIdx.y + 1) * a.tpb, ntiles);
  for (int tile = blockIdx.y * a.tpb; tile < tile_end; ++tile) {
    const int ywave = tile * TROWS + wave * (NY * kR);
    const int yw = ywave + (c >> LOGX) * kR;    // first output row of this lane
    const bool wave_live = ywave < H;

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
      Win6 nxt[kR + 2];
      {
        const float* plane = Xb + (long)min(kc0 + j, a.K - 1) * HW;
#pragma unroll
        for (int iy = 0; iy < kR + 2; ++iy) nxt[iy] = load_win(plane, yw - 1 + iy, x0, H, W, rep);
      }
      for (int g = 0; g < ng; ++g) {
        Win6 win[kR + 2];
#pragma unroll
        for (int iy = 0; iy < kR + 2; ++iy) win[iy] = nxt[iy];
        if (g + 1 < ng && !(a.dbg & 2)) {
          const float* plane = Xb + (long)min(kc0 + 4 * (g + 1) + j, a.K - 1) * HW;
#pragma unroll
          for (int iy = 0; iy < kR + 2; ++iy) nxt[iy] = load_win(plane, yw - 1 + iy, x0, H, W, rep);
        }
        float av[9][MT];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[t][mt] = As[(((a.dbg & 4) ? 0 : g * 4 + j) * 9 + t) * ldA + mt * 16 + c];
        float al[9][LG];
        if (LEFT > 0) {
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int lg = 0; lg < LEFT; ++lg)
              al[t][lg] = As[(((a.dbg & 4) ? 0 : g * 4 + j) * 9 + t) * ldA + MT * 16 + lg * 4 + (lane & 3)];
        }
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
      }
    }

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
          float* row = a.Y + (long)b * a.y_bs + (long)m * HW + (long)y * W;
          const f32x4 v = {acc[r][mt][0][reg], acc[r][mt][1][reg], acc[r][mt][2][reg], acc[r][mt][3][reg]};
          if (x0 + 3 < W) {
            store4u(row + x0, v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (x0 + e < W) row[x0 + e] = v[e];
          }
        }
#pragma unroll
      for (int lg = 0; lg < LEFT; ++lg) {          // lane (c, j) stores row 16*MT + 4*lg + j of the 4-row group
        const int m = m0 + MT * 16 + lg * 4 + j;
        if (m >= a.M) continue;
        float* row = a.Y + (long)b * a.y_bs + (long)m * HW + (long)y * W;
        const f32x4 v = {pick4(accl[r][lg][0], j), pick4(accl[r][lg][1], j), pick4(accl[r][lg][2], j), pick4(accl[r][lg][3], j)};
        if (x0 + 3 < W) {
          store4u(row + x0, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (x0 + e < W) row[x0 + e] = v[e];
        }
      }
    }
  }
}

template <int MT, int LEFT, int LOGX>
int launch_c3x(C3Args a, int B, hipStream_t s) {
  constexpr int MB = 16 * MT + 4 * LEFT;
  constexpr int ldA = c3_lda(MB);
  constexpr int XL = 1 << LOGX, NY = 16 >> LOGX;
  const int kc = ((a.K < kKC ? a.K : kKC) + 3) & ~3;
  const size_t lds = (size_t)kc * 9 * ldA * sizeof(float);
  const int xt = (a.W + XL * 4 - 1) / (XL * 4);
  const int ntiles = (a.H + 4 * NY * kR - 1) / (4 * NY * kR);
  long tpb = 1;
  if (a.K <= kKC) {                              // weights stay resident: walk several row tiles per block
    tpb = (long)xt * ntiles * B * a.nmb / 2048;
    tpb = tpb < 1 ? 1 : (tpb > 4 ? 4 : tpb);
  }
  a.tpb = (int)tpb;
  dim3 grid((unsigned)xt, (unsigned)((ntiles + tpb - 1) / tpb), (unsigned)(B * a.nmb));
  hipLaunchKernelGGL((conv3_kernel<MT, LEFT, LOGX>), grid, dim3(kThreads), lds, s, a);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

template <int MT, int LEFT>
int launch_c3(const C3Args& a, int B, hipStream_t s) {
  // pick the x extent of a lane group (64 / 32 / 16 pixels) that wastes the fewest columns
  int best = 4;
  long best_cols = 1L << 60;
  for (int lx = 4; lx >= 2; --lx) {
    const int tw = 4 << lx;
    const long cols = (long)((a.W + tw - 1) / tw) * tw;
    if (cols < best_cols) { best_cols = cols; best = lx; }
  }
  if (best == 4) return launch_c3x<MT, LEFT, 4>(a, B, s);
  if (best == 3) return launch_c3x<MT, LEFT, 3>(a, B, s);
  return launch_c3x<MT, LEFT, 2>(a, B, s);
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
  int nseg4;        // ceil(nseg / 4): x-segment groups
  int nnt;          // number of 16-wide n tiles
};

struct Win10 {
  float v[10];
};

__device__ __forceinline__ Win10 load_win10(const float* __restrict__ plane, int yy, int xs, int H, int W, bool ok,
                                            bool replicate) {
  Win10 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.v[i] = 0.f;
  if (yy < 0 || yy >= H) {
    if (replicate) yy = yy < 0 ? 0 : H - 1; else ok = false;
  }
  if (!ok) return r;
  const float* row = plane + (long)yy * W;
  if (xs >= 1 && xs + 8 < W) {
    const f32x4 a = load4u(row + xs), b = load4u(row + xs + 4);
    r.v[0] = row[xs - 1];
    r.v[1] = a[0]; r.v[2] = a[1]; r.v[3] = a[2]; r.v[4] = a[3];
    r.v[5] = b[0]; r.v[6] = b[1]; r.v[7] = b[2]; r.v[8] = b[3];
    r.v[9] = row[xs + 8];
  } else {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      int x = xs - 1 + i;
      if (x < 0) { if (replicate) x = 0; else continue; }
      if (x >= W) { if (replicate) x = W - 1; else continue; }
      r.v[i] = row[x];
    }
  }
  return r;
}

template <int MT>
__global__ __launch_bounds__(kThreads) void conv3_wgrad_kernel(C3WgArgs a) {
  extern __shared__ float red[];                 // [4 waves][MT*4][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, j = lane >> 4;
  const int b = blockIdx.z;
  const int mb = blockIdx.y / a.nnt, nt = blockIdx.y - mb * a.nnt;
  const int m0 = mb * 16 * MT, n0 = nt * 16;
  const int segg = blockIdx.x % a.nseg4, rchunk = blockIdx.x / a.nseg4;
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  const int xs = ((segg * 4 + wave) * 32) + 8 * j;         // this lane's first pixel of the segment
  const bool seg_ok = (segg * 4 + wave) * 32 < W;
  const int ya = rchunk * a.rr, yb = min(ya + a.rr, H);
  const bool rep = a.replicate != 0;

  f32x4 acc[9][MT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (seg_ok && ya < yb) {
    // rows / columns past M / N are clamped to a valid plane: they only reach accumulator entries
    // that are never stored
    const float* xpl = a.X + (long)b * a.x_bs + (long)min(n0 + r, a.N - 1) * HW;
    const float* ypl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ypl[mt] = a.dY + (long)b * a.dy_bs + (long)min(m0 + mt * 16 + r, a.M - 1) * HW;
    auto load_dy = [&](int y, float (&av)[MT][8]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float* row = ypl[mt] + (long)y * W;
        if (xs + 7 < W) {
          const f32x4 p = load4u(row + xs), q = load4u(row + xs + 4);
          av[mt][0] = p[0]; av[mt][1] = p[1]; av[mt][2] = p[2]; av[mt][3] = p[3];
          av[mt][4] = q[0]; av[mt][5] = q[1]; av[mt][6] = q[2]; av[mt][7] = q[3];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) av[mt][e] = (xs + e < W) ? row[xs + e] : 0.f;
        }
      }
    };
    Win10 w0 = load_win10(xpl, ya - 1, xs, H, W, true, rep);
    Win10 w1 = load_win10(xpl, ya, xs, H, W, true, rep);
    Win10 w2n = load_win10(xpl, ya + 1, xs, H, W, true, rep);
    float avn[MT][8];
    load_dy(ya, avn);
    for (int y = ya; y < yb; ++y) {
      const Win10 w2 = w2n;
      float av[MT][8];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 8; ++e) av[mt][e] = avn[mt][e];
      if (y + 1 < yb) {                           // prefetch the next row's operands
        w2n = load_win10(xpl, y + 2, xs, H, W, true, rep);
        load_dy(y + 1, avn);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[dx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w0.v[e + dx], acc[dx][mt], 0, 0, 0);
            acc[3 + dx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w1.v[e + dx], acc[3 + dx][mt], 0, 0, 0);
            acc[6 + dx][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], w2.v[e + dx], acc[6 + dx][mt], 0, 0, 0);
          }
      w0 = w1; w1 = w2;
    }
  }

  // sum the four waves' tiles through LDS (one tap per round), one slab per block
  float* slab = a.slabs + (((long)b * gridDim.x + blockIdx.x) * (long)a.M) * a.N * 9;
  constexpr int TILE = MT * 4;
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float v = 0.f;
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) v = (tt == t) ? acc[tt][mt][reg] : v;    // static register indexing
        red[(wave * TILE + mt * 4 + reg) * 64 + lane] = v;
      }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TILE * 64; idx += kThreads) {
      const float v = (red[idx] + red[TILE * 64 + idx]) + (red[2 * TILE * 64 + idx] + red[3 * TILE * 64 + idx]);
      const int l = idx & 63, q = idx >> 6;
      const int reg = q & 3, mt = q >> 2;
      const int m = m0 + mt * 16 + (l >> 4) * 4 + reg, n = n0 + (l & 15);
      if (m < a.M && n < a.N) slab[((long)m * a.N + n) * 9 + t] = v;
    }
  }
}

__global__ __launch_bounds__(256) void c3_reduce_kernel(const float* __restrict__ slabs, int n_red, long ne, float* __restrict__ out) {
  __shared__ float part[8][33];
  const int ex = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + ex;
  float t = 0.f;
  if (i < ne)
    for (int k = sl; k < n_red; k += 8) t += slabs[(long)k * ne + i];
  part[sl][ex] = t;
  __syncthreads();
  if (sl == 0 && i < ne)
    out[i] = ((part[0][ex] + part[1][ex]) + (part[2][ex] + part[3][ex])) + ((part[4][ex] + part[5][ex]) + (part[6][ex] + part[7][ex]));
}

// rows per chunk: enough blocks to fill the chip (>= ~1024) without going below 16 rows per block
inline int wg_rows(int B, int M, int N, int H, int W) {
  const int T = (M + 15) / 16;
  const int nblk = (T + 2) / 3;
  const int MT = (T + nblk - 1) / nblk;
  const int per_chunk = (((W + 31) / 32 + 3) / 4) * ((T + MT - 1) / MT) * ((N + 15) / 16) * B;
  int nrc = (1024 + per_chunk - 1) / per_chunk;
  const int max_nrc = H / 16 > 0 ? H / 16 : 1;
  nrc = nrc < 1 ? 1 : (nrc > max_nrc ? max_nrc : nrc);
  return (H + nrc - 1) / nrc;
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

void cidnet_debug_c3_flags(int flags) { g_c3_dbg = flags; }

int cidnet_conv3x3(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, int replicate, float* Y,
                   long y_bs, int B, int M, int K, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0);
  C3Args a{};
  a.X = X; a.x_bs = x_bs; a.Wt = Wt; a.w_ms = w_ms; a.w_ks = w_ks; a.Y = Y; a.y_bs = y_bs;
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

long cidnet_conv3x3_wgrad_ws_floats(int B, int M, int N, int H, int W) {
  if (c3_thin_applies(M, N) && !(g_c3_dbg & 8)) return (long)B * c3_thin_wgrad_chunks(H, W) * M * N * 9;
  const int rr = wg_rows(B, M, N, H, W);
  const long chunks = (long)(((W + 31) / 32 + 3) / 4) * ((H + rr - 1) / rr);
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
    hipLaunchKernelGGL(c3_reduce_kernel, dim3((unsigned)((ne + 31) / 32)), dim3(256), 0, (hipStream_t)stream, ws,
                       B * c3_thin_wgrad_chunks(H, W), ne, dW);
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  C3WgArgs a{};
  a.dY = dY; a.dy_bs = dy_bs; a.X = X; a.x_bs = x_bs; a.slabs = ws; a.M = M; a.N = N; a.H = H; a.W = W;
  a.replicate = replicate; a.rr = wg_rows(B, M, N, H, W);
  a.nseg4 = ((W + 31) / 32 + 3) / 4;
  a.nnt = (N + 15) / 16;
  const int chunks = a.nseg4 * ((H + a.rr - 1) / a.rr);
  const int T = (M + 15) / 16;
  const int nblk = (T + 2) / 3;
  const int MT = (T + nblk - 1) / nblk;
  const int nmb = (T + MT - 1) / MT;
  dim3 grid((unsigned)chunks, (unsigned)(nmb * a.nnt), (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)4 * MT * 4 * 64 * sizeof(float);
  if (MT == 1) hipLaunchKernelGGL((conv3_wgrad_kernel<1>), grid, dim3(kThreads), lds, s, a);
  else if (MT == 2) hipLaunchKernelGGL((conv3_wgrad_kernel<2>), grid, dim3(kThreads), lds, s, a);
  else hipLaunchKernelGGL((conv3_wgrad_kernel<3>), grid, dim3(kThreads), lds, s, a);
  CIDNET_LAUNCH_STATUS();
  const long ne = (long)M * N * 9;
  hipLaunchKernelGGL(c3_reduce_kernel, dim3((unsigned)((ne + 31) / 32)), dim3(256), 0, s, ws, B * chunks, ne, dW);
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
