// Weight gradient of the dense 3x3 convolution (zero pad 1) with fp32 operands on the BF16 matrix cores: exact "bf16x3" split
// products (arithmetic: conv3x.hip).  dW[m][ci][tap] = sum over pixels of dY[m][px] X[ci][px + tap]
// (autograd of net/transformer_utils.py:39,58).
//
// k = pixels.  A first attempt (profiles/r03_e_...rejected.txt) split the operands in the k loop, once per fragment: 44 VALU
// instructions per fragment that feeds only 18 MFMAs -- VALU-bound, slower than the fp32-MFMA kernel.  Here every element is
// split ONCE, while its tile is staged into LDS exactly as conv3x.hip stages activations (pixel-major, channels innermost,
// three bf16 levels), and the k loop reads fragments with ds_read_b64_tr_b16: the transposing LDS read hands a lane the four
// consecutive PIXELS (k) of one channel (row / column of the product) out of a [pixel][channel] image.
//
//  * Block = 4 waves, persistent, two per CU; a tile = 4 rows x 32 pixels of dY (48 output channels of chunk mc, zero padded)
//    and the same tile plus a 1-pixel halo of X (36 input channels of chunk nc): 3 x (6 x 34 x 72 B + 4 x 32 x 96 B) = 79 KB.
//  * GEMM per tile row (32 pixels = one MFMA depth):  dW[48][324] += dY^T[48][32] * Xcol[32][324], columns c = 36 tap + ci
//    (tap-major, so that the four consecutive columns a lane of the transposing read addresses are four consecutive channels
//    of ONE tap: one 8-byte piece of the pixel at that tap's row / column offset).  324 columns = 21 tiles; wave w owns
//    tiles w, w + 4, ...: 5 or 6 column tiles x 3 row tiles of accumulators (<= 72 registers) that live in registers for
//    the WHOLE launch -- a block writes one slab at the end, a second kernel sums the slabs in fixed order.
//  * Per tile row a wave reads 9 A fragments (3 row tiles x 3 levels, shared by all its column tiles) and 3 B fragments per
//    column tile, two transposing reads each, one column tile ahead of the MFMAs (counted lgkmcnt), 18 MFMAs per column tile.
// No packed-fp32 / SDWA instructions (hvi-cidnet_amd/build.py).
#include "common.h"

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int kWThreads = 256;
constexpr int kWTH = 4, kWTW = 32;
constexpr int kWMC = 48, kWNC = 36;                         // output channels / input channels per block
constexpr int kWXRow = (kWTW + 2) * kWNC * 2;               // 2448 B per staged X row (34 pixels x 36 channels)
constexpr int kWXLevel = (kWTH + 2) * kWXRow;               // 14688
constexpr int kWYPix = kWMC * 2;                            // 96 B per staged dY pixel
constexpr int kWYRow = kWTW * kWYPix;                       // 3072
constexpr int kWYLevel = kWTH * kWYRow;                     // 12288
constexpr int kWY0 = 3 * kWXLevel;                          // 44064
constexpr int kWLds = kWY0 + 3 * kWYLevel;                  // 80928 B: two blocks per CU
constexpr int kWCols = 9 * kWNC;                            // 324
constexpr int kWNTiles = (kWCols + 15) / 16;                // 21
constexpr int kWNTW = (kWNTiles + 3) / 4;                   // column tiles per wave (6; the last waves own 5)
static_assert(2 * kWLds <= 160 * 1024, "two blocks per CU");

struct W3Args {
  const float* dY; long dy_bs;
  const float* X; long x_bs;
  float* slabs;                        // [pair][block][48][324] (columns tap-major)
  int B, M, N, H, W;
  int tiles_x, tiles_y, mchunks, nchunks;
};

// exact three-way split of two fp32 values into packed bf16 pairs (lo half = a, hi half = b), round to nearest even
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const bf16x2 h0 = __builtin_convertvector(f32x2{a, b}, bf16x2);
  p0 = __builtin_bit_cast(unsigned, h0);
  const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xFFFF0000u);
  const bf16x2 h1 = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
  p1 = __builtin_bit_cast(unsigned, h1);
  const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xFFFF0000u);
  const bf16x2 h2 = __builtin_convertvector(f32x2{sa, sb}, bf16x2);
  p2 = __builtin_bit_cast(unsigned, h2);
}

// transposing LDS read: per 16-lane group, lane 4 q + p supplies the address of (row q, four 16-bit columns 4 p .. 4 p + 3);
// lane i receives column i of the four rows.  EXEC must be all ones (no divergence around these reads).
#define W3_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define W3_WAIT6(N, s)                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                   \
               : "+v"((s)[0]), "+v"((s)[1]), "+v"((s)[2]), "+v"((s)[3]), "+v"((s)[4]), "+v"((s)[5]))

__device__ __forceinline__ bf16x8 frag_of(u64 lo, u64 hi) {
  const uint4 q = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
  return __builtin_bit_cast(bf16x8, q);
}

// the six transposing reads of one column tile's B fragments (3 levels x 2 halves of the 32-pixel depth), tile row R
template <int R>
__device__ __forceinline__ void b_issue(u64 (&s)[6], unsigned addr) {
  W3_TR(s[0], addr, R * kWXRow);
  W3_TR(s[1], addr, R * kWXRow + 4 * kWNC * 2);
  W3_TR(s[2], addr, R * kWXRow + kWXLevel);
  W3_TR(s[3], addr, R * kWXRow + kWXLevel + 4 * kWNC * 2);
  W3_TR(s[4], addr, R * kWXRow + 2 * kWXLevel);
  W3_TR(s[5], addr, R * kWXRow + 2 * kWXLevel + 4 * kWNC * 2);
}

__global__ __launch_bounds__(kWThreads, 2) void conv3xw_kernel(W3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int H = a.H, W = a.W, M = a.M;
  const long HW = (long)H * W;
  const int mc = blockIdx.y / a.nchunks, nc = blockIdx.y - mc * a.nchunks;

  // this lane's transposing-read addresses: pixel 8 g + q of a tile row (+ 4 for the second half of the depth)
  const unsigned aA = (unsigned)(kWY0 + (8 * g + q) * kWYPix + 8 * p);          // + 32 mt + 4 * 96 h + level + row
  unsigned aB[kWNTW];
#pragma unroll
  for (int j = 0; j < kWNTW; ++j) {
    int c = (wave + 4 * j) * 16 + 4 * p;
    if (c >= kWCols) c = 0;                                  // past the last column: any valid piece (never stored)
    const int tap = c / kWNC, ci = c - tap * kWNC;
    const int dy = tap / 3, dx = tap - 3 * dy;
    aB[j] = (unsigned)(dy * kWXRow + (8 * g + q + dx) * (kWNC * 2) + ci * 2);   // + 4 * 72 h + level + row
  }

  f32x4 acc[kWNTW][3];
#pragma unroll
  for (int j = 0; j < kWNTW; ++j)
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[j][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned tiles_per_img = a.tiles_x * a.tiles_y;
  const unsigned ntiles = (unsigned)a.B * tiles_per_img;
  // the second half of the grid starts late so that the two blocks of a CU alternate staging and MFMA phases (conv3x.hip)
  if (blockIdx.x >= (gridDim.x >> 1) && gridDim.x > 1) {
    __builtin_amdgcn_s_sleep(64); __builtin_amdgcn_s_sleep(64);
  }

  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = (int)(tile / tiles_per_img), tr = (int)(tile - (unsigned)b * tiles_per_img);
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * kWTH, x0 = tx * kWTW;
    __syncthreads();                                         // the previous tile's fragment reads are done
    // ---- stage: unit = (row, 4-channel group, pixel quad): X with its halo (quads from x0 - 4), then dY ----
    {
      const float* xb = a.X + (long)b * a.x_bs + (long)nc * kWNC * HW;
      const float* yb = a.dY + (long)b * a.dy_bs;
      constexpr int XQ = 10, XCG = kWNC / 4, XU = (kWTH + 2) * XCG * XQ;          // 540
      constexpr int YQ = 8, YCG = kWMC / 4, YU = kWTH * YCG * YQ;                 // 384
      constexpr int ROUNDS = (XU + YU + kWThreads - 1) / kWThreads;
#pragma unroll 2
      for (int rnd = 0; rnd < ROUNDS; ++rnd) {
        const int u = tid + rnd * kWThreads;
        if (u >= XU + YU) continue;
        const bool isx = u < XU;
        const int uu = isx ? u : u - XU;
        const int nq = isx ? XQ : YQ, ncg = isx ? XCG : YCG;
        const int qq = uu % nq, t = uu / nq;
        const int cg = t % ncg, ry = t / ncg;
        const int gy = isx ? y0 - 1 + ry : y0 + ry;
        const int gx0 = isx ? x0 - 4 + 4 * qq : x0 + 4 * qq;
        const int ch0 = isx ? 4 * cg : mc * kWMC + 4 * cg;                        // first of the unit's four channels
        const int nch = isx ? 4 : min(4, M - ch0);                                // live channels (dY rows past M are zero)
        const bool row_in = gy >= 0 && gy < H;
        const float* src = (isx ? xb : yb) + (long)ch0 * HW + (long)(row_in ? gy : 0) * W;
        float v[4][4];                                       // [channel][pixel]
        if (row_in && gx0 >= 0 && gx0 + 3 < W && nch == 4) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 w4 = load4u(src + (long)c * HW + gx0);
            v[c][0] = w4[0]; v[c][1] = w4[1]; v[c][2] = w4[2]; v[c][3] = w4[3];
          }
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int gx = gx0 + j;
              v[c][j] = (row_in && gx >= 0 && gx < W && c < nch) ? src[(long)c * HW + gx] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int lc = isx ? 4 * qq + j - 3 : 4 * qq + j;  // column inside the staged tile
          if (lc < 0 || lc >= (isx ? kWTW + 2 : kWTW)) continue;
          unsigned p0a, p1a, p2a, p0b, p1b, p2b;
          split3_pair(v[0][j], v[1][j], p0a, p1a, p2a);
          split3_pair(v[2][j], v[3][j], p0b, p1b, p2b);
          unsigned char* dst = isx ? xs + ry * kWXRow + lc * (kWNC * 2) + cg * 8 : xs + kWY0 + ry * kWYRow + lc * kWYPix + cg * 8;
          const int lvl = isx ? kWXLevel : kWYLevel;
          *reinterpret_cast<uint2*>(dst) = uint2{p0a, p0b};
          *reinterpret_cast<uint2*>(dst + lvl) = uint2{p1a, p1b};
          *reinterpret_cast<uint2*>(dst + 2 * lvl) = uint2{p2a, p2b};
        }
      }
    }
    __syncthreads();

    // ---- k loop: one 32-pixel tile row per step ----
#define W3_ROW(R)                                                                                                    \
    {                                                                                                                \
      u64 af[3][6];                                                                                                  \
      _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) {                                                             \
        W3_TR(af[mt][0], aA, R * kWYRow + 32 * mt);                                                                  \
        W3_TR(af[mt][1], aA, R * kWYRow + 32 * mt + 4 * kWYPix);                                                     \
        W3_TR(af[mt][2], aA, R * kWYRow + 32 * mt + kWYLevel);                                                       \
        W3_TR(af[mt][3], aA, R * kWYRow + 32 * mt + kWYLevel + 4 * kWYPix);                                          \
        W3_TR(af[mt][4], aA, R * kWYRow + 32 * mt + 2 * kWYLevel);                                                   \
        W3_TR(af[mt][5], aA, R * kWYRow + 32 * mt + 2 * kWYLevel + 4 * kWYPix);                                      \
      }                                                                                                              \
      u64 s0[6], s1[6];                                                                                              \
      b_issue<R>(s0, aB[0]);                                                                                         \
      bf16x8 a0[3], a1[3], a2[3];                                                                                    \
      _Pragma("unroll") for (int j = 0; j < kWNTW; ++j) {                                                            \
        if (wave + 4 * j >= kWNTiles) break;                 /* wave-uniform */                                      \
        u64 (&cs)[6] = (j & 1) ? s1 : s0;                                                                            \
        u64 (&nx)[6] = (j & 1) ? s0 : s1;                                                                            \
        const bool more = j + 1 < kWNTW && wave + 4 * (j + 1) < kWNTiles;                                            \
        if (more) { b_issue<R>(nx, aB[j + 1 < kWNTW ? j + 1 : j]); W3_WAIT6(6, cs); } else { W3_WAIT6(0, cs); }      \
        if (j == 0) {                                         /* the A reads were issued before s0: they have landed */ \
          _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) {                                                         \
            asm volatile("" : "+v"(af[mt][0]), "+v"(af[mt][1]), "+v"(af[mt][2]), "+v"(af[mt][3]), "+v"(af[mt][4]), "+v"(af[mt][5])); \
            a0[mt] = frag_of(af[mt][0], af[mt][1]); a1[mt] = frag_of(af[mt][2], af[mt][3]); a2[mt] = frag_of(af[mt][4], af[mt][5]); \
          }                                                                                                          \
        }                                                                                                            \
        const bf16x8 b0 = frag_of(cs[0], cs[1]), b1 = frag_of(cs[2], cs[3]), b2 = frag_of(cs[4], cs[5]);            \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[mt], b0, acc[j][mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b1, acc[j][mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[mt], b2, acc[j][mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[mt], b0, acc[j][mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[mt], b1, acc[j][mt], 0, 0, 0); \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[mt], b0, acc[j][mt], 0, 0, 0); \
      }                                                                                                              \
    }
    W3_ROW(0) W3_ROW(1) W3_ROW(2) W3_ROW(3)
#undef W3_ROW
  }

  // ---- this block's partial dW of the (mc, nc) chunk pair: lane (n, g) holds rows 4 g + reg, column n of every tile ----
  float* slab = a.slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (kWMC * kWCols);
  const int n = lane & 15;
#pragma unroll
  for (int j = 0; j < kWNTW; ++j) {
    const int c = (wave + 4 * j) * 16 + n;
    if (wave + 4 * j >= kWNTiles || c >= kWCols) continue;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) slab[(mt * 16 + 4 * g + reg) * kWCols + c] = acc[j][mt][reg];
  }
}

// dW[m][ci][tap] = sum over the blocks of the chunk pair (mc, nc) of slab[m - 48 mc][36 tap + ci - 36 nc]; fixed order
__global__ __launch_bounds__(256) void conv3xw_reduce_kernel(const float* __restrict__ slabs, int nblk, int M, int N, int nchunks,
                                                             float* __restrict__ dW) {
  const int i = blockIdx.x * 256 + threadIdx.x;               // element of dW in memory order
  if (i >= M * N * 9) return;
  const int m = i / (N * 9), rem = i - m * (N * 9), ci = rem / 9, tap = rem - ci * 9;
  const int mc = m / kWMC, nc = ci / kWNC;
  const float* s = slabs + ((long)(mc * nchunks + nc) * nblk) * (kWMC * kWCols) + (m - mc * kWMC) * kWCols + tap * kWNC + (ci - nc * kWNC);
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int k = 0;
  for (; k + 3 < nblk; k += 4) {
    t0 += s[(long)k * (kWMC * kWCols)]; t1 += s[(long)(k + 1) * (kWMC * kWCols)];
    t2 += s[(long)(k + 2) * (kWMC * kWCols)]; t3 += s[(long)(k + 3) * (kWMC * kWCols)];
  }
  for (; k < nblk; ++k) t0 += s[(long)k * (kWMC * kWCols)];
  dW[i] = (t0 + t1) + (t2 + t3);
}

inline int w3_blocks_per_pair(int B, int M, int N, int H, int W) {
  const int pairs = ((M + kWMC - 1) / kWMC) * (N / kWNC);
  const long ntiles = (long)B * ((W + kWTW - 1) / kWTW) * ((H + kWTH - 1) / kWTH);
  long nb = 512 / pairs;                                      // persistent: two resident blocks per CU in total
  if (nb < 1) nb = 1;
  if (nb > ntiles) nb = ntiles;
  return (int)nb;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* input channels in whole 36-channel chunks (CIDNet's 36 / 72 / 144); planes at least one pixel quad wide */
int cidnet_conv3x3_wgrad_bf16x3_supported(int M, int N, int H, int W) { return M >= 1 && N >= 36 && N % 36 == 0 && H >= 1 && W >= 4; }

long cidnet_conv3x3_wgrad_bf16x3_ws_floats(int B, int M, int N, int H, int W) {
  if (!cidnet_conv3x3_wgrad_bf16x3_supported(M, N, H, W)) return 0;
  const long pairs = (long)((M + kWMC - 1) / kWMC) * (N / kWNC);
  return pairs * w3_blocks_per_pair(B, M, N, H, W) * (kWMC * kWCols);
}

int cidnet_conv3x3_wgrad_bf16x3(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, float* ws, long ws_floats,
                                int B, int M, int N, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(dY && X && dW && ws && B > 0 && M > 0 && N > 0 && H > 0 && W > 0);
  if (!cidnet_conv3x3_wgrad_bf16x3_supported(M, N, H, W)) return CIDNET_ERR_SHAPE;
  if (ws_floats < cidnet_conv3x3_wgrad_bf16x3_ws_floats(B, M, N, H, W)) return CIDNET_ERR_WS;
  W3Args a{dY, dy_bs, X, x_bs, ws, B, M, N, H, W, (W + kWTW - 1) / kWTW, (H + kWTH - 1) / kWTH, (M + kWMC - 1) / kWMC, N / kWNC};
  const int nblk = w3_blocks_per_pair(B, M, N, H, W);
  static bool attr = false;                                   // idempotent: raises the kernel's dynamic-LDS limit once
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3xw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kWLds);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3xw_kernel, dim3((unsigned)nblk, (unsigned)(a.mchunks * a.nchunks)), dim3(kWThreads), kWLds, s, a);
  CIDNET_LAUNCH_STATUS();
  const int ne = M * N * 9;
  hipLaunchKernelGGL(conv3xw_reduce_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, ws, nblk, M, N, a.nchunks, dW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
