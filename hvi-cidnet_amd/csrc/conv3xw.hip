// Weight gradient of the dense 3x3 convolution (zero pad 1) with fp32 operands on the BF16 matrix cores: exact "bf16x3" split
// products (arithmetic: conv3x.hip).  dW[co][ci][dy][dx] = sum over pixels of dY[co][y][x] X[ci][y + dy - 1][x + dx - 1]
// (autograd of net/transformer_utils.py:39,58).
//
// k = pixels.  A first attempt (profiles/r03_e_...rejected.txt) split the operands in the k loop, once per fragment: 44 VALU
// instructions per fragment that feeds only 18 MFMAs -- VALU-bound, slower than the fp32-MFMA kernel.  Here every element is
// split ONCE, while its tile is staged into LDS (pixel-major, channels innermost, three bf16 levels), and the k loop reads
// fragments with ds_read_b64_tr_b16: the transposing LDS read hands a lane the four consecutive PIXELS (k) of one channel
// out of a [pixel][channel] image.
//
//  * The product is arranged as  D[(dx, co)][(dy, ci)] += sum_x' dY[co][y][x' - dx + 1] X[ci][y + dy - 1][x']  with k = x'
//    running over the 32 columns of a tile row: the horizontal tap shifts the dY operand, the vertical tap the X operand.
//    For a chunk of 36 output x 36 input channels that is a 108 x 108 product = 7 x 7 MFMA tiles (49; rows = co and
//    columns = (tap, ci) would take 3 x 21 = 63 for the same chunk).  Four consecutive rows (columns) of the product are
//    four consecutive channels of ONE shift: one 8-byte piece of a staged pixel, which is what a lane of the transposing
//    read addresses.
//  * Block = 8 waves, persistent, one per CU; a tile = 4 rows x 32 pixels: X with a one-row halo above and below (6 x 32
//    pixels), dY with a one-pixel halo left and right (4 x 34), 80-byte pixels (36 bf16 + pad: the four 16-lane groups of a
//    transposing read then hit disjoint banks), three levels: 78,720 B per tile, TWO tile buffers.
//  * Software pipeline per tile: a thread first issues the global loads of ITS share of the next tile into registers, then
//    runs the MFMA loop of the current tile out of one buffer, then splits the loaded values and writes them to the other
//    buffer; one barrier per tile.  (With the staging and the MFMA phase in sequence -- the first version of this kernel,
//    two 4-wave blocks per CU -- the time was the SUM of the two phases, 420 us at 36->36 400x600, however the blocks
//    were staggered: profiles/r03_f_conv3xw_ablation_first_version.txt.)
//    The loads are inline assembly (see unit_load); half of the waves split / write BEFORE their MFMA loop, one tile
//    further ahead (see the period loop in the kernel).
//  * Wave w = (group w & 3, half w >> 2): the half selects tile rows {0,1} or {2,3} (a split of k), the group a
//    near-rectangular 12 / 12 / 12 / 13-tile part of the 7 x 7 product, so the four SIMDs carry equal MFMA work.  The
//    accumulators (<= 52 registers) live in registers for the WHOLE launch -- a wave writes one slab at the end, a second
//    kernel sums the slabs in fixed order into dW's (co, ci, dy, dx) layout.
// No packed-fp32 / SDWA instructions (hvi-cidnet_amd/build.py).
#include "common.h"
#include "cidnet_hip.h"

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int kWMmaWaves = 8, kWStageWaves = 4;              // consumer / producer waves of a block
constexpr int kWThreads = 64 * (kWMmaWaves + kWStageWaves);
constexpr int kWTH = 4, kWTW = 32;
constexpr int kWC = 36;                                     // channels per chunk on both sides
constexpr int kWPix = 80;                                   // bytes per staged pixel and level
constexpr int kWXRow = kWTW * kWPix;                        // 2560
constexpr int kWXLevel = (kWTH + 2) * kWXRow;               // 15360
constexpr int kWYCols = kWTW + 2;                           // 34
constexpr int kWYRow = kWYCols * kWPix;                     // 2720
constexpr int kWYLevel = kWTH * kWYRow;                     // 10880
constexpr int kWY0 = 3 * kWXLevel;                          // 46080
constexpr int kWBuf = kWY0 + 3 * kWYLevel;                  // 78720 B per tile buffer
constexpr int kWLds = 2 * kWBuf;                            // 157440
constexpr int kWDim = 3 * kWC;                              // 108 rows / columns of the chunk's product
constexpr int kWT = (kWDim + 15) / 16;                      // 7 tiles per side
constexpr int kWSlab = kWDim * kWDim;
static_assert(kWLds <= 160 * 1024, "two tile buffers per CU");
static_assert(2 * kWXLevel + (kWTH + 1) * kWXRow + 4 * kWPix < 65536, "ds offsets are 16-bit");

struct W3Args {
  const float* dY; long dy_bs;
  const float* X; long x_bs;
  float* slabs;                        // [pair][block][108][108]
  int B, M, N, H, W;
  int tiles_x, tiles_y, nchunks;
#ifdef CIDNET_DEBUG
  int dbg;                             // timing study: 1 stage only the first tile, 2 no fragment reads / MFMAs,
                                       // 8 loads without split / LDS writes (no switch may sit between a load and its wait)
#endif
};
#ifdef CIDNET_DEBUG
int g_w3_dbg = 0;
#define W3_DBG(bit) (a.dbg & (bit))
#else
#define W3_DBG(bit) 0
#endif

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
// wait until at most the NEXT fragment set's reads (2 LV of them; `next` = false: none) are outstanding
template <int LV>
__device__ __forceinline__ void frag_wait(u64 (&s)[2 * LV], const bool next) {
  if constexpr (LV == 3) {
    if (next) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]));
  } else {
    static_assert(LV == 1, "operand levels: 1 or 3");
    if (next) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(s[0]), "+v"(s[1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]));
  }
}

__device__ __forceinline__ bf16x8 frag_of(u64 lo, u64 hi) {
  const uint4 q = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
  return __builtin_bit_cast(bf16x8, q);
}

// the transposing reads of one fragment set (LV levels x 2 halves of the 32-pixel depth) at byte offset OFF
template <int OFF, int LEVEL, int LV>
__device__ __forceinline__ void frag_issue(u64 (&s)[2 * LV], unsigned addr) {
  W3_TR(s[0], addr, OFF);
  W3_TR(s[1], addr, OFF + 4 * kWPix);
  if constexpr (LV > 1) {
    W3_TR(s[2], addr, OFF + LEVEL);
    W3_TR(s[3], addr, OFF + LEVEL + 4 * kWPix);
  }
  if constexpr (LV > 2) {
    W3_TR(s[4], addr, OFF + 2 * LEVEL);
    W3_TR(s[5], addr, OFF + 2 * LEVEL + 4 * kWPix);
  }
}

// One tile row R (of the wave's two) of a rectangular part of the product: row tiles RT0 .. RT0 + RTN - 1, column tiles
// CT0 .. CT0 + CTN - 1, accumulators acc[ACC0 + ct * RTN + rt].  A fragments (dY) stay in registers for the row, B fragments
// (X) are read one column tile ahead of their MFMAs.
template <int RT0, int RTN, int CT0, int CTN, int ACC0, int R, int LV>
__device__ __forceinline__ void part_row(f32x4 (&acc)[13], const unsigned (&aA)[kWT], const unsigned (&aB)[kWT]) {
  u64 af[RTN][2 * LV];
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) frag_issue<R * kWYRow, kWYLevel, LV>(af[rt], aA[RT0 + rt]);
  u64 s0[2 * LV], s1[2 * LV];
  frag_issue<R * kWXRow, kWXLevel, LV>(s0, aB[CT0]);
  bf16x8 al[RTN][LV];
#pragma unroll
  for (int ct = 0; ct < CTN; ++ct) {
    u64 (&cs)[2 * LV] = (ct & 1) ? s1 : s0;
    u64 (&nx)[2 * LV] = (ct & 1) ? s0 : s1;
    if (ct + 1 < CTN) frag_issue<R * kWXRow, kWXLevel, LV>(nx, aB[CT0 + (ct + 1 < CTN ? ct + 1 : ct)]);
    frag_wait<LV>(cs, ct + 1 < CTN);
    if (ct == 0) {                                           // issued before s0: landed
#pragma unroll
      for (int rt = 0; rt < RTN; ++rt) {
        if constexpr (LV == 3) asm volatile("" : "+v"(af[rt][0]), "+v"(af[rt][1]), "+v"(af[rt][2]), "+v"(af[rt][3]), "+v"(af[rt][4]), "+v"(af[rt][5]));
        else asm volatile("" : "+v"(af[rt][0]), "+v"(af[rt][1]));
#pragma unroll
        for (int l = 0; l < LV; ++l) al[rt][l] = frag_of(af[rt][2 * l], af[rt][2 * l + 1]);
      }
    }
    bf16x8 bl[LV];
#pragma unroll
    for (int l = 0; l < LV; ++l) bl[l] = frag_of(cs[2 * l], cs[2 * l + 1]);
    f32x4* c = &acc[ACC0 + ct * RTN];
    // small terms first: level sums 2, 1, 0 (a2 b0, a1 b1, a0 b2, a1 b0, a0 b1, a0 b0 with three levels; a0 b0 with one)
#pragma unroll
    for (int sum = 2; sum >= 0; --sum)
#pragma unroll
      for (int i = sum; i >= 0; --i) {
        if (i >= LV || sum - i >= LV) continue;
#pragma unroll
        for (int rt = 0; rt < RTN; ++rt) c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt][i], bl[sum - i], c[rt], 0, 0, 0);
      }
  }
}

// the four groups' parts of the 7 x 7 product: 4x3, 4x3, 3x4 and 3x3 + 4x1 tiles.  The group is a template parameter of the
// whole MFMA-wave loop: with the four variants in one loop body the register allocator shuffled the accumulators between
// the variants' assignments and spilled.
template <int GRP, int R, int LV>
__device__ __forceinline__ void group_row(f32x4 (&acc)[13], const unsigned (&aA)[kWT], const unsigned (&aB)[kWT]) {
  if (GRP == 0) {
    part_row<0, 4, 0, 3, 0, R, LV>(acc, aA, aB);
  } else if (GRP == 1) {
    part_row<0, 4, 3, 3, 0, R, LV>(acc, aA, aB);
  } else if (GRP == 2) {
    part_row<4, 3, 0, 4, 0, R, LV>(acc, aA, aB);
  } else {
    part_row<4, 3, 4, 3, 0, R, LV>(acc, aA, aB);
    part_row<0, 4, 6, 1, 9, R, LV>(acc, aA, aB);
  }
}

template <int RT0, int RTN, int CT0, int CTN, int ACC0>
__device__ __forceinline__ void part_store(const f32x4 (&acc)[13], float* slab, int n, int g) {
#pragma unroll
  for (int ct = 0; ct < CTN; ++ct) {
    const int c = 16 * (CT0 + ct) + n;
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * (RT0 + rt) + 4 * g + reg;
        if (r < kWDim && c < kWDim) slab[r * kWDim + c] = acc[ACC0 + ct * RTN + rt][reg];
      }
  }
}

// one staging unit = 4 channels x 4 consecutive pixels of one tile row: X rows y0 - 1 .. y0 + 4, columns x0 .. x0 + 31
// (8 quads); dY rows y0 .. y0 + 3, columns x0 - 4 .. x0 + 35 (10 quads, of which columns x0 - 1 .. x0 + 32 are staged)
constexpr int kWXUnits = (kWTH + 2) * (kWC / 4) * 8;        // 432
constexpr int kWYUnits = kWTH * (kWC / 4) * 10;             // 360
constexpr int kWStageThreads = 64 * kWStageWaves;
constexpr int kWRounds = (kWXUnits + kWYUnits + kWStageThreads - 1) / kWStageThreads;   // 4
struct Unit {
  int live, isx, ry, cg, col;          // col: first pixel of the quad relative to x0
};
__device__ __forceinline__ Unit unit_of(int u) {
  Unit t;
  t.live = u < kWXUnits + kWYUnits;
  t.isx = u < kWXUnits;
  const int uu = t.isx ? u : u - kWXUnits;
  const int nq = t.isx ? 8 : 10;
  const int qq = uu % nq, r = uu / nq;
  t.cg = r % (kWC / 4);
  t.ry = r / (kWC / 4);
  t.col = t.isx ? 4 * qq : 4 * qq - 4;
  return t;
}

// The global loads of a unit are inline assembly: compiler-generated loads make the waitcnt pass put "s_waitcnt vmcnt(0)"
// in front of the MFMA loop's (inline-assembly) LDS reads and between the two rounds -- the loads then do not stay in flight
// behind the MFMAs.  Every load is an unconditional dwordx4 from a clamped, always valid address; what the unit must not
// see is repaired after the wait (unit_fix): fix = 4 the quad lies outside the image (zero), fix = 1..3 the quad straddles
// the right edge and was loaded from column W - 4 instead (shift by fix pixels; only when W is not a multiple of 4).
__device__ __forceinline__ void unit_load(const Unit& t, const float* xb, const float* yb, int y0, int x0, int H, int W, long HW,
                                          f32x4 (&q)[4], int& fix) {
  const int gy = t.isx ? y0 - 1 + t.ry : y0 + t.ry;
  const int gx0 = x0 + t.col;                                // a multiple of 4: never straddles the left edge
  const bool out = gy < 0 || gy >= H || gx0 < 0 || gx0 >= W;
  const bool part = !out && gx0 + 3 >= W;
  fix = out ? 4 : part ? gx0 - (W - 4) : 0;
  const int sx = out ? 0 : part ? W - 4 : gx0, sy = out ? 0 : gy;
  const float* src = (t.isx ? xb : yb) + (long)(4 * t.cg) * HW + (long)sy * W + sx;
  if (t.live) {
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[c]) : "v"(src + (long)c * HW));
  }
}

// all loads issued so far have landed: every later use of the rounds' values depends on this statement
__device__ __forceinline__ void units_wait(f32x4 (&q)[kWRounds][4]) {
  static_assert(kWRounds == 4, "one operand per loaded register quad");
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]),
                 "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[2][3]), "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]), "+v"(q[3][3]));
}

template <int LV>
__device__ __forceinline__ void unit_write(const Unit& t, const f32x4 (&q)[4], int fix, unsigned char* buf, int dbg = 0) {
  if (!t.live) return;
  float v[16];                                               // [channel][pixel]
#pragma unroll
  for (int c = 0; c < 4; ++c) { v[4 * c] = q[c][0]; v[4 * c + 1] = q[c][1]; v[4 * c + 2] = q[c][2]; v[4 * c + 3] = q[c][3]; }
  if (fix != 0) {                                            // image border: rare
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float e1 = v[4 * c + 1], e2 = v[4 * c + 2], e3 = v[4 * c + 3];
      v[4 * c] = fix == 1 ? e1 : fix == 2 ? e2 : fix == 3 ? e3 : 0.f;
      v[4 * c + 1] = fix == 1 ? e2 : fix == 2 ? e3 : 0.f;
      v[4 * c + 2] = fix == 1 ? e3 : 0.f;
      v[4 * c + 3] = 0.f;
    }
  }
#ifdef CIDNET_DEBUG
  if (dbg & 8) {                                             // timing study: consume the loads with one store
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += v[i];
    *reinterpret_cast<float*>(buf + kWBuf - 4096 + (threadIdx.x & 63) * 4) = sum;
    return;
  }
#endif
  unsigned char* base = t.isx ? buf + t.ry * kWXRow + t.col * kWPix + t.cg * 8
                              : buf + kWY0 + t.ry * kWYRow + (t.col + 1) * kWPix + t.cg * 8;
  const int lvl = t.isx ? kWXLevel : kWYLevel;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int lc = t.isx ? t.col + j : t.col + j + 1;        // column inside the staged tile
    if (lc < 0 || lc >= (t.isx ? kWTW : kWYCols)) continue;
    unsigned char* dst = base + j * kWPix;
    if constexpr (LV == 1) {                                   // one level: round to nearest bf16 (same tile geometry, level 0 only)
      const unsigned pa = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[j], v[4 + j]}, bf16x2));
      const unsigned pb = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[8 + j], v[12 + j]}, bf16x2));
      *reinterpret_cast<uint2*>(dst) = uint2{pa, pb};
    } else {
      unsigned p0a, p1a, p2a, p0b, p1b, p2b;
      split3_pair(v[j], v[4 + j], p0a, p1a, p2a);
      split3_pair(v[8 + j], v[12 + j], p0b, p1b, p2b);
      *reinterpret_cast<uint2*>(dst) = uint2{p0a, p0b};
      *reinterpret_cast<uint2*>(dst + lvl) = uint2{p1a, p1b};
      *reinterpret_cast<uint2*>(dst + 2 * lvl) = uint2{p2a, p2b};
    }
  }
}

template <int GRP, int LV>
__device__ __forceinline__ void mma_waves(const W3Args& a, unsigned char* xs, int kh, int lane, int nt) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  // this lane's transposing-read addresses in tile buffer 0: k = pixel 8 g + q of a tile row (+ 4 for the second half)
  unsigned aA[kWT], aB[kWT];
#pragma unroll
  for (int t = 0; t < kWT; ++t) {
    int r4 = 16 * t + 4 * p4;
    if (r4 >= kWDim) r4 = 0;                                 // past the last row / column: any valid piece (never stored)
    const int sh = r4 / kWC, ch = r4 - sh * kWC;
    aA[t] = (unsigned)(kWY0 + 2 * kh * kWYRow + (8 * g + q + 2 - sh) * kWPix + ch * 2);   // dY column x' - dx + 1 (halo at 0)
    aB[t] = (unsigned)((2 * kh + sh) * kWXRow + (8 * g + q) * kWPix + ch * 2);            // X row y + dy - 1 (halo at 0)
  }
  f32x4 acc[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();                                           // periods -2 and -1: tile 0 is being staged
  __syncthreads();
  for (int p = 0; p < nt; ++p) {
    if (!W3_DBG(2)) {
      group_row<GRP, 0, LV>(acc, aA, aB);
      group_row<GRP, 1, LV>(acc, aA, aB);
    }
    if (!W3_DBG(1)) {                                        // the next tile is in the other buffer
      const unsigned d = (p & 1) ? (unsigned)-kWBuf : (unsigned)kWBuf;
#pragma unroll
      for (int t = 0; t < kWT; ++t) { aA[t] += d; aB[t] += d; }
    }
    __syncthreads();
  }

  // ---- the block's partial of the (mc, nc) chunk pair: the two k halves are added through LDS (fixed order), then lane
  // (n, g) of the first half's waves stores rows 4 g + reg, column n of every tile of its group ----
  __syncthreads();                                           // (kept for symmetry with the staging waves: LDS is free)
  f32x4* xch = reinterpret_cast<f32x4*>(xs) + (GRP * 13) * 64 + lane;
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < 13; ++i) xch[i * 64] = acc[i];
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int i = 0; i < 13; ++i) acc[i] += xch[i * 64];
  float* slab = a.slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * kWSlab;
  const int n = lane & 15;
  if (GRP == 0) {
    part_store<0, 4, 0, 3, 0>(acc, slab, n, g);
  } else if (GRP == 1) {
    part_store<0, 4, 3, 3, 0>(acc, slab, n, g);
  } else if (GRP == 2) {
    part_store<4, 3, 0, 4, 0>(acc, slab, n, g);
  } else {
    part_store<4, 3, 4, 3, 0>(acc, slab, n, g);
    part_store<0, 4, 6, 1, 9>(acc, slab, n, g);
  }
}

// LV: bf16 levels of BOTH operands that enter the products: 3 = six products, the fp32-exact parity mode; 1 = operands rounded
// to nearest bf16, one product (bf16 autocast arithmetic with fp32 accumulation; the staging waves convert instead of split)
template <int LV>
__global__ __launch_bounds__(kWThreads, 1) void conv3xw_kernel(W3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const long HW = (long)H * W;
  const int mc = blockIdx.y / a.nchunks, nc = blockIdx.y - mc * a.nchunks;
  const unsigned tiles_per_img = a.tiles_x * a.tiles_y;
  const unsigned ntiles = (unsigned)a.B * tiles_per_img;
  const unsigned G = gridDim.x;
  const int nt = (int)((ntiles - blockIdx.x + G - 1) / G);   // this block's tiles j = 0 .. nt - 1: tile blockIdx.x + j G, buffer j & 1

  // Periods p = -2 .. nt - 1, one barrier each.  In period p the MFMA waves run the MFMA loop of tile p; the staging waves
  // wait for the loads they issued in period p - 1, split / write tile p + 1 into the other buffer, and issue the loads of
  // tile p + 2, which then have a whole period to land.  The two roles are separate loops (disjoint register live ranges)
  // with the same number of barriers.
  if (wave >= kWMmaWaves) {
    // There is ONE static load site and its wait is unconditional, so the values pending across the loop edge have a
    // single definition: no copy of a load destination register before its wait (tests/test_abi.py checks the code object).
    Unit un[kWRounds];
    f32x4 pq[kWRounds][4];
    int fix[kWRounds];
    const float* xc = a.X + (long)nc * kWC * HW;
    const float* yc = a.dY + (long)mc * kWC * HW;
#pragma unroll
    for (int r = 0; r < kWRounds; ++r) un[r] = unit_of((tid - 64 * kWMmaWaves) + r * kWStageThreads);
    for (int p = -2; p < nt; ++p) {
      units_wait(pq);
      const bool wr = p + 1 >= 0 && p + 1 < nt && !(W3_DBG(1) && p + 1 > 0);
      if (wr) {
        unsigned char* wbuf = xs + (((p + 1) & 1) && !W3_DBG(1) ? kWBuf : 0);
#pragma unroll
        for (int r = 0; r < kWRounds; ++r) unit_write<LV>(un[r], pq[r], fix[r], wbuf, W3_DBG(8));
      }
      int jl = p + 2;
      jl = jl >= nt ? nt - 1 : jl;                           // past the block's range: a valid tile, loaded and not used
      const unsigned tile = blockIdx.x + (unsigned)jl * G;
      const int b = (int)(tile / tiles_per_img), tr = (int)(tile - (unsigned)b * tiles_per_img);
      const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
#pragma unroll
      for (int r = 0; r < kWRounds; ++r)
        unit_load(un[r], xc + (long)b * a.x_bs, yc + (long)b * a.dy_bs, ty * kWTH, tx * kWTW, H, W, HW, pq[r], fix[r]);
      __syncthreads();
    }
    units_wait(pq);                                          // the last (unused) loads
    __syncthreads();                                         // the two barriers of the MFMA waves' epilogue
    __syncthreads();
    return;
  }

  const int grp = wave & 3, kh = wave >> 2;
  if (grp == 0) mma_waves<0, LV>(a, xs, kh, lane, nt);
  else if (grp == 1) mma_waves<1, LV>(a, xs, kh, lane, nt);
  else if (grp == 2) mma_waves<2, LV>(a, xs, kh, lane, nt);
  else mma_waves<3, LV>(a, xs, kh, lane, nt);
}

// dW[co][ci][dy][dx] = sum over the nslab slabs of the chunk pair (co / 36, ci / 36) of slab[36 dx + co % 36][36 dy + ci % 36].
// A block sums 32 consecutive slab elements (coalesced reads): thread (e, sub) adds slabs sub, sub + 8, ..., the eight
// partial sums are then added in fixed order.
__global__ __launch_bounds__(256) void conv3xw_reduce_kernel(const float* __restrict__ slabs, int nslab, int npairs, int N, int nchunks,
                                                             float* __restrict__ dW) {
  __shared__ float part[8][32];
  const int e = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int E = blockIdx.x * 32 + e;
  const bool live = E < npairs * kWSlab;
  const int pair = live ? E / kWSlab : 0, j = live ? E - pair * kWSlab : 0;
  const float* s = slabs + (long)pair * nslab * kWSlab + j;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int k = sub;
  for (; k + 24 < nslab; k += 32) {
    t0 += s[(long)k * kWSlab]; t1 += s[(long)(k + 8) * kWSlab];
    t2 += s[(long)(k + 16) * kWSlab]; t3 += s[(long)(k + 24) * kWSlab];
  }
  for (; k < nslab; k += 8) t0 += s[(long)k * kWSlab];
  part[sub][e] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (sub != 0 || !live) return;
  float t = part[0][e];
#pragma unroll
  for (int i = 1; i < 8; ++i) t += part[i][e];
  const int r = j / kWDim, c = j - r * kWDim;
  const int dx = r / kWC, co = r - dx * kWC, dy = c / kWC, ci = c - dy * kWC;
  const int mc = pair / nchunks, nc = pair - mc * nchunks;
  dW[((long)(mc * kWC + co) * N + nc * kWC + ci) * 9 + dy * 3 + dx] = t;
}

inline int w3_blocks_per_pair(int B, int M, int N, int H, int W) {
  const int pairs = (M / kWC) * (N / kWC);
  const long ntiles = (long)B * ((W + kWTW - 1) / kWTW) * ((H + kWTH - 1) / kWTH);
  long nb = 256 / pairs;                                      // persistent: one resident block per CU in total
  if (nb < 1) nb = 1;
  if (nb > ntiles) nb = ntiles;
  return (int)nb;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* whole 36-channel chunks on both sides (CIDNet's 36 / 72 / 144); planes at least one pixel quad wide */
int cidnet_conv3x3_wgrad_bf16x3_supported(int M, int N, int H, int W) {
  return M >= kWC && M % kWC == 0 && N >= kWC && N % kWC == 0 && H >= 1 && W >= 4;
}

long cidnet_conv3x3_wgrad_bf16x3_ws_floats(int B, int M, int N, int H, int W) {
  if (!cidnet_conv3x3_wgrad_bf16x3_supported(M, N, H, W)) return 0;
  const long pairs = (long)(M / kWC) * (N / kWC);
  return pairs * w3_blocks_per_pair(B, M, N, H, W) * kWSlab;
}

int cidnet_conv3x3_wgrad_bf16x3(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, float* ws, long ws_floats,
                                int B, int M, int N, int H, int W, void* stream) {
  return cidnet_conv3x3_wgrad_bf16x3_lv(dY, dy_bs, X, x_bs, dW, ws, ws_floats, B, M, N, H, W, 3, stream);
}

/* levels: bf16 levels of both operands that enter the products: 3 (six products, fp32-exact) or 1 (one product) */
int cidnet_conv3x3_wgrad_bf16x3_lv(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, float* ws, long ws_floats,
                                   int B, int M, int N, int H, int W, int levels, void* stream) {
  CIDNET_CHECK_ARG(dY && X && dW && ws && B > 0 && M > 0 && N > 0 && H > 0 && W > 0 && (levels == 1 || levels == 3));
  if (!cidnet_conv3x3_wgrad_bf16x3_supported(M, N, H, W)) return CIDNET_ERR_SHAPE;
  if (ws_floats < cidnet_conv3x3_wgrad_bf16x3_ws_floats(B, M, N, H, W)) return CIDNET_ERR_WS;
  W3Args a{dY, dy_bs, X, x_bs, ws, B, M, N, H, W, (W + kWTW - 1) / kWTW, (H + kWTH - 1) / kWTH, N / kWC};
#ifdef CIDNET_DEBUG
  a.dbg = g_w3_dbg;
#endif
  const int nblk = w3_blocks_per_pair(B, M, N, H, W);
  static LdsLimit lds3, lds1;                                 // once per device: the kernels' dynamic-LDS limit
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nblk, (unsigned)((M / kWC) * (N / kWC)));
  if (levels == 3) {
    if (const hipError_t e = lds3.raise(reinterpret_cast<const void*>(&conv3xw_kernel<3>), kWLds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3xw_kernel<3>, grid, dim3(kWThreads), kWLds, s, a);
  } else {
    if (const hipError_t e = lds1.raise(reinterpret_cast<const void*>(&conv3xw_kernel<1>), kWLds); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv3xw_kernel<1>, grid, dim3(kWThreads), kWLds, s, a);
  }
  CIDNET_LAUNCH_STATUS();
  const int npairs = (M / kWC) * (N / kWC);
  hipLaunchKernelGGL(conv3xw_reduce_kernel, dim3((unsigned)((npairs * kWSlab + 31) / 32)), dim3(256), 0, s, ws, nblk, npairs, N, a.nchunks, dW);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

#ifdef CIDNET_DEBUG
void cidnet_debug_c3xw_flags(int flags) { g_w3_dbg = flags; }
#endif

}  // extern "C"
