// Dense 3x3 convolution (zero pad 1) with fp32 operands on the BF16 matrix cores: exact "bf16x3" split products.
// Second design (round 3); replaces round 2's conv3s.hip, which kept the weights in registers, gave every wave ONE
// output-channel tile and therefore re-read every activation fragment from LDS three times (LDS-bound, no faster than
// the fp32-MFMA kernel).
//
// Arithmetic.  An fp32 value is the exact sum of three bf16 values a = a0 + a1 + a2 (round-to-nearest splits, every
// remainder exact), and a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 ab): six v_mfma_f32_16x16x32_bf16
// (exact products, fp32 accumulation) per 32-deep k-block reproduce the fp32 product to within its own rounding.
// On gfx950 the fp32 MFMA runs at the VALU's 64 FLOP/clk/SIMD while the bf16 MFMA runs at 1024: six products cost
// 3/8 of the fp32 instruction's time, on a pipe the VALU does not share.  Reference semantics: the dense 3x3 convs of
// NormDownsample / NormUpsample (net/transformer_utils.py:39,58), forward and data gradient (flip).
//
// GEMM view per (tile, 48-channel chunk, 36-channel input chunk):  Y[m][px] += sum_k A[m][k] B[k][px].
//  * The input tile ((8+2) x (32+2) pixels x KCH channels) is split once while it is staged into LDS, pixel-major with the
//    channels innermost and NO padding between pixels: [level][row][x][KCH] bf16.  For a fixed kernel row dy the operands
//    of one output pixel -- (dx = 0..2) x (KCH channels) -- are then 3*KCH CONSECUTIVE halfs, so k runs over them in 16-byte
//    groups ("slots": G = ceil(3 KCH / 8) per dy, 3 G in all, four slots per MFMA) and only the last group of a run is
//    padded: KCH = 36 -> 42 slots = 11 k-blocks for 324 real k (92 %), against 12 with per-tap padding.
//  * B fragment of lane (n, g): the 16 bytes of slot 4 kb + g of pixel n: two ds_read_b64 (pixels are 72 B apart, so only
//    8-byte alignment is guaranteed; issued as inline asm -- the compiler would merge the pair into a half-rate
//    ds_read2_b64 -- with counted lgkmcnt waits one N-tile ahead of the MFMAs).  The row pitch is chosen == 2 (mod 8)
//    dwords so that the two image rows of a wave's N-tile and the neighbouring slot of the lane group g+1 fall on
//    disjoint banks.
//  * A fragments (weights) are split ONCE per call by conv3x_prep_kernel into MFMA fragment order in a workspace
//    (L2-resident: 101 KB per 48 x 36 chunk) and stream through registers one k-block ahead: 9 x 16 B per lane per
//    k-block (3 channel tiles x 3 levels), shared by the 72 MFMAs of the wave's four N-tiles.
//  * Every wave computes ALL THREE 16-channel tiles of the chunk for its 64 pixels (2 rows x 32): a B fragment feeds 18
//    MFMAs.  N-tile e of a wave holds pixel e of each of its 16 pixel quads, so a lane ends up with four consecutive
//    pixels of a channel row: float4 stores (and float4 loads of the optional addend).
//  * Blocks are persistent (two per CU, 73 KB of LDS each): one stages while the other computes; the second half of the
//    grid starts late so that the pairs do not run in lockstep (worth 3 %).
// Measured on MI355X (8 x 36 -> 36 x 400 x 600): 362-384 us against the fp32-MFMA kernel's 527-542 us; the k loop with
// everything but its 792 MFMAs per tile removed takes 240 us (profiles/r03_b_conv3x_ablation_single_wave_pipeline.txt):
// under bf16-MFMA load the chip sustains ~1.6 GHz, and 31 % of the issued MFMA work is padding (36 -> 48 channels, 324 ->
// 352 k).  A one-block-per-CU variant that pipelined everything inside one wave per SIMD (512 registers, double-buffered
// tiles, split and output stores interleaved with the MFMAs) measured 395 us: every stalled load or store stops that
// SIMD's only wave; loading a tile's activations in one batch instead of two rounds at a time measured slower too.
// No packed-fp32 / SDWA instructions: built with -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0.
#include "common.h"
#include "cidnet_hip.h"

namespace cidnet {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kXThreads = 256;
constexpr int kXTH = 8, kXTW = 32, kXPH = kXTH + 2, kXPW = kXTW + 2;
constexpr int kXMC = 48;                                    // output channels per block chunk (three 16-row tiles)

constexpr int x3_row_pitch(int kch) {
  int rp = kXPW * kch + (8 * ((3 * kch + 7) / 8) - 3 * kch);  // pixels + the overrun of the last pixel's dy-run
  rp = (rp + 3) / 4 * 4;
  while ((rp / 2) % 8 != 2) rp += 4;
  return rp;
}

template <int KCH>
struct X3 {
  static constexpr int G = (3 * KCH + 7) / 8;               // 16-byte groups per dy-run
  static constexpr int SLOTS = 3 * G;
  static constexpr int NKB = (SLOTS + 3) / 4;               // k-blocks (MFMAs deep) per input-channel chunk
  static constexpr int RP = x3_row_pitch(KCH);              // halfs per staged row
  static constexpr int LEVEL_BYTES = kXPH * RP * 2;
  static constexpr int LDS_BYTES = 3 * LEVEL_BYTES;         // three activation levels (XL = 3); XL levels in general
  static constexpr int lds_bytes(int xl) { return xl * LEVEL_BYTES; }
  static constexpr int FRAGS = NKB * 9;                     // uint4-per-lane fragments per (channel chunk, input chunk)
  static_assert(KCH % 4 == 0, "pixels must stay 8-byte aligned");
  static_assert(2 * LEVEL_BYTES + 2 * RP * 2 + G * 16 + 3 * KCH * 2 < 65536, "ds_read offset field");
};

struct X3Args {
  const float* X; long x_bs;
  const uint4* A;                      // split weights in fragment order (conv3x_prep_kernel)
  const float* R; long r_bs;           // optional addend
  float* Y; long y_bs;
  int B, M, K, H, W;
  int tiles_x, tiles_y, mchunks, kchunks;
  int stagger;                         // start delay of the second half of the grid, in units of s_sleep 64
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

// ---- weights -> fragment order --------------------------------------------------------------------------------------
// A[mc][kc][kb][mt * 3 + level][lane] (uint4 = 8 bf16): lane (r = lane & 15, g = lane >> 4) holds, for output channel
// m = 48 mc + 16 mt + r, the eight k of slot s = 4 kb + g: position 8 i + c of the dy-run (dy = s / G, i = s % G), i.e.
// tap (dy, dx = pos / KCH), input channel kc KCH + pos % KCH; zero past the run, past K's chunk or past M.
template <int KCH>
__device__ __forceinline__ void conv3x_prep_item(const float* __restrict__ Wt, long w_ms, long w_ks, int flip, int M, int K,
                                                 uint4* __restrict__ A, int total, int idx) {
  using T = X3<KCH>;
  if (idx >= total) return;
  const int lane = idx & 63;
  int f = idx >> 6;
  const int mt = f % 3; f /= 3;
  const int kb = f % T::NKB; f /= T::NKB;
  const int kchunks = K / KCH;
  const int kc = f % kchunks, mc = f / kchunks;
  const int r = lane & 15, g = lane >> 4;
  const int m = kXMC * mc + 16 * mt + r;
  const int s = 4 * kb + g;
  const int dy = s / T::G, i = s - dy * T::G;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int pos = 8 * i + c;
    const int dx = pos / KCH, ch = kc * KCH + (pos - dx * KCH);
    const int tap = 3 * dy + dx;
    const bool ok = s < T::SLOTS && pos < 3 * KCH && m < M;
    v[c] = ok ? Wt[(long)m * w_ms + (long)ch * w_ks + (flip ? 8 - tap : tap)] : 0.f;
  }
  uint4 o[3];
  split3_pair(v[0], v[1], o[0].x, o[1].x, o[2].x);
  split3_pair(v[2], v[3], o[0].y, o[1].y, o[2].y);
  split3_pair(v[4], v[5], o[0].z, o[1].z, o[2].z);
  split3_pair(v[6], v[7], o[0].w, o[1].w, o[2].w);
  uint4* dst = A + ((((long)mc * kchunks + kc) * T::NKB + kb) * 9 + 3 * mt) * 64 + lane;
  dst[0] = o[0]; dst[64] = o[1]; dst[128] = o[2];
}

template <int KCH>
__global__ __launch_bounds__(256) void conv3x_prep_kernel(const float* Wt, long w_ms, long w_ks, int flip, int M, int K,
                                                          uint4* A, int total) {
  conv3x_prep_item<KCH>(Wt, w_ms, w_ks, flip, M, K, A, total, blockIdx.x * 256 + threadIdx.x);
}

template <int KCH>
constexpr int conv3x_prep_total(int M, int K) { return ((M + kXMC - 1) / kXMC) * (K / KCH) * X3<KCH>::NKB * 3 * 64; }

// Many layers in ONE launch (see pwx_split_w_batch_kernel): row r of the table (8 x int64) = source pointer, destination
// pointer, M, K, w_ms, w_ks, first block of the row, flip.
constexpr int kPrepRow = 8;
template <int KCH>
__global__ __launch_bounds__(256) void conv3x_prep_batch_kernel(const long long* __restrict__ table, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[(long)mid * kPrepRow + 6] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* r = table + (long)lo * kPrepRow;
  const int M = (int)r[2], K = (int)r[3];
  conv3x_prep_item<KCH>(reinterpret_cast<const float*>(r[0]), r[4], r[5], (int)r[7], M, K, reinterpret_cast<uint4*>(r[1]),
                        conv3x_prep_total<KCH>(M, K), (int)(((long)blockIdx.x - r[6]) * 256 + threadIdx.x));
}

// ---- LDS fragment reads as inline asm (see header) ---------------------------------------------------------------------
typedef unsigned long long u64;
#define X3_DSREAD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

template <int KCH, int XL>
struct BSet { u64 v[2 * XL]; };        // [level][half]

template <int KCH, int XL>
__device__ __forceinline__ void b_issue(BSet<KCH, XL>& s, unsigned addr, const int e) {
  using T = X3<KCH>;
  // `e` is a compile-time constant at every call site (fully unrolled): the offsets fold into the instruction
  switch (e) {
#define X3_CASE(E)                                                     \
    case E:                                                            \
      X3_DSREAD(s.v[0], addr, E * KCH * 2);                            \
      X3_DSREAD(s.v[1], addr, E * KCH * 2 + 8);                        \
      if constexpr (XL > 1) {                                          \
        X3_DSREAD(s.v[2], addr, E * KCH * 2 + T::LEVEL_BYTES);         \
        X3_DSREAD(s.v[3], addr, E * KCH * 2 + T::LEVEL_BYTES + 8);     \
      }                                                                \
      if constexpr (XL > 2) {                                          \
        X3_DSREAD(s.v[4], addr, E * KCH * 2 + 2 * T::LEVEL_BYTES);     \
        X3_DSREAD(s.v[5], addr, E * KCH * 2 + 2 * T::LEVEL_BYTES + 8); \
      }                                                                \
      break;
    X3_CASE(0) X3_CASE(1) X3_CASE(2) X3_CASE(3)
#undef X3_CASE
  }
}

// wait until at most the NEXT set's reads (2 XL of them; `next` = false: none) are outstanding; the set's registers are
// "produced" here, so no consumer can be scheduled above the wait
template <int KCH, int XL>
__device__ __forceinline__ void b_wait(BSet<KCH, XL>& s, const bool next) {
  if constexpr (XL == 3) {
    if (next) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]));
  } else {
    static_assert(XL == 1, "activation levels: 1 or 3");
    if (next) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(s.v[0]), "+v"(s.v[1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s.v[0]), "+v"(s.v[1]));
  }
}

__device__ __forceinline__ bf16x8 frag_of(u64 lo, u64 hi) {
  const uint4 q = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
  return __builtin_bit_cast(bf16x8, q);
}

// last pixel quad of a row whose width is not a multiple of 4 (150- and 75-pixel rows): 1..3 pixels, kept out of line so
// that the float4 path stays branch-free
__device__ __attribute__((noinline)) void store_ragged(float* yp, const float* rp, float v0, float v1, float v2, int cnt) {
  yp[0] = v0 + (rp ? rp[0] : 0.f);
  if (cnt > 1) yp[1] = v1 + (rp ? rp[1] : 0.f);
  if (cnt > 2) yp[2] = v2 + (rp ? rp[2] : 0.f);
}

// WL / XL: bf16 levels of the weights / activations that enter the products (all pairs i + j <= 2, small terms first).
// (3, 3): six products = the fp32 product to fp32 rounding (the parity mode).  (1, 1): both operands rounded to nearest
// bf16, one product -- the arithmetic of a bf16 autocast conv with fp32 accumulation and fp32 output; the staging pass then
// converts instead of splitting (1 VALU op per pair instead of 11) and the tile takes a third of the LDS.  (3, 1): exact
// weights, rounded activations, three products.
// resident blocks per CU: 2 with three activation levels (73 KB of LDS, ~180 VGPRs); 4 with one (24 KB, ~116 VGPRs) -- that
// mode has a sixth of the matrix-core work and lives on loads in flight
constexpr int x3_blocks_per_cu(int xl) { return xl == 1 ? 4 : 2; }

template <int KCH, int WL, int XL>
__global__ __launch_bounds__(kXThreads, x3_blocks_per_cu(XL)) void conv3x_kernel(X3Args a) {
  using T = X3<KCH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];       // [XL][PH][RP] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, M = a.M;
  const long HW = (long)H * W;

  // the row pads (run overrun of a row's last pixel) are read with zero weights: they must hold finite values
  for (int i = tid; i < T::lds_bytes(XL) / 16; i += kXThreads) reinterpret_cast<uint4*>(xs)[i] = uint4{0u, 0u, 0u, 0u};

  // this lane's byte offset of slot 4 kb + g inside a pixel's three runs
  unsigned soff[T::NKB];
#pragma unroll
  for (int kb = 0; kb < T::NKB; ++kb) {
    int s = 4 * kb + g;
    if (s >= T::SLOTS) s = 0;                                  // empty slot: zero weights, any valid address
    const int dy = s / T::G, i = s - dy * T::G;
    soff[kb] = (unsigned)(dy * T::RP * 2 + i * 16);
  }
  // pixel quad n of this wave: rows 2 wave + (n >> 3), columns 4 (n & 7) .. + 3; N-tile e = pixel e of every quad
  const int qrow = 2 * wave + (n >> 3), qcol = 4 * (n & 7);
  const unsigned pbase = (unsigned)((qrow * T::RP + qcol * KCH) * 2);

  const unsigned tiles_per_img = a.tiles_x * a.tiles_y;
  const unsigned nwork = (unsigned)a.B * tiles_per_img * a.mchunks;
  // work ids are dealt so that the blocks of one XCD (ids 8 apart share an L2) walk neighbouring tiles
  const unsigned per_xcd = gridDim.x >> 3;
  const unsigned first = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  // Two blocks share a CU and alternate "stage" and "k loop" phases.  Started together they stay in lockstep (both stage,
  // then both share the matrix pipe): the second half of the grid -- the blocks that land in the CUs' second slots --
  // starts late, so that one block stages while the other computes.  The offset persists: a block's period does not
  // depend on its phase.
  if (blockIdx.x >= (gridDim.x >> 1) && a.stagger > 0) {
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(64);
  }

  for (unsigned work = first; work < nwork; work += gridDim.x) {
    const int mc = (int)(work % (unsigned)a.mchunks);
    const unsigned tile = work / (unsigned)a.mchunks;
    const int b = (int)(tile / tiles_per_img), tr = (int)(tile - (unsigned)b * tiles_per_img);
    const int ty = tr / a.tiles_x, tx = tr - ty * a.tiles_x;
    const int y0 = ty * kXTH, x0 = tx * kXTW;

    f32x4 acc[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < a.kchunks; ++kc) {
      __syncthreads();                                         // the previous chunk's / tile's fragment reads are done
      // ---- stage (8+2) x (32+2) pixels x KCH channels: unit = (row, 4-channel group, aligned pixel quad) ----
      {
        const float* xb = a.X + (long)b * a.x_bs + (long)kc * KCH * HW;
        constexpr int NQ = 10, NCG = KCH / 4, UNITS = kXPH * NCG * NQ;
        constexpr int ROUNDS = (UNITS + kXThreads - 1) / kXThreads;
#pragma unroll 2
        for (int rnd = 0; rnd < ROUNDS; ++rnd) {
          const int u = tid + rnd * kXThreads;
          if (u < UNITS) {
            const int q = u % NQ, t = u / NQ;
            const int cg = t % NCG, ry = t / NCG;
            const int gy = y0 - 1 + ry, gx0 = x0 - 4 + 4 * q;
            const bool row_in = gy >= 0 && gy < H;
            const float* src = xb + (long)(4 * cg) * HW + (long)(row_in ? gy : 0) * W;
            float v[4][4];                                   // [channel][pixel]
            if (row_in && gx0 >= 0 && gx0 + 3 < W && q > 0 && q < NQ - 1) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 w4 = load4u(src + (long)c * HW + gx0);
                v[c][0] = w4[0]; v[c][1] = w4[1]; v[c][2] = w4[2]; v[c][3] = w4[3];
              }
            } else {
              // halo quads (only pixel 3 of quad 0 and pixel 0 of quad 9 are part of the tile) and image borders
#pragma unroll
              for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int gx = gx0 + j, lc = 4 * q + j - 3;
                  const bool ok = row_in && gx >= 0 && gx < W && lc >= 0 && lc < kXPW;
                  v[c][j] = ok ? src[(long)c * HW + gx] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int lc = 4 * q + j - 3;                  // column inside the staged tile
              if (lc < 0 || lc >= kXPW) continue;
              unsigned char* dst = xs + (ry * T::RP + lc * KCH + 4 * cg) * 2;
              if constexpr (XL == 1) {
                const unsigned pa = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0][j], v[1][j]}, bf16x2));
                const unsigned pb = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2][j], v[3][j]}, bf16x2));
                *reinterpret_cast<uint2*>(dst) = uint2{pa, pb};
              } else {
                unsigned p0a, p1a, p2a, p0b, p1b, p2b;
                split3_pair(v[0][j], v[1][j], p0a, p1a, p2a);
                split3_pair(v[2][j], v[3][j], p0b, p1b, p2b);
                *reinterpret_cast<uint2*>(dst) = uint2{p0a, p0b};
                *reinterpret_cast<uint2*>(dst + T::LEVEL_BYTES) = uint2{p1a, p1b};
                *reinterpret_cast<uint2*>(dst + 2 * T::LEVEL_BYTES) = uint2{p2a, p2b};
              }
            }
          }
        }
      }
      __syncthreads();

      // ---- compute: 3 channel tiles x 4 N-tiles per wave, NKB k-blocks ----
      const uint4* Ab = a.A + ((long)mc * a.kchunks + kc) * (T::FRAGS * 64) + lane;
      // weight fragments of a k-block: [mt][level] in the prepared layout; only the first WL levels are fetched
      uint4 an[3][WL];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int l = 0; l < WL; ++l) an[mt][l] = Ab[(3 * mt + l) * 64];
      BSet<KCH, XL> s0, s1;
      b_issue<KCH, XL>(s0, pbase + soff[0], 0);
#pragma unroll
      for (int kb = 0; kb < T::NKB; ++kb) {
        bf16x8 af[3][WL];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
          for (int l = 0; l < WL; ++l) af[mt][l] = __builtin_bit_cast(bf16x8, an[mt][l]);
        if (kb + 1 < T::NKB) {
#pragma unroll
          for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int l = 0; l < WL; ++l) an[mt][l] = Ab[((kb + 1) * 9 + 3 * mt + l) * 64];
        }
        const unsigned addr = pbase + soff[kb];
        const unsigned addr_next = pbase + soff[kb + 1 < T::NKB ? kb + 1 : kb];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          BSet<KCH, XL>& cur = (e & 1) ? s1 : s0;
          BSet<KCH, XL>& nxt = (e & 1) ? s0 : s1;
          const bool more = e < 3 || kb + 1 < T::NKB;
          if (more) {
            if (e < 3) b_issue<KCH, XL>(nxt, addr, e + 1); else b_issue<KCH, XL>(nxt, addr_next, 0);
          }
          b_wait<KCH, XL>(cur, more);
          bf16x8 bl[XL];
#pragma unroll
          for (int l = 0; l < XL; ++l) bl[l] = frag_of(cur.v[2 * l], cur.v[2 * l + 1]);
          // smallest terms first (level sums 2, 1, 0); the three channel tiles interleave, so dependent MFMAs are three
          // issues apart
#pragma unroll
          for (int sum = 2; sum >= 0; --sum)
#pragma unroll
            for (int i = sum; i >= 0; --i) {                      // weight level i, activation level sum - i
              if (i >= WL || sum - i >= XL) continue;
#pragma unroll
              for (int mt = 0; mt < 3; ++mt) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][i], bl[sum - i], acc[mt][e], 0, 0, 0);
            }
        }
      }
    }

    // ---- epilogue: lane (n, g) holds rows 4 g + reg of each channel tile for the four pixels of quad n ----
    const int y = y0 + qrow, x = x0 + qcol;
    if (y < H && x < W) {
      const int cnt = min(4, W - x);
      const long o0 = (long)y * W + x;
      float* yb = a.Y + (long)b * a.y_bs + o0;
      const float* rb = a.R ? a.R + (long)b * a.r_bs + o0 : nullptr;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int m = kXMC * mc + 16 * mt + 4 * g + reg;
          if (m >= M) continue;
          float v0 = acc[mt][0][reg], v1 = acc[mt][1][reg], v2 = acc[mt][2][reg], v3 = acc[mt][3][reg];
          float* yp = yb + (long)m * HW;
          if (cnt == 4) {
            if (rb) {
              const f32x4 r4 = load4u(rb + (long)m * HW);
              v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
            }
            store4u(yp, f32x4{v0, v1, v2, v3});
          } else {
            store_ragged(yp, rb ? rb + (long)m * HW : nullptr, v0, v1, v2, cnt);
          }
        }
      }
    }
  }
}

template <int KCH>
int launch_conv3x_prep(const float* Wt, long w_ms, long w_ks, int flip, int M, int K, float* ws, long ws_floats, hipStream_t s) {
  const int total = conv3x_prep_total<KCH>(M, K);
  hipLaunchKernelGGL((conv3x_prep_kernel<KCH>), dim3((total + 255) / 256), dim3(256), 0, s, Wt, w_ms, w_ks, flip, M, K,
                     reinterpret_cast<uint4*>(ws), total);
  return CIDNET_OK;
}

template <int KCH, int WL, int XL>
int launch_conv3x(X3Args a, const float* wprep, hipStream_t s) {
  using T = X3<KCH>;
  a.mchunks = (a.M + kXMC - 1) / kXMC;
  a.kchunks = a.K / KCH;
  a.A = reinterpret_cast<const uint4*>(wprep);
  a.tiles_x = (a.W + kXTW - 1) / kXTW;
  a.tiles_y = (a.H + kXTH - 1) / kXTH;
  constexpr int lds_bytes = T::lds_bytes(XL);
  static LdsLimit lds;                                        // once per device: the kernel's dynamic-LDS limit
  if (const hipError_t e = lds.raise(reinterpret_cast<const void*>(&conv3x_kernel<KCH, WL, XL>), lds_bytes); e != hipSuccess) return (int)e;
  const long nwork = (long)a.B * a.tiles_x * a.tiles_y * a.mchunks;
  long nblk = 256 * x3_blocks_per_cu(XL);                     // persistent: every resident slot of the 256 CUs
  if (nblk > nwork) nblk = nwork;
  a.stagger = 2;                                              // ~8k cycles: about the length of a staging phase
  hipLaunchKernelGGL((conv3x_kernel<KCH, WL, XL>), dim3((unsigned)nblk), dim3(kXThreads), lds_bytes, s, a);
  return CIDNET_OK;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

/* input-channel counts the split-product kernel covers: whole 36-channel chunks (CIDNet's 36 / 72 / 144) */
int cidnet_conv3x3_bf16x3_supported(int M, int K) { return K >= 36 && K % 36 == 0 && M >= 1; }

long cidnet_conv3x3_bf16x3_ws_floats(int M, int K) {
  if (!cidnet_conv3x3_bf16x3_supported(M, K)) return 0;
  return (long)((M + kXMC - 1) / kXMC) * (K / 36) * X3<36>::FRAGS * 64 * 4;
}

/* the weight preparation alone: Wt -> ws (split, MFMA fragment order) */
int cidnet_conv3x3_bf16x3_prep(const float* Wt, long w_ms, long w_ks, int flip, float* ws, long ws_floats, int M, int K, void* stream) {
  CIDNET_CHECK_ARG(Wt && ws && M > 0 && K > 0);
  if (!cidnet_conv3x3_bf16x3_supported(M, K)) return CIDNET_ERR_SHAPE;
  if (ws_floats < cidnet_conv3x3_bf16x3_ws_floats(M, K)) return CIDNET_ERR_WS;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  const int rc = launch_conv3x_prep<36>(Wt, w_ms, w_ks, flip, M, K, ws, ws_floats, (hipStream_t)stream);
  if (rc != CIDNET_OK) return rc;
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

long cidnet_conv3x3_bf16x3_prep_blocks(int M, int K) {
  return cidnet_conv3x3_bf16x3_supported(M, K) ? (conv3x_prep_total<36>(M, K) + 255) / 256 : 0;
}

int cidnet_conv3x3_bf16x3_prep_batch(const long long* table, int n, long total_blocks, void* stream) {
  CIDNET_CHECK_ARG(table && n > 0 && total_blocks > 0);
  hipLaunchKernelGGL((conv3x_prep_batch_kernel<36>), dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table, n);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

/* the convolution with weights already prepared by cidnet_conv3x3_bf16x3_prep; w_levels / x_levels: bf16 levels of the
 * weights / activations that enter the products -- (3, 3) the fp32-exact six products, (3, 1) and (1, 1) the bf16 modes */
int cidnet_conv3x3_bf16x3_pre_lv(const float* X, long x_bs, const float* Wprep, const float* R, long r_bs, float* Y, long y_bs,
                                 int B, int M, int K, int H, int W, int w_levels, int x_levels, void* stream) {
  CIDNET_CHECK_ARG(X && Wprep && Y && B > 0 && M > 0 && K > 0 && H > 0 && W > 0);
  if (!cidnet_conv3x3_bf16x3_supported(M, K)) return CIDNET_ERR_SHAPE;
  CIDNET_CHECK_ARG((reinterpret_cast<uintptr_t>(Wprep) & 15) == 0);
  X3Args a{X, x_bs, nullptr, R, r_bs, Y, y_bs, B, M, K, H, W, 0, 0, 0, 0, 0};
  int rc;
  if (w_levels == 3 && x_levels == 3) rc = launch_conv3x<36, 3, 3>(a, Wprep, (hipStream_t)stream);
  else if (w_levels == 3 && x_levels == 1) rc = launch_conv3x<36, 3, 1>(a, Wprep, (hipStream_t)stream);
  else if (w_levels == 1 && x_levels == 1) rc = launch_conv3x<36, 1, 1>(a, Wprep, (hipStream_t)stream);
  else return CIDNET_ERR_ARG;
  if (rc != CIDNET_OK) return rc;
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_conv3x3_bf16x3_pre(const float* X, long x_bs, const float* Wprep, const float* R, long r_bs, float* Y, long y_bs, int B,
                              int M, int K, int H, int W, void* stream) {
  return cidnet_conv3x3_bf16x3_pre_lv(X, x_bs, Wprep, R, r_bs, Y, y_bs, B, M, K, H, W, 3, 3, stream);
}

int cidnet_conv3x3_bf16x3(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, const float* R, long r_bs,
                          float* Y, long y_bs, float* ws, long ws_floats, int B, int M, int K, int H, int W, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && ws && B > 0 && M > 0 && K > 0 && H > 0 && W > 0);
  const int rc = cidnet_conv3x3_bf16x3_prep(Wt, w_ms, w_ks, flip, ws, ws_floats, M, K, stream);
  if (rc != CIDNET_OK) return rc;
  return cidnet_conv3x3_bf16x3_pre(X, x_bs, ws, R, r_bs, Y, y_bs, B, M, K, H, W, stream);
}

}  // extern "C"
