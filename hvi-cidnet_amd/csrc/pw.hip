// K4: pointwise (1x1) convolution as an fp32-MFMA GEMM over NCHW planes -- forward, data gradient
// (same kernel, transposed weight strides), per-sample weights (the attention AV*project_out
// fold), fused epilogues (residual add; bilinear x2 add + PReLU of NormUpsample), and the weight
// gradient (split-K over pixels into slabs + fixed-order slab reduction).
//
// Reference call sites: net/LCA.py:13,15,17,22-23,40,51,57,61,66 and net/transformer_utils.py:60,66.
//
// GEMM view per sample:  Y[M x HW] = A[M x K] * X[K x HW],  HW contiguous.
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32).  Operand placement is chosen so that NO activation
// goes through LDS and every global access is 16 B per lane:
//   lane l = (c = l&15, j = l>>4) loads the float4 X[k0+j][p0+4c .. p0+4c+3]; element e of it is
//   the B operand (k = j, col = c) of MFMA #e, whose output columns are the pixels {p0+4c+e}.
//   After the K loop, register `reg` of the four accumulators e=0..3 holds Y[row][p0+4c+0..3]:
//   one float4 store per lane.  16 lanes cover 256 contiguous bytes of a plane row.
// Weights (A operand) are tiny and staged once per K-chunk in LDS, k-major with a leading
// dimension == 16 (mod 32) so the two k-rows of a 32-lane half hit disjoint banks.
#include "common.h"
#include "cidnet_hip.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr int kDepth = 4;     // k-steps of the activation operand prefetched per wave
// timing-study switches exist only in -DCIDNET_DEBUG builds; the shipped library has no mutable process state
#ifdef CIDNET_DEBUG
int g_pw_force_mt = 0;          // force the channel-tile count per block
long g_pw_target_blocks = 512;   // blocks a launch aims for (each walks several pixel tiles)
bool g_pw_target_set = false;    // cidnet_debug_pw_flags overrode it
int g_pw_dbg = 0;             // cidnet_debug_pw_flags: 1 no stores, 2 no K loop, 4 LDS kernel only
#else
constexpr int g_pw_force_mt = 0;
constexpr long g_pw_target_blocks = 512;
constexpr bool g_pw_target_set = false;
constexpr int g_pw_dbg = 0;
#endif

// element types of the activation operands as compile-time tags: the fp32 instantiations are exactly the round-1
// kernels (a run-time type switch cost the register-resident kernel 30-50 spilled VGPRs)
template <int XD, int YD> struct PwDT { static constexpr int X = XD, Y = YD; };

struct PwArgs {
  const void* X; long x_bs;           // element type xdt (CIDNET_F32 / CIDNET_BF16); strides in elements
  const float* Wt; long w_bs, w_ms, w_ks;
  void* Y; long y_bs;                 // element type ydt
  int xdt, ydt;
  const float* R; long r_bs;
  const float* Z; int zh, zw;
  const float* slope;
  float* Ypre;
  int M, K; long HW; int W;
  int kc, tpb, tile0, ntile_lim;   // K rows per LDS chunk; tiles per block; first tile; end tile (launcher)
  int dbg;              // ablation switches for kernel timing studies (0 in production)
};

__device__ __forceinline__ f32x4 load_px4(const float* row, long p, long HW, bool valid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid && p < HW) {
    if (p + 3 < HW) {
      v = load4u(row + p);
    } else {
      for (int e = 0; e < 4; ++e)
        if (p + e < HW) v[e] = row[p + e];
    }
  }
  return v;
}

// typed variants: `off` = element offset of the row inside the tensor `base` of element type dt
__device__ __forceinline__ f32x4 load_px4t(const void* base, long off, long p, long HW, bool valid, int dt) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid && p < HW) {
    if (p + 3 < HW) {
      v = ld4t(base, off + p, dt);
    } else {
      for (int e = 0; e < 4; ++e)
        if (p + e < HW) v[e] = ld1t(base, off + p + e, dt);
    }
  }
  return v;
}

__device__ __forceinline__ void store_px4t(void* base, long off, long p, long HW, int dt, f32x4 v) {
  if (p + 3 < HW) {
    st4t(base, off + p, dt, v);
  } else {
    for (int e = 0; e < 4; ++e)
      if (p + e < HW) st1t(base, off + p + e, dt, v[e]);
  }
}

__device__ __forceinline__ void store_px4(float* row, long p, long HW, f32x4 v) {
  if (p + 3 < HW) {
    store4u(row + p, v);
  } else {
    for (int e = 0; e < 4; ++e)
      if (p + e < HW) row[p + e] = v[e];
  }
}

struct UpTap {
  int o00, o01, o10, o11;
  float lx, ly;
};

// bilinear, align_corners=True, exactly x2 output (nn.UpsamplingBilinear2d, transformer_utils.py:59)
__device__ __forceinline__ UpTap up_tap(long p, int W, int zh, int zw) {
  const int H = 2 * zh;
  const int y = (int)(p / W), x = (int)(p - (long)y * W);
  const float sh = (H > 1) ? (float)(zh - 1) / (float)(H - 1) : 0.f;
  const float sw = (W > 1) ? (float)(zw - 1) / (float)(W - 1) : 0.f;
  const float fy = sh * (float)y, fx = sw * (float)x;
  const int y0 = (int)fy, x0 = (int)fx;
  UpTap t;
  t.ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f);
  t.lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
  const int y1 = y0 + (y0 < zh - 1 ? 1 : 0), x1 = x0 + (x0 < zw - 1 ? 1 : 0);
  t.o00 = y0 * zw + x0; t.o01 = y0 * zw + x1; t.o10 = y1 * zw + x0; t.o11 = y1 * zw + x1;
  return t;
}

// EPI: 0 plain, 1 + residual R, 2 + bilinear_x2(Z) then PReLU (writes optional pre-activation)
// A block stages its weight panel in LDS once (when all of K fits: `single`) and then walks `tpb`
// consecutive 256-pixel tiles.  TAIL = false is the streaming kernel: it requires HW % 4 == 0, so
// every lane's 4 pixels are all inside or all outside the plane; outside lanes load a clamped
// (valid) address and store nothing, hence no bounds code in the loop.  TAIL = true is the fully
// checked variant, launched only for the last tile of planes with HW % 4 != 0.
template <int MT, int EPI, bool TAIL, class DT>
__global__ __launch_bounds__(kThreads) void pw_conv_kernel(PwArgs a) {
  extern __shared__ float As[];
  constexpr int MB = 16 * MT;
  constexpr int ldA = (MT % 2 == 0) ? MB + 16 : MB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, j = lane >> 4;
  const int b = blockIdx.z;
  const int m0 = blockIdx.y * MB;
  const long HW = a.HW;
  const long xb0 = (long)b * a.x_bs;
  constexpr int xdt = DT::X, ydt = DT::Y;
  const float* Wb = a.Wt + (long)b * a.w_bs;
  const int kcmax = a.kc;
  const bool single = a.K <= kcmax;
  const long tile_beg = a.tile0 + (long)blockIdx.x * a.tpb;
  const long tile_end = min(tile_beg + a.tpb, (long)a.ntile_lim);

  // Weight panel -> LDS as [k][m].  Consecutive threads follow the unit-stride axis of the weight tensor (k for the
  // forward layout, m for the transposed / data-gradient layout) with 16 B loads where the strides allow it: a thread
  // takes 4 consecutive elements of a run.  (One 4 B load and a run-time division per element cost ~10 % of a block.)
  auto stage = [&](int kc0, int kcn, int kcn4) {
    if (a.w_ks == 1) {
      const int nv = kcn4 >> 2;                                   // float4 groups along k per output channel
      for (int i = tid; i < nv * MB; i += kThreads) {
        const int mm = i / nv, k0 = (i - mm * nv) << 2;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (m0 + mm < a.M) {
          const float* src = Wb + (long)(m0 + mm) * a.w_ms + (kc0 + k0);
          if (k0 + 3 < kcn) {
            const f32x4 q = load4u(src);
            x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (k0 + q < kcn) x[q] = src[q];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) As[(k0 + q) * ldA + mm] = x[q];
      }
    } else if (a.w_ms == 1) {
      const int nv = MB >> 2;                                     // float4 groups along m per k (MB is a multiple of 16)
      const int mvalid = a.M - m0;
      for (int i = tid; i < kcn4 * nv; i += kThreads) {
        const int kk = i / nv, mm0 = (i - kk * nv) << 2;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (kk < kcn) {
          const float* src = Wb + (long)(m0 + mm0) + (long)(kc0 + kk) * a.w_ks;
          if (mm0 + 3 < mvalid) {
            const f32x4 q = load4u(src);
            x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (mm0 + q < mvalid) x[q] = src[q];
          }
        }
        *reinterpret_cast<f32x4*>(&As[kk * ldA + mm0]) = f32x4{x[0], x[1], x[2], x[3]};
      }
    } else {
      for (int i = tid; i < kcn4 * MB; i += kThreads) {
        const int kk = i / MB, mm = i - kk * MB;
        float v = 0.f;
        if (kk < kcn && m0 + mm < a.M) v = Wb[(long)(m0 + mm) * a.w_ms + (long)(kc0 + kk) * a.w_ks];
        As[kk * ldA + mm] = v;
      }
    }
  };
  if (single) {
    stage(0, a.K, (a.K + 3) & ~3);
    __syncthreads();
  }

  for (long tile = tile_beg; tile < tile_end; ++tile) {
    const long p0 = tile * 256 + wave * 64 + 4 * c;
    const long pld = TAIL ? p0 : (p0 < HW - 4 ? p0 : HW - 4);     // clamped load position (streaming kernel)
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kc0 = 0; kc0 < ((a.dbg & 2) ? 0 : a.K); kc0 += kcmax) {
      const int kcn = min(kcmax, a.K - kc0);
      const int kcn4 = (kcn + 3) & ~3;
      if (!single) {
        __syncthreads();
        stage(kc0, kcn, kcn4);
        __syncthreads();
      }
      // Rows past K are clamped to a valid row: their A entries are zero-padded in LDS.
      const int klast = kcn - 1;
      if (!TAIL) {
        // kDepth k-steps of X stay in flight per wave (HBM latency >> the MFMA time of one k-step)
        const long xp = xb0 + (long)kc0 * HW + pld;
        f32x4 ring[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) ring[d] = ld4t(a.X, xp + (long)min(4 * d + j, klast) * HW, xdt);
#pragma unroll 1
        for (int kb = 0; kb < kcn4; kb += 4 * kDepth) {
#pragma unroll
          for (int d = 0; d < kDepth; ++d) {        // static ring slots: no register rotation, counted vmcnt
            const int k4 = kb + 4 * d;
            if (k4 >= kcn4) break;
            const f32x4 xc = ring[d];
            ring[d] = ld4t(a.X, xp + (long)min(k4 + 4 * kDepth + j, klast) * HW, xdt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const float av = As[(k4 + j) * ldA + mt * 16 + c];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc[e], acc[mt][e], 0, 0, 0);
            }
          }
        }
      } else {
        f32x4 xv = load_px4t(a.X, xb0 + (long)(kc0 + j) * HW, p0, HW, j < kcn, xdt);
        for (int k4 = 0; k4 < kcn4; k4 += 4) {
          const f32x4 xc = xv;
          if (k4 + 4 < kcn4) xv = load_px4t(a.X, xb0 + (long)(kc0 + k4 + 4 + j) * HW, p0, HW, k4 + 4 + j < kcn, xdt);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float av = As[(k4 + j) * ldA + mt * 16 + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc[e], acc[mt][e], 0, 0, 0);
          }
        }
      }
    }

    if (p0 >= HW) continue;
    if (a.dbg & 1) {                      // ablation: keep the accumulators alive, store nothing
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(acc[mt][e]));
      continue;
    }
    UpTap tap[4];
    float slope = 0.f;
    if (EPI == 2) {
      slope = a.slope[0];
#pragma unroll
      for (int e = 0; e < 4; ++e) tap[e] = up_tap(p0 + e < HW ? p0 + e : HW - 1, a.W, a.zh, a.zw);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + mt * 16 + j * 4 + reg;
        if (m >= a.M) continue;
        f32x4 v = {acc[mt][0][reg], acc[mt][1][reg], acc[mt][2][reg], acc[mt][3][reg]};
        if (EPI == 1) {
          const float* rrow = a.R + (long)b * a.r_bs + (long)m * HW;
          v += TAIL ? load_px4(rrow, p0, HW, true) : load4u(rrow + p0);
        }
        if (EPI == 2) {
          const float* z = a.Z + ((long)b * a.M + m) * ((long)a.zh * a.zw);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const UpTap& t = tap[e];
            const float top = (1.f - t.lx) * z[t.o00] + t.lx * z[t.o01];
            const float bot = (1.f - t.lx) * z[t.o10] + t.lx * z[t.o11];
            v[e] += (1.f - t.ly) * top + t.ly * bot;
          }
          if (a.Ypre) {
            float* prow = a.Ypre + (long)b * a.y_bs + (long)m * HW;
            if (TAIL) store_px4(prow, p0, HW, v); else store4u(prow + p0, v);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
        }
        const long yoff = (long)b * a.y_bs + (long)m * HW;
        if (TAIL) store_px4t(a.Y, yoff, p0, HW, ydt, v); else st4t(a.Y, yoff + p0, ydt, v);
      }
      // keep the scheduler from hoisting every accumulator read-out above the first store
      // (it would cost 16*MT extra VGPRs and halve the occupancy)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Register-resident weights: for small K (<= 4*KS) the whole A panel of a block (KS x MT
// fragments) lives in VGPRs, so the kernel has no LDS, no barrier and no per-k-step LDS latency; a
// block walks `tpb` pixel tiles and prefetches the next tile's first k-steps before it stores.
// Requires HW % 4 == 0 (streaming addressing, see pw_conv_kernel) -- the ragged tail tile of other
// planes goes through pw_conv_kernel<.., TAIL = true>.
// LEFT = 1: the block's last 1..4 output channels form a 4-row group computed with v_mfma_f32_4x4x1_16b_f32 instead of a
// padded 16-row tile (M = 36 = 2 tiles + 1 group: 288 instead of 384 MFMA cycles per k-step; same idiom as conv3.hip):
// the instruction's 16 blocks are (k-slot j) x (4 lanes), lane (c, j) feeds its own pixel values as B and
// W[16 MT + (c & 3)][4 ks + j] as A, so a 16-lane group accumulates the share of ITS k-slot; the four shares are added
// across the groups once per tile and lane (c, j) stores row 16 MT + j.
template <int MT, int EPI, int KS, class DT, int LEFT = 0>
__global__ __launch_bounds__(kThreads, ((EPI == 2 && !(MT + LEFT <= 3 && KS == 9)) || MT >= 5 ? 1 : 2)) void pw_conv_rega_kernel(PwArgs a) {
  constexpr int MB = 16 * MT + 4 * LEFT;
  // prefetch distance in k-steps: a whole 9-step tile ahead (~4600 MFMA cycles per wave, and the
  // co-resident wave doubles it) -- HBM latency under load is several thousand cycles
  constexpr int D = KS == 24 ? 8 : 9;
  static_assert(KS % D == 0, "ring slot of step s must be s % D in every tile");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, j = lane >> 4;
  const int b = blockIdx.z;
  const int m0 = blockIdx.y * MB;
  const long HW = a.HW;
  const long xb0 = (long)b * a.x_bs;
  constexpr int xdt = DT::X, ydt = DT::Y;
  const float* Wb = a.Wt + (long)b * a.w_bs;
  const long tile_beg = (long)blockIdx.x * a.tpb;
  const long tile_end = min(tile_beg + a.tpb, (long)a.ntile_lim);
  const int klast = a.K - 1;

  // The block's weight panel goes through LDS once: read from global along the unit-stride axis of the weight tensor
  // (coalesced), then each lane picks its KS x MT fragments.  Reading the fragments straight from global is a gather
  // of 64 separate 4-byte requests per load -- 11 us of address processing per block on the 72 -> 382 layer, more
  // than a pixel tile's MFMA time (tools/micro_pw.py, target-blocks sweep).
  constexpr int KP = 4 * KS, LDW = KP + 1;       // odd row stride: conflict-free column writes and fragment reads
  __shared__ float Ws[MB * LDW];
  {
    constexpr int TOT = MB * KP, NIT = (TOT + kThreads - 1) / kThreads;
    const bool k_contig = a.w_ks == 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * kThreads + (int)threadIdx.x;
      const int mm = k_contig ? idx / KP : idx % MB, kk = k_contig ? idx % KP : idx / MB;
      float v = 0.f;
      if (idx < TOT && m0 + mm < a.M && kk < a.K) v = Wb[(long)(m0 + mm) * a.w_ms + (long)kk * a.w_ks];
      if (idx < TOT) Ws[mm * LDW + kk] = v;
    }
  }
  __syncthreads();
  float areg[KS][MT];
  float al[LEFT ? KS : 1];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) areg[ks][mt] = Ws[(mt * 16 + c) * LDW + 4 * ks + j];
    if (LEFT) al[ks] = Ws[(MT * 16 + (c & 3)) * LDW + 4 * ks + j];
  }

  auto xrow = [&](long tile, int ks) -> long {
    const long p0 = tile * 256 + wave * 64 + 4 * c;
    const long pld = p0 < HW - 4 ? p0 : HW - 4;
    return xb0 + (long)min(4 * ks + j, klast) * HW + pld;
  };
  f32x4 ring[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < KS) ring[d] = ld4t(a.X, xrow(tile_beg, d), xdt);

  for (long tile = tile_beg; tile < tile_end; ++tile) {
    const long p0 = tile * 256 + wave * 64 + 4 * c;
    f32x4 acc[MT][4];
    f32x4 accl[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) accl[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 xc = ring[ks % D];
      if (!(a.dbg & 8)) {
        if (ks + D < KS) ring[ks % D] = ld4t(a.X, xrow(tile, ks + D), xdt);
        else if (tile + 1 < tile_end && ks + D - KS < KS) ring[ks % D] = ld4t(a.X, xrow(tile + 1, ks + D - KS), xdt);   // next tile's head
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[ks][mt], xc[e], acc[mt][e], 0, 0, 0);
      if (LEFT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) accl[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(al[ks], xc[e], accl[e], 0, 0, 0);
      }
    }
    if (LEFT) {                                      // add the four k-slot shares (all lanes take part)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = accl[e][q];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          accl[e][q] = v;
        }
    }

    if (p0 >= HW) continue;
    if (a.dbg & 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(acc[mt][e]));
      continue;
    }
    // x2 bilinear epilogue.  The lane's four pixels lie in one output row (W % 4 == 0, checked by the launcher), so they
    // share the two source rows and draw on at most four consecutive source columns: per output channel TWO 16-byte loads
    // of the low-resolution plane and a 4 x 4 horizontal weight matrix, instead of sixteen 4-byte gathers (the gathers
    // made this kernel run at 2.6 TB/s; 576 of them per wave and tile against 9 operand loads).
    float slope = 0.f, ly = 0.f, cw[4][4];
    long zo0 = 0, zo1 = 0;
    if (EPI == 2) {                                  // try_rega sends planes with W % 4 != 0 (150-pixel rows) to pw_conv_kernel
      slope = a.slope[0];
      const int W = a.W, zh = a.zh, zw = a.zw, H = 2 * zh;
      const int y = (int)(p0 / W), x = (int)(p0 - (long)y * W);
      const float sh = (H > 1) ? (float)(zh - 1) / (float)(H - 1) : 0.f;
      const float sw = (W > 1) ? (float)(zw - 1) / (float)(W - 1) : 0.f;
      const float fy = sh * (float)y;
      const int y0 = (int)fy, y1 = y0 + (y0 < zh - 1 ? 1 : 0);
      ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f);
      const int start = min((int)(sw * (float)x), zw - 4);
      zo0 = (long)y0 * zw + start; zo1 = (long)y1 * zw + start;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float fx = sw * (float)(x + e);
        const int x0 = (int)fx, x1 = x0 + (x0 < zw - 1 ? 1 : 0);
        const float lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) cw[e][q] = (x0 - start == q ? 1.f - lx : 0.f) + (x1 - start == q ? lx : 0.f);
      }
    }
    auto emit = [&](int m, f32x4 v) __attribute__((always_inline)) {
      if (EPI == 1) v += load4u(a.R + (long)b * a.r_bs + (long)m * HW + p0);
      if (EPI == 2) {
        const float* z = a.Z + ((long)b * a.M + m) * ((long)a.zh * a.zw);
        const f32x4 zt = load4u(z + zo0), zb = load4u(z + zo1);
        const f32x4 zl = zt + ly * (zb - zt);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (cw[e][0] * zl[0] + cw[e][1] * zl[1]) + (cw[e][2] * zl[2] + cw[e][3] * zl[3]);
        if (a.Ypre) store4u(a.Ypre + (long)b * a.y_bs + (long)m * HW + p0, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
      }
      st4t(a.Y, (long)b * a.y_bs + (long)m * HW + p0, ydt, v);
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + mt * 16 + j * 4 + reg;
        if (m >= a.M) continue;
        emit(m, f32x4{acc[mt][0][reg], acc[mt][1][reg], acc[mt][2][reg], acc[mt][3][reg]});
      }
    }
    if (LEFT) {                                      // lane (c, j) stores row 16 MT + j of the 4-row group
      const int m = m0 + MT * 16 + j;
      if (m < a.M) {
        auto pick = [](f32x4 v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3])); };
        emit(m, f32x4{pick(accl[0], j), pick(accl[1], j), pick(accl[2], j), pick(accl[3], j)});
      }
    }
  }
}

template <int MT, int EPI, int KS, class DT, int LEFT = 0>
int launch_pw_rega(PwArgs a, int B, long nstream, hipStream_t s) {
  constexpr int MB = 16 * MT + 4 * LEFT;
  const long mblocks = (a.M + MB - 1) / MB;
  // 512 blocks are resident (two per CU).  Store-heavy, bandwidth-bound layers (M >= 2K, below ~25 FLOP/B: e.g. the
  // 36 -> 190 project_in at 200x300) run 15 % faster as about two rounds of shorter blocks; read-heavy and MFMA-bound
  // ones prefer one round (tools/micro_pw.py target-blocks sweeps)
  const bool store_heavy = a.M >= 2 * a.K && (long)a.M * a.K < 50L * (a.M + a.K);
  const long target = g_pw_target_set ? g_pw_target_blocks : (store_heavy ? 1024 : 512);
  long tpb = (nstream * mblocks * B + target - 1) / target;
  tpb = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
  a.tpb = (int)tpb;
  a.tile0 = 0;
  a.ntile_lim = (int)nstream;
  dim3 grid((unsigned)((nstream + tpb - 1) / tpb), (unsigned)mblocks, (unsigned)B);
  hipLaunchKernelGGL((pw_conv_rega_kernel<MT, EPI, KS, DT, LEFT>), grid, dim3(kThreads), 0, s, a);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

// -> true if a register-resident instantiation exists for (MT, K); launches it.  Register budget
// (2 waves/SIMD): MT <= 4 with 9 k-steps, MT <= 3 with 18 or 24; MT = 5 (65..80 output channels in ONE block, one block
// per CU) with 9 or 18 k-steps.
template <int MT, int EPI, class DT>
bool try_rega(const PwArgs& a, int B, long nstream, hipStream_t s, int* rc) {
  const int ks = (a.K + 3) / 4;
  if (EPI == 2 && ((a.W & 3) != 0 || a.zw < 4)) return false;   // its x2 epilogue wants a lane's four pixels in one row
  // all output channels in one block and the last tile holds 1..4 of them (M = 36): 4-row group instead of a padded tile
  if constexpr (MT >= 2 && MT <= 4) {
    const int rem = a.M - 16 * (MT - 1);
    if (a.M <= 16 * MT && rem >= 1 && rem <= 4 && !(a.dbg & 32)) {
      if (ks <= 9) { *rc = launch_pw_rega<MT - 1, EPI, 9, DT, 1>(a, B, nstream, s); return true; }
      if constexpr (MT <= 3) {
        if (ks <= 18) { *rc = launch_pw_rega<MT - 1, EPI, 18, DT, 1>(a, B, nstream, s); return true; }
        if (ks <= 24) { *rc = launch_pw_rega<MT - 1, EPI, 24, DT, 1>(a, B, nstream, s); return true; }
      }
    }
  }
  if constexpr (MT <= 5) {
    if (ks <= 9) { *rc = launch_pw_rega<MT, EPI, 9, DT>(a, B, nstream, s); return true; }
  }
  if constexpr (MT <= 3 || MT == 5) {
    if (ks <= 18) { *rc = launch_pw_rega<MT, EPI, 18, DT>(a, B, nstream, s); return true; }
  }
  if constexpr (MT <= 3) {
    if (ks <= 24) { *rc = launch_pw_rega<MT, EPI, 24, DT>(a, B, nstream, s); return true; }
  }
  return false;
}

// Split-K variant for small planes (e.g. 50x75 = 3750 pixels) with large K: too few 256-pixel tiles
// exist to fill 256 CUs, so a block takes ONE 64-pixel group and its four waves split the K loop
// (k-step s goes to wave s % 4).  Each wave keeps its quarter of the weight panel in registers
// (KSW x MT fragments) across the `tpb` pixel groups the block walks; per group the four partial
// accumulators are summed through LDS and each thread stores one float4.  No weight staging, no
// barrier inside the K loop.  EPI: 0 plain, 1 + residual.
template <int MT, int EPI, int KSW, class DT>
__global__ __launch_bounds__(kThreads) void pw_conv_splitk_kernel(PwArgs a) {
  extern __shared__ float red[];                 // [4 waves][MT*16][64]
  constexpr int MB = 16 * MT;
  constexpr int D = KSW >= 8 ? 8 : KSW;          // prefetch distance (k-steps of this wave)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, j = lane >> 4;
  const int b = blockIdx.z;
  const int m0 = blockIdx.y * MB;
  const long HW = a.HW;
  const long xb0 = (long)b * a.x_bs;
  constexpr int xdt = DT::X, ydt = DT::Y;
  const float* Wb = a.Wt + (long)b * a.w_bs;
  const long ngroups = (HW + 63) / 64;
  const long g_beg = (long)blockIdx.x * a.tpb;
  const long g_end = min(g_beg + a.tpb, ngroups);
  const int klast = a.K - 1;

  float areg[KSW][MT];
#pragma unroll
  for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + c, k = 4 * (wave + 4 * ks) + j;
      areg[ks][mt] = (m < a.M && k < a.K) ? Wb[(long)m * a.w_ms + (long)k * a.w_ks] : 0.f;
    }

  for (long grp = g_beg; grp < g_end; ++grp) {
    const long p0 = grp * 64 + 4 * c;
    const bool inside = p0 + 3 < HW;             // lanes straddling / past the end take checked loads
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto xload = [&](int ks) -> f32x4 {
      const long row = xb0 + (long)min(4 * (wave + 4 * ks) + j, klast) * HW;
      return inside ? ld4t(a.X, row + p0, xdt) : load_px4t(a.X, row, p0, HW, true, xdt);
    };
    f32x4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = xload(d);
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      const f32x4 xc = ring[ks % D];
      if (ks + D < KSW) ring[ks % D] = xload(ks + D);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[ks][mt], xc[e], acc[mt][e], 0, 0, 0);
    }
    // cross-wave sum
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[((wave * MT * 16) + (mt * 4 + e) * 4 + reg) * 64 + lane] = acc[mt][e][reg];
    __syncthreads();
    const int reg = wave;                        // thread (reg = tid>>6, lane) owns rows j*4+reg of every m-tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + j * 4 + reg;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = ((mt * 4 + e) * 4 + reg) * 64 + lane;
        v[e] = (red[idx] + red[MT * 16 * 64 + idx]) + (red[2 * MT * 16 * 64 + idx] + red[3 * MT * 16 * 64 + idx]);
      }
      if (m < a.M && p0 < HW) {
        if (EPI == 1) v += load_px4(a.R + (long)b * a.r_bs + (long)m * HW, p0, HW, true);
        store_px4t(a.Y, (long)b * a.y_bs + (long)m * HW, p0, HW, ydt, v);
      }
    }
  }
}

template <int MT, int EPI, int KSW, class DT>
int launch_pw_splitk(PwArgs a, int B, hipStream_t s) {
  constexpr int MB = 16 * MT;
  const long mblocks = (a.M + MB - 1) / MB;
  const long ngroups = (a.HW + 63) / 64;
  // one round of the 512 resident blocks (was two): a block first gathers its weight fragments, and fewer, longer blocks
  // amortise that -- 2.51 -> 2.32 ms per step over the coarse-level launches.  (Staging the panel through LDS in 128-k
  // chunks instead of gathering it measured slower: 2.75 ms.)
  long tpb = (ngroups * mblocks * B + 511) / 512;
  tpb = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
  a.tpb = (int)tpb;
  dim3 grid((unsigned)((ngroups + tpb - 1) / tpb), (unsigned)mblocks, (unsigned)B);
  const size_t lds = (size_t)4 * MT * 16 * 64 * sizeof(float);
  hipLaunchKernelGGL((pw_conv_splitk_kernel<MT, EPI, KSW, DT>), grid, dim3(kThreads), lds, s, a);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

// small-plane / large-K dispatch; -> true if handled.  A wave can keep at most 24 k-steps x 3 channel
// tiles of weights in registers at 2 waves/SIMD, i.e. K <= 384 per launch: larger K runs as two launches,
// the second accumulating into Y through the residual epilogue.
template <int EPI, class DT>
int splitk_one(const PwArgs& a, int B, hipStream_t s) {
  const int ksw = (a.K + 15) / 16;               // k-steps per wave
  const int T = (a.M + 15) / 16;
  const int MT = T >= 3 ? 3 : T;
#define SK(mt, kk) return launch_pw_splitk<mt, EPI, kk, DT>(a, B, s);
  if (MT == 1) { if (ksw <= 9) SK(1, 9) if (ksw <= 18) SK(1, 18) SK(1, 24) }
  if (MT == 2) { if (ksw <= 9) SK(2, 9) if (ksw <= 18) SK(2, 18) SK(2, 24) }
  if (ksw <= 9) SK(3, 9) if (ksw <= 18) SK(3, 18) SK(3, 24)
#undef SK
}

template <int EPI, class DT>
bool try_splitk(const PwArgs& a, int B, hipStream_t s, int* rc) {
  // planes up to 8192 pixels (too few pixel tiles to fill the chip otherwise), and up to 16384 when K is too deep
  // for the LDS-resident weight panel (that kernel would re-stage the panel for every pixel tile)
  if (a.HW > 16384 || (a.HW > 8192 && a.K <= 320) || a.K < 64 || a.K > 768) return false;
  if (a.K > 384 && a.ydt != 0) return false;     // the two-launch form accumulates through an f32 Y
  if (a.K <= 384) { *rc = splitk_one<EPI, DT>(a, B, s); return true; }
  PwArgs lo = a, hi = a;
  lo.K = 384;
  *rc = splitk_one<EPI, DT>(lo, B, s);
  if (*rc != CIDNET_OK) return true;
  hi.K = a.K - 384;
  hi.X = a.xdt ? (const void*)((const bf16_t*)a.X + 384L * a.HW) : (const void*)((const float*)a.X + 384L * a.HW);
  hi.Wt = a.Wt + 384L * a.w_ks;
  hi.R = (const float*)a.Y; hi.r_bs = a.y_bs;    // accumulate: Y += A[:, 384:] * X[384:]  (Y is f32 here, see below)
  *rc = splitk_one<1, DT>(hi, B, s);
  return true;
}

template <int MT, int EPI, class DT>
int launch_pw_epi(PwArgs a, int B, hipStream_t s) {
  constexpr int MB = 16 * MT;
  constexpr int ldA = (MT % 2 == 0) ? MB + 16 : MB;
  const int kcmax = ((60 * 1024) / (ldA * 4)) & ~3;          // K rows that fit 60 KB of LDS
  const int k4 = (a.K + 3) & ~3;
  a.kc = k4 <= kcmax ? k4 : kcmax;
  const size_t lds = (size_t)a.kc * ldA * sizeof(float);
  const long ntiles = (a.HW + 255) / 256;
  const long mblocks = (a.M + MB - 1) / MB;
  const bool ragged = (a.HW % 4) != 0 || a.HW < 4;           // last tile needs the checked kernel
  const long nstream = ragged ? ntiles - 1 : ntiles;
  bool nstream_done = false;
  if (nstream > 0 && !(g_pw_dbg & 4)) {
    int rc = CIDNET_OK;
    if (try_rega<MT, EPI, DT>(a, B, nstream, s, &rc)) {
      if (rc != CIDNET_OK) return rc;
      nstream_done = true;
    }
  }
  if (nstream > 0 && !nstream_done) {
    long tpb = 1;
    if (a.K <= a.kc) {                                       // weights stay resident: walk several tiles
      tpb = (nstream * mblocks * B + g_pw_target_blocks - 1) / g_pw_target_blocks;
      tpb = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
    }
    a.tpb = (int)tpb;
    a.tile0 = 0;
    a.ntile_lim = (int)nstream;
    dim3 grid((unsigned)((nstream + tpb - 1) / tpb), (unsigned)mblocks, (unsigned)B);
    hipLaunchKernelGGL((pw_conv_kernel<MT, EPI, false, DT>), grid, dim3(kThreads), lds, s, a);
    CIDNET_LAUNCH_STATUS();
  }
  if (ragged) {
    a.tpb = 1;
    a.tile0 = (int)(ntiles - 1);
    a.ntile_lim = (int)ntiles;
    dim3 grid(1u, (unsigned)mblocks, (unsigned)B);
    hipLaunchKernelGGL((pw_conv_kernel<MT, EPI, true, DT>), grid, dim3(kThreads), lds, s, a);
    CIDNET_LAUNCH_STATUS();
  }
  return CIDNET_OK;
}

template <int MT, class DT>
int launch_pw(const PwArgs& a, int epi, int B, hipStream_t s) {
  if (epi == 0) return launch_pw_epi<MT, 0, DT>(a, B, s);
  if (epi == 1) return launch_pw_epi<MT, 1, DT>(a, B, s);
  if constexpr (DT::X == 0 && DT::Y == 0) return launch_pw_epi<MT, 2, DT>(a, B, s);      // the upsample epilogue is fp32 only
  return CIDNET_ERR_SHAPE;
}

template <class DT>
int dispatch_pw_dt(PwArgs a, int epi, int B, hipStream_t s) {
  a.dbg = g_pw_dbg;
  if (epi != 2 && !(g_pw_dbg & 16)) {
    int rc = CIDNET_OK;
    if (epi == 0 ? try_splitk<0, DT>(a, B, s, &rc) : try_splitk<1, DT>(a, B, s, &rc)) return rc;
  }
  const int T = (a.M + 15) / 16;
  const int ks = (a.K + 3) / 4;
  int MT;
  if (ks <= 24 && !(g_pw_dbg & 4)) {
    // register-resident weights: at most 4 (ks <= 9) or 3 channel tiles per block.  Every block of output channels
    // re-reads the whole input, so take the largest tile count that pads at most one tile in total
    // (tools/sweep_pw.py: M=72 runs 88 us with 3+2(+1 padded) tiles, 120 us as five single-tile blocks)
    const int mtmax = ks <= 9 ? 4 : 3;
    int best = 1, best_pad = 1 << 30;
    for (int mt = mtmax; mt >= 1; --mt) {        // least padding among those, the larger tile count on a tie (M = 36: 3, not 4)
      const int pad = ((T + mt - 1) / mt) * mt;
      if (pad <= T + 1 && pad < best_pad) { best = mt; best_pad = pad; }
    }
    MT = best;
    // 65..80 output channels (the 36 -> 72 kv conv, the 72 x 72 attention maps): all five tiles in one block at one block per
    // CU, so the input is read once instead of twice (3 + 2 tiles): 70 -> 60 us and 120 -> 85 us at 200x300.  Six tiles
    // (95 / 190 channels) and four tiles at 18 k-steps spill and measured slower.
    if (T == 5 && ks <= 18) MT = 5;
    if (g_pw_force_mt >= 1 && g_pw_force_mt <= mtmax) MT = g_pw_force_mt;
  } else {
    int nblk = (T + 5) / 6;
    MT = (T + nblk - 1) / nblk;
    // small planes (e.g. 50x75): shrink the channel tile until ~512 blocks exist (never below 2 tiles)
    const long ntiles = (a.HW + 255) / 256;
    while (MT > 2 && ntiles * B * ((T + MT - 1) / MT) < 512) --MT;
  }
  switch (MT) {
    case 1: return launch_pw<1, DT>(a, epi, B, s);
    case 2: return launch_pw<2, DT>(a, epi, B, s);
    case 3: return launch_pw<3, DT>(a, epi, B, s);
    case 4: return launch_pw<4, DT>(a, epi, B, s);
    case 5: return launch_pw<5, DT>(a, epi, B, s);
    default: return launch_pw<6, DT>(a, epi, B, s);
  }
}

// instantiated type pairs: all fp32; fp32 in / bf16 out (a hidden tensor is produced); bf16 in / fp32 out (consumed)
int dispatch_pw(const PwArgs& a, int epi, int B, hipStream_t s) {
  if (a.xdt == 0 && a.ydt == 0) return dispatch_pw_dt<PwDT<0, 0>>(a, epi, B, s);
  if (a.xdt == 0 && a.ydt == 1) return dispatch_pw_dt<PwDT<0, 1>>(a, epi, B, s);
  if (a.xdt == 1 && a.ydt == 0) return dispatch_pw_dt<PwDT<1, 0>>(a, epi, B, s);
  return CIDNET_ERR_SHAPE;
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dW[m][n] = sum_{b,p} dY[b][m][p] * X[b][n][p]
// Both MFMA operands are "row = channel, k = pixel", so both are plain float4 loads along the
// pixel axis; the k-slot permutation (lane j, element e) <-> pixel p + 8j + e is shared by A and B.
// Each wave owns a pixel sub-range and a private accumulator tile, written to its own slab.
// ---------------------------------------------------------------------------------------------
struct WgArgs {
  const void* dY; long dy_bs; int ddt;      // element types ddt / xdt (CIDNET_F32 / CIDNET_BF16)
  const void* X; long x_bs; int xdt;
  float* slabs;         // [B][chunks][M*N]
  int M, N; long HW; int pch;   // pixels per block (multiple of 128)
  int nnb;              // number of n-blocks
  int bf3;              // fp32 operands on the bf16 matrix cores (exact split products)
};

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));

// eight fp32 values (this lane's eight consecutive pixels of one row = the eight k of its lane group in a 16x16x32 MFMA)
// -> the row's fragment at three bf16 levels, v = l0 + l1 + l2 exactly (round to nearest, every remainder exact)
// one level: the eight values rounded to nearest bf16
__device__ __forceinline__ uint4 wg_round8(const f32x4& lo, const f32x4& hi) {
  auto pair = [](float x, float y) { return __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{x, y}, wg_bf16x2)); };
  return uint4{pair(lo[0], lo[1]), pair(lo[2], lo[3]), pair(hi[0], hi[1]), pair(hi[2], hi[3])};
}

__device__ __forceinline__ void wg_split8(const f32x4& lo, const f32x4& hi, uint4 (&f)[3]) {
  auto pair = [](float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    const wg_bf16x2 h0 = __builtin_convertvector(wg_f32x2{x, y}, wg_bf16x2);
    p0 = __builtin_bit_cast(unsigned, h0);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xFFFF0000u);
    const wg_bf16x2 h1 = __builtin_convertvector(wg_f32x2{rx, ry}, wg_bf16x2);
    p1 = __builtin_bit_cast(unsigned, h1);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xFFFF0000u);
    const wg_bf16x2 h2 = __builtin_convertvector(wg_f32x2{sx, sy}, wg_bf16x2);
    p2 = __builtin_bit_cast(unsigned, h2);
  };
  pair(lo[0], lo[1], f[0].x, f[1].x, f[2].x);
  pair(lo[2], lo[3], f[0].y, f[1].y, f[2].y);
  pair(hi[0], hi[1], f[0].z, f[1].z, f[2].z);
  pair(hi[2], hi[3], f[0].w, f[1].w, f[2].w);
}

// BF3: the products run on the BF16 matrix cores as six exact bf16 cross products per fp32 product (see conv3x.hip): a
// lane's eight consecutive pixels of a row ARE its 16x16x32 fragment, so the loads do not change; 6 MFMAs of 16 cycles
// replace 8 of 32 per tile pair and step, at 11 VALU instructions per loaded pixel pair for the split.
// LV = 3: that; LV = 1: both operands rounded to nearest bf16, ONE product (the bf16 mode: bf16 autocast arithmetic with fp32
// accumulation; any storage types); LV = 0: the fp32 MFMA.
template <int MT, int NT, class DT, int LV = 0>             // DT::X = type of dY, DT::Y = type of X
__global__ __launch_bounds__(kThreads) void pw_wgrad_kernel(WgArgs a) {
  extern __shared__ float red[];                 // [4 waves][MT*NT*4 regs][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, j = lane >> 4;
  const int b = blockIdx.z;
  const int mb = blockIdx.y / a.nnb, nb = blockIdx.y - mb * a.nnb;
  const int m0 = mb * 16 * MT, n0 = nb * 16 * NT;
  const long HW = a.HW;
  const long pbeg = (long)blockIdx.x * a.pch;
  const long pend = (pbeg + a.pch < HW) ? pbeg + a.pch : HW;
  constexpr int ddt = DT::X, xdt = DT::Y;
  long arow[MT], brow[NT];                  // element offsets of this lane's rows
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) arow[mt] = (long)b * a.dy_bs + (long)min(m0 + mt * 16 + r, a.M - 1) * HW;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) brow[nt] = (long)b * a.x_bs + (long)min(n0 + nt * 16 + r, a.N - 1) * HW;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Rows past M / N are clamped to a valid row: they only feed accumulator rows / columns that are
  // never stored.  Steps that lie fully inside [pbeg, pend) use unconditional 16 B loads.
  f32x4 av[MT][2], bv[NT][2];
  long p = pbeg + wave * 32;
  auto load_step = [&](long ps) {
    const long pl = ps + 8 * j;
    if (ps + 32 <= pend) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { av[mt][0] = ld4t(a.dY, arow[mt] + pl, ddt); av[mt][1] = ld4t(a.dY, arow[mt] + pl + 4, ddt); }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { bv[nt][0] = ld4t(a.X, brow[nt] + pl, xdt); bv[nt][1] = ld4t(a.X, brow[nt] + pl + 4, xdt); }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        av[mt][0] = load_px4t(a.dY, arow[mt], pl, pend, true, ddt);
        av[mt][1] = load_px4t(a.dY, arow[mt], pl + 4, pend, true, ddt);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bv[nt][0] = load_px4t(a.X, brow[nt], pl, pend, true, xdt);
        bv[nt][1] = load_px4t(a.X, brow[nt], pl + 4, pend, true, xdt);
      }
    }
  };
  if (p < pend) load_step(p);
  for (; p < pend; p += 128) {
    f32x4 ac[MT][2], bc[NT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { ac[mt][0] = av[mt][0]; ac[mt][1] = av[mt][1]; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { bc[nt][0] = bv[nt][0]; bc[nt][1] = bv[nt][1]; }
    if (p + 128 < pend) load_step(p + 128);       // prefetch this wave's next 32-pixel step
    if constexpr (LV == 1) {
      uint4 af[MT], bf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = wg_round8(ac[mt][0], ac[mt][1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = wg_round8(bc[nt][0], bc[nt][1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, af[mt]), __builtin_bit_cast(wg_bf16x8, bf[nt]),
                                                                acc[mt][nt], 0, 0, 0);
    } else if constexpr (LV == 3) {
      uint4 af[MT][3], bf[NT][3];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wg_split8(ac[mt][0], ac[mt][1], af[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wg_split8(bc[nt][0], bc[nt][1], bf[nt]);
      // smallest terms first; the tile pairs interleave, so dependent MFMAs are MT * NT issues apart
#define CIDNET_WG_TERM(AL, BL)                                                                                     \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)              \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, af[mt][AL]),            \
                                                            __builtin_bit_cast(wg_bf16x8, bf[nt][BL]), acc[mt][nt], 0, 0, 0)
      CIDNET_WG_TERM(2, 0);
      CIDNET_WG_TERM(1, 1);
      CIDNET_WG_TERM(0, 2);
      CIDNET_WG_TERM(1, 0);
      CIDNET_WG_TERM(0, 1);
      CIDNET_WG_TERM(0, 0);
#undef CIDNET_WG_TERM
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[mt][h][e], bc[nt][h][e], acc[mt][nt], 0, 0, 0);
    }
  }

  // sum the four waves' tiles through LDS, one slab per block
  constexpr int TILE = MT * NT * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) red[(wave * TILE + (mt * NT + nt) * 4 + reg) * 64 + lane] = acc[mt][nt][reg];
  __syncthreads();
  float* slab = a.slabs + (((long)b * gridDim.x + blockIdx.x) * (long)a.M) * a.N;
  for (int idx = threadIdx.x; idx < TILE * 64; idx += kThreads) {
    const float v = (red[idx] + red[TILE * 64 + idx]) + (red[2 * TILE * 64 + idx] + red[3 * TILE * 64 + idx]);
    const int l = idx & 63, q = idx >> 6;
    const int reg = q & 3, t = q >> 2;
    const int mt = t / NT, nt = t - mt * NT;
    const int m = m0 + mt * 16 + (l >> 4) * 4 + reg, n = n0 + nt * 16 + (l & 15);
    if (m < a.M && n < a.N) slab[(long)m * a.N + n] = v;
  }
}

// out[o][m*ld + n] (+)= sum_r slabs[(o*n_red + r)][m*N + n].  A block reduces 8 elements with 32 lanes each striding
// over the slabs (two independent partial sums per lane keep its loads in flight together: these are short,
// latency-bound launches), then folds the 32 partials through LDS in a fixed order (reproducible).
constexpr int kRedElems = 8;

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_red, int M, int N,
                                                           float* __restrict__ out, long out_os, long out_ld, int accumulate) {
  __shared__ float part[32][kRedElems + 1];
  const long ne = (long)M * N;
  const int ex = threadIdx.x & (kRedElems - 1), sl = threadIdx.x / kRedElems;
  const long i = (long)blockIdx.x * kRedElems + ex;
  const int o = blockIdx.y;
  float t0 = 0.f, t1 = 0.f;
  if (i < ne) {
    const float* s = slabs + (long)o * n_red * ne + i;
    int rr = sl;
    for (; rr + 32 < n_red; rr += 64) { t0 += s[(long)rr * ne]; t1 += s[(long)(rr + 32) * ne]; }
    if (rr < n_red) t0 += s[(long)rr * ne];
  }
  part[sl][ex] = t0 + t1;
  __syncthreads();
  if (sl == 0 && i < ne) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (part[4 * q][ex] + part[4 * q + 1][ex]) + (part[4 * q + 2][ex] + part[4 * q + 3][ex]);
    const float tot = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    const int m = (int)(i / N), n = (int)(i - (long)m * N);
    float* dst = out + (long)o * out_os + (long)m * out_ld + n;
    *dst = accumulate ? *dst + tot : tot;
  }
}

template <int MT, int NT, class DT>
int launch_wg_dt(const WgArgs& a, int chunks, int nmb, int B, hipStream_t s) {
  dim3 grid((unsigned)chunks, (unsigned)(nmb * a.nnb), (unsigned)B);
  if (a.bf3 == 1) hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT, DT, 1>), grid, dim3(kThreads), (size_t)MT * NT * 4 * 64 * 4 * sizeof(float), s, a);
  else hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT, DT>), grid, dim3(kThreads), (size_t)MT * NT * 4 * 64 * 4 * sizeof(float), s, a);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

// instantiated type pairs (dY, X): all fp32; a bf16 hidden tensor on either side
template <int MT, int NT>
int launch_wg(const WgArgs& a, int chunks, int nmb, int B, hipStream_t s) {
  if (a.ddt == 0 && a.xdt == 0 && a.bf3 == 3) {
    dim3 grid((unsigned)chunks, (unsigned)(nmb * a.nnb), (unsigned)B);
    hipLaunchKernelGGL((pw_wgrad_kernel<MT, NT, PwDT<0, 0>, 3>), grid, dim3(kThreads), (size_t)MT * NT * 4 * 64 * 4 * sizeof(float), s, a);
    CIDNET_LAUNCH_STATUS();
    return CIDNET_OK;
  }
  if (a.ddt == 0 && a.xdt == 0) return launch_wg_dt<MT, NT, PwDT<0, 0>>(a, chunks, nmb, B, s);
  if (a.ddt == 1 && a.xdt == 0) return launch_wg_dt<MT, NT, PwDT<1, 0>>(a, chunks, nmb, B, s);
  if (a.ddt == 0 && a.xdt == 1) return launch_wg_dt<MT, NT, PwDT<0, 1>>(a, chunks, nmb, B, s);
  return launch_wg_dt<MT, NT, PwDT<1, 1>>(a, chunks, nmb, B, s);
}

inline int pick_tiles(int dim, int maxt) {   // tiles per block for a dimension of `dim` channels
  const int T = (dim + 15) / 16;
  const int nblk = (T + maxt - 1) / maxt;
  return (T + nblk - 1) / nblk;
}

#ifdef CIDNET_DEBUG
int g_wg_force_pch = 0;          // timing studies (cidnet_debug_pw_flags bit 7 set: bits 8.. = pixels per block / 128)
#else
constexpr int g_wg_force_pch = 0;
#endif

// Pixels per block of the weight-gradient kernel (multiple of 128 = one step of the block's four waves).  A block costs
// its pixel steps plus about 6 steps' worth of prologue / LDS reduction / slab write, and 256 CUs hold 512 blocks at a
// time, so the launch costs about ceil(blocks / 512) * (steps + 6): take the cheapest of a few sizes.
inline int wgrad_pch(int B, int M, int N, long HW) {
  if (g_wg_force_pch > 0) return g_wg_force_pch;
  if (g_pw_dbg & 64) return HW >= 16384 ? 1024 : 512;            // previous fixed rule (A/B runs)
  const int MT = pick_tiles(M, 3), NT = pick_tiles(N, 3);
  const long per_chunk = (long)B * (((M + 15) / 16 + MT - 1) / MT) * (((N + 15) / 16 + NT - 1) / NT);
  int best = 512;
  long best_cost = -1;
  for (int pch : {512, 768, 1024, 1536, 2048, 3072, 4096}) {
    if (pch > 512 && pch > HW) break;
    const long blocks = per_chunk * ((HW + pch - 1) / pch);
    const long cost = ((blocks + 511) / 512) * (pch / 128 + 6);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = pch; }
  }
  return best;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

#ifdef CIDNET_DEBUG
void cidnet_debug_pw_flags(int flags) {
  g_pw_dbg = flags & 0xFF;
  g_pw_force_mt = (flags >> 28) & 7;
  if (flags & 128) { g_wg_force_pch = 128 * ((flags >> 8) & 0xFFFFF); return; }
  g_wg_force_pch = 0;
  g_pw_target_set = ((flags >> 8) & 0xFFFFF) != 0;
  g_pw_target_blocks = g_pw_target_set ? (flags >> 8) & 0xFFFFF : 512;
}
#endif

int cidnet_pw_conv_t(const void* X, int x_dt, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks, void* Y, int y_dt,
                     long y_bs, const float* R, long r_bs, int B, int M, int K, long HW, void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Y && B > 0 && M > 0 && K > 0 && HW > 0 && (x_dt | 1) == 1 && (y_dt | 1) == 1);
  PwArgs a{};
  a.X = X; a.x_bs = x_bs; a.Wt = Wt; a.w_bs = w_bs; a.w_ms = w_ms; a.w_ks = w_ks; a.Y = Y; a.y_bs = y_bs;
  a.xdt = x_dt; a.ydt = y_dt;
  a.R = R; a.r_bs = r_bs; a.M = M; a.K = K; a.HW = HW;
  return dispatch_pw(a, R ? 1 : 0, B, (hipStream_t)stream);
}

int cidnet_pw_conv(const float* X, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks, float* Y, long y_bs,
                   const float* R, long r_bs, int B, int M, int K, long HW, void* stream) {
  return cidnet_pw_conv_t(X, 0, x_bs, Wt, w_bs, w_ms, w_ks, Y, 0, y_bs, R, r_bs, B, M, K, HW, stream);
}

int cidnet_pw_conv_up_prelu(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, const float* Z,
                            const float* slope, float* Y, float* Ypre, int B, int M, int K, int zh, int zw,
                            void* stream) {
  CIDNET_CHECK_ARG(X && Wt && Z && slope && Y && B > 0 && M > 0 && K > 0 && zh > 0 && zw > 0);
  PwArgs a{};
  a.HW = 4L * zh * zw; a.W = 2 * zw;
  a.X = X; a.x_bs = x_bs; a.Wt = Wt; a.w_bs = 0; a.w_ms = w_ms; a.w_ks = w_ks; a.Y = Y; a.y_bs = (long)M * a.HW;
  a.Z = Z; a.zh = zh; a.zw = zw; a.slope = slope; a.Ypre = Ypre; a.M = M; a.K = K;
  return dispatch_pw(a, 2, B, (hipStream_t)stream);
}

long cidnet_pw_wgrad_ws_floats(int B, int M, int N, long HW) {
  const int pch = wgrad_pch(B, M, N, HW);
  const long chunks = (HW + pch - 1) / pch;
  return (long)B * chunks * M * N;
}

int cidnet_pw_wgrad_t(const void* dY, int dy_dt, long dy_bs, const void* X, int x_dt, long x_bs, float* dW, long dw_ld,
                      int per_sample, int flags, float* ws, long ws_floats, int B, int M, int N, long HW, void* stream);

int cidnet_pw_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, long dw_ld, int per_sample,
                    int flags, float* ws, long ws_floats, int B, int M, int N, long HW, void* stream) {
  return cidnet_pw_wgrad_t(dY, 0, dy_bs, X, 0, x_bs, dW, dw_ld, per_sample, flags, ws, ws_floats, B, M, N, HW, stream);
}

int cidnet_pw_wgrad_t(const void* dY, int dy_dt, long dy_bs, const void* X, int x_dt, long x_bs, float* dW, long dw_ld,
                      int per_sample, int flags, float* ws, long ws_floats, int B, int M, int N, long HW, void* stream) {
  CIDNET_CHECK_ARG(dY && X && dW && ws && B > 0 && M > 0 && N > 0 && HW > 0 && (dy_dt | 1) == 1 && (x_dt | 1) == 1);
  if (ws_floats < cidnet_pw_wgrad_ws_floats(B, M, N, HW)) return CIDNET_ERR_WS;
  WgArgs a{};
  a.dY = dY; a.dy_bs = dy_bs; a.ddt = dy_dt; a.X = X; a.x_bs = x_bs; a.xdt = x_dt; a.slabs = ws; a.M = M; a.N = N; a.HW = HW;
  a.pch = wgrad_pch(B, M, N, HW);
  a.bf3 = (flags & CIDNET_WGRAD_FP32_MFMA) ? 0 : (flags & CIDNET_WGRAD_BF16_1LEVEL) ? 1 : 3;   // operand levels (0: fp32 MFMA)
  const int accumulate = flags & CIDNET_WGRAD_ACCUMULATE;
  const int chunks = (int)((HW + a.pch - 1) / a.pch);
  const int MT = pick_tiles(M, 3), NT = pick_tiles(N, 3);
  const int nmb = ((M + 15) / 16 + MT - 1) / MT;
  a.nnb = ((N + 15) / 16 + NT - 1) / NT;
  hipStream_t s = (hipStream_t)stream;
  int rc;
#define WG_CASE(mt, nt) if (MT == mt && NT == nt) rc = launch_wg<mt, nt>(a, chunks, nmb, B, s); else
  WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(3, 1) WG_CASE(3, 2)
  WG_CASE(3, 3) rc = CIDNET_ERR_SHAPE;
#undef WG_CASE
  if (rc != CIDNET_OK) return rc;
  const long ne = (long)M * N;
  dim3 grid((unsigned)((ne + kRedElems - 1) / kRedElems), per_sample ? (unsigned)B : 1u);
  const int n_red = per_sample ? chunks : B * chunks;
  hipLaunchKernelGGL(reduce_slabs_kernel, grid, dim3(256), 0, s, ws, n_red, M, N, dW, (long)M * dw_ld, dw_ld, accumulate);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
