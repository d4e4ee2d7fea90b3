// K8: tile-resident IEL (gated feed-forward of an LCA block), forward.
//
//   out = [res +] W_out * ( (tanh(dw1(u1)) + u1) * (tanh(dw2(u2)) + u2) ),   [u1; u2] = dw(W_in * xn)
//
// Reference: IEL.forward, net/LCA.py:60-67 (project_in 1x1 C -> 2h, depthwise 3x3 on 2h, chunk, the two gated
// depthwise branches, project_out 1x1 h -> C) plus the residual of I_LCA (net/LCA.py:92).
//
// The unfused chain wrote and re-read three hidden tensors ((B,2h,H,W) twice, (B,h,H,W) once: 1.46 GB per launch at
// 200x300, h = 95).  Here a workgroup owns one TH x TW output tile and walks the hidden channels in chunks of 8 gate
// channels (= 16 rows of W_in: 8 of the first half, the matching 8 of the second half).  Per chunk:
//   S0  p  = W_in[chunk] * xn      on the (TH+4) x (TW+4) halo region   fp32 MFMA, xn straight from global / L2
//   S1  u  = dw(p)                 on (TH+2) x (TW+2), zero outside the image (it is dwconv1/2's zero padding)
//   S2  g  = (tanh(dw1 u1)+u1) * (tanh(dw2 u2)+u2) on TH x TW;  u written to HBM once (only the backward reads it)
//   S3  acc += W_out[:, chunk] * g                                      fp32 MFMA, accumulators live across chunks
// p, u and g only ever exist in LDS.  HBM traffic: xn in, out (and u when training) -- nothing else.
//
// Wave specialisation: 8 waves, two per SIMD.  Waves 0-3 ("GEMM waves") run S0 and S3 on the matrix pipe and own
// all output accumulators; waves 4-7 ("stencil waves") run S1 and S2 on the vector ALU.  The chunk loop is a
// two-stage software pipeline with two barriers per chunk:
//     part A:  S3(i-1) || S1(i)        part B:  S0(i+1) || S2(i)
// so every SIMD always has one wave feeding its matrix pipe and one feeding its VALU (the two pipes run
// concurrently, MI355X_MICROARCH.md "Two waves per SIMD"), and one LDS buffer each for p, u, g suffices: S1(i) has
// finished reading p(i) when S0(i+1) overwrites it, S3(i-1) has finished reading g(i-1) when S2(i) overwrites it.
//
// STATUS (round 2, measured on MI355X, tools/micro_iel.py / tools/iel_phases.py): results equal the unfused chain's
// (max |diff| <= 2e-6, u bit-close), but at 8x36x200x300 the kernel takes 500-570 us against the chain's 460 us: with
// fp32 operands the matrix pipe is the SIMD's own 32 fp32 lanes (64 FLOP/clk/SIMD for v_mfma_f32_16x16x4_f32 AND
// for v_fma_f32), so the MFMA stages and the stencil stages do not overlap -- a stencil wave alone issues every ~4
// cycles, next to an MFMA-streaming wave every ~10 -- and the halo recompute (1.6x on project_in) is paid on top.
// The uniform schedule (SPEC_ = false: all waves run all stages) measured 820 us.  The op layer therefore keeps the
// unfused chain by default (CIDNET_IEL_FUSED=1 switches the forward over); the kernel is the base of the bf16 path,
// where the MFMA runs on the separate, 16x faster matrix cores and the chain is purely HBM-bound.
//
// MFMA operand layout (as in pw.hip): v_mfma_f32_16x16x4_f32, A = weights (row = channel lane&15, k = lane>>4),
// B = activations.  Lane (c = lane&15, j = lane>>4) loads 4 consecutive pixels 4c..4c+3 of channel 4ks+j as one
// 16-byte access and feeds element e to MFMA e, so MFMA e covers pixels {4c+e}: the four results give each lane
// 4 channels x 4 CONSECUTIVE pixels, and every LDS / global access of the kernel moves 16 bytes per lane.
#include "common.h"

namespace cidnet {
namespace {

constexpr int kIelThreads = 512;
constexpr int kCP = 8;            // gate channels per chunk

// SPEC_: wave-specialised schedule (4 GEMM + 4 stencil waves) or the uniform one (all 8 waves run every stage)
template <int C_, int TH_, int TW_, bool SPEC_ = true>
struct IelT {
  static constexpr int C = C_, TH = TH_, TW = TW_;
  static constexpr bool SPEC = SPEC_;
  static constexpr int NGW = SPEC_ ? 4 : 8;            // waves that run the MFMA stages
  static constexpr int NSL = SPEC_ ? 256 : 512;        // lanes that run the stencil stages
  static_assert(C % 4 == 0 && TW % 4 == 0, "k-steps of 4 channels, 4-pixel lanes");
  static constexpr int KS = C / 4;                     // k-steps of project_in
  static constexpr int MT = (C + 15) / 16;             // output-channel tiles of project_out
  static constexpr int PH = TH + 4, PW = TW + 4, PN = PH * PW;
  static constexpr int NGP = (PN + 63) / 64;           // 64-pixel groups of the p region
  static constexpr int PS = NGP * 64 + 8;              // floats per channel in pbuf (+8: the 6-wide window of the last quad over-reads)
  static constexpr int UH = TH + 2, UW = TW + 4, UQ = UW / 4;   // u region, stored TW+4 wide (TW+2 used)
  static constexpr int US = UH * UW + 8;
  static constexpr int GN = TH * TW, NGO = (GN + 63) / 64, GS = NGO * 64;   // GS multiple of 64: conflict-free b128 operand reads
  static constexpr int NU = NGO * MT;                  // (pixel group, channel tile) units of project_out
  static constexpr int UPW = (NU + NGW - 1) / NGW;     // ... per GEMM wave
  static constexpr int PPW = (NGP + NGW - 1) / NGW;    // p groups per GEMM wave
  static constexpr int LDS_FLOATS = 2 * kCP * PS + 2 * kCP * US + kCP * GS;
};

#ifdef IEL_TIMING
// phase cycles of wave 0 (GEMM) and wave 4 (stencil) of every block: [S3, wait A, S0, wait B | S1, wait A, S2, wait B]
__device__ unsigned long long g_iel_phase[8 * 4096];
#define IEL_T0() unsigned long long t__ = __builtin_amdgcn_s_memtime()
#define IEL_TICK(slot)                                                            \
  do {                                                                            \
    const unsigned long long n__ = __builtin_amdgcn_s_memtime();                  \
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 4096) g_iel_phase[8 * blockIdx.x + (slot)] += n__ - t__; \
    t__ = n__;                                                                    \
  } while (0)
#else
#define IEL_T0()
#define IEL_TICK(slot)
#endif

struct IelArgs {
  const float* xn; const float* res; const float* w_in; const float* w_dw; const float* w_dw1; const float* w_dw2;
  const float* w_out; float* u; float* out;
  int B, h, H, W, tiles_x, tiles_y;
};

template <class T>
__global__ __launch_bounds__(kIelThreads, 2) void iel_fwd_kernel(IelArgs a) {
  constexpr int C = T::C, TH = T::TH, TW = T::TW, KS = T::KS, PW = T::PW, PS = T::PS, UW = T::UW, UQ = T::UQ, US = T::US,
                GS = T::GS, kNGW = T::NGW;
  constexpr bool SPEC = T::SPEC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* pbuf = lds;                          // [16][PS]   rows 0-7: first-half channels, 8-15: second half
  float* ubuf = pbuf + 2 * kCP * PS;          // [16][US]
  float* gbuf = ubuf + 2 * kCP * US;          // [8][GS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, j = lane >> 4;
  const int H = a.H, W = a.W, h = a.h;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const long HW = (long)H * W;
  const int nchunk = (h + kCP - 1) / kCP;
  const bool gemm = !SPEC || wave < kNGW;       // this wave runs the MFMA stages (the others, if any, the stencil stages)

  // ---------------- GEMM-wave state ----------------
  f32x4 acc[T::UPW][4];
  int poff[T::PPW];               // lane offset of this wave's p groups into xn[b]: k row j + clamped row + RAW quad column
  int pmask[T::PPW];              // bit e: element e lies inside the image (elsewhere p must be 0: the zero padding of dw)
  bool psafe[T::PPW];             // every lane's quad lies inside the xn tensor (always, except its very first / last row)
  bool pin[T::PPW];               // whole wave inside the image: nothing to mask
  if (gemm) {
#pragma unroll
    for (int u = 0; u < T::UPW; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[u][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long lo = -(long)b * C * HW, hi = (long)(a.B - b) * C * HW - 4;      // bounds of the whole tensor relative to xn[b]
#pragma unroll
    for (int i = 0; i < T::PPW; ++i) {
      const int g = wave + kNGW * i;
      const int q0 = 64 * g + 4 * c;
      const int ry = q0 / PW, rx = q0 - ry * PW;
      const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
      const bool rowok = q0 < T::PN && gy >= 0 && gy < H;
      const int gyc = min(max(gy, 0), H - 1);
      int m = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (rowok && gx + e >= 0 && gx + e < W) m |= 1 << e;
      pmask[i] = m;
      // Elements outside the image are read from wherever the raw column lands (the neighbouring row) and zeroed
      // after the MFMAs; only in-image elements must be read from their own address.
      poff[i] = j * (int)HW + gyc * W + gx;
      const long first = (long)poff[i], last = first + (long)(4 * (KS - 1)) * HW;
      psafe[i] = __all(first >= lo && last <= hi) != 0;
      pin[i] = __all(m == 15) != 0;
    }
  }

  // ======================= stencil stages =======================
    const int sid = SPEC ? tid - kNGW * 64 : tid;          // stencil lane 0 .. NSL-1
    // S1 (u = dw(p)): NSL/16 lanes per channel;  S2 (gate): NSL/8 lanes per gate channel
    constexpr int L1 = T::NSL / 16, L2 = T::NSL / 8;
    const int cl = sid / L1, t1 = sid % L1;
    const int pl = sid / L2, t2 = sid % L2;
    float w[9], w1[9], w2[9];
    // weights are fetched one stage ahead, so their (global / L2) latency hides behind the running stage
    auto ld_w = [&](int ch) {
      const int pr = kCP * ch + (cl & 7);
      const bool ok = pr < h;
      const float* src = a.w_dw + (long)(ok ? (cl < 8 ? 0 : h) + pr : 0) * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v = src[k];
        w[k] = ok ? v : 0.f;
      }
    };
    auto ld_w12 = [&](int ch) {
      const int pr = kCP * ch + pl;
      const bool ok = pr < h;
      const long o = (long)(ok ? pr : 0) * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v1 = a.w_dw1[o + k], v2 = a.w_dw2[o + k];
        w1[k] = ok ? v1 : 0.f;
        w2[k] = ok ? v2 : 0.f;
      }
    };
    // Both stencil stages are software-pipelined over their work items: the LDS window of item n+1 is requested before
    // item n is computed (one stencil wave per SIMD: nothing else would cover the LDS latency), two buffers alternate
    // in a loop unrolled by two so no register moves are spent on the hand-over.
    struct Win { f32x4 q[3]; float2 s[3]; };
    auto s1 = [&]() {
      const float* pc = pbuf + cl * PS;
      float* uc = ubuf + cl * US;
      constexpr int NIT = T::UH * UQ;
      auto load = [&](int it, Win& wv) {
        const int itc = min(it, NIT - 1);
        const int ry = itc / UQ, qx = itc - ry * UQ;
        const float* src = pc + ry * PW + 4 * qx;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          wv.q[dy] = *reinterpret_cast<const f32x4*>(src + dy * PW);
          wv.s[dy] = *reinterpret_cast<const float2*>(src + dy * PW + 4);
        }
      };
      auto comp = [&](int it, const Win& wv) {
        if (it >= NIT) return;
        const int ry = it / UQ, qx = it - ry * UQ;
        float r[3][6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          r[dy][0] = wv.q[dy][0]; r[dy][1] = wv.q[dy][1]; r[dy][2] = wv.q[dy][2]; r[dy][3] = wv.q[dy][3];
          r[dy][4] = wv.s[dy].x; r[dy][5] = wv.s[dy].y;
        }
        const int uy = y0 - 1 + ry, ux = x0 - 1 + 4 * qx;
        const bool rowok = uy >= 0 && uy < H;
        // tap-major: the four pixels' FMA chains advance together (a pixel-major expression compiles to one 9-deep
        // dependent chain after the other, and a single wave then issues at the FMA latency, not the issue rate)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = w[0] * r[0][e];
#pragma unroll
        for (int k = 1; k < 9; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(w[k], r[k / 3][e + k % 3], v[e]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (rowok && ux + e >= 0 && ux + e < W) ? v[e] : 0.f;      // u outside the image is dwconv1/2's zero padding
        *reinterpret_cast<f32x4*>(uc + ry * UW + 4 * qx) = o;
      };
      Win wa, wb;
      load(t1, wa);
      for (int it = t1; it < NIT; it += 2 * L1) {
        load(it + L1, wb);
        comp(it, wa);
        load(it + 2 * L1, wa);
        comp(it + L1, wb);
      }
    };
    auto s2 = [&](int ch) {
      const int pr = kCP * ch + pl;
      const float* ua = ubuf + pl * US;
      const float* ub = ubuf + (kCP + pl) * US;
      float* gc = gbuf + pl * GS;
      constexpr int TQ = TW / 4, NIT = TH * TQ;
      auto load = [&](int it, Win& wu, Win& wv) {
        const int itc = min(it, NIT - 1);
        const int ry = itc / TQ, qx = itc - ry * TQ;
        const int o = ry * UW + 4 * qx;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          wu.q[dy] = *reinterpret_cast<const f32x4*>(ua + o + dy * UW);
          wu.s[dy] = *reinterpret_cast<const float2*>(ua + o + dy * UW + 4);
          wv.q[dy] = *reinterpret_cast<const f32x4*>(ub + o + dy * UW);
          wv.s[dy] = *reinterpret_cast<const float2*>(ub + o + dy * UW + 4);
        }
      };
      auto comp = [&](int it, const Win& wu, const Win& wv) {
        if (it >= NIT) return;
        const int ry = it / TQ, qx = it - ry * TQ;
        float ra[3][6], rb[3][6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          ra[dy][0] = wu.q[dy][0]; ra[dy][1] = wu.q[dy][1]; ra[dy][2] = wu.q[dy][2]; ra[dy][3] = wu.q[dy][3];
          ra[dy][4] = wu.s[dy].x; ra[dy][5] = wu.s[dy].y;
          rb[dy][0] = wv.q[dy][0]; rb[dy][1] = wv.q[dy][1]; rb[dy][2] = wv.q[dy][2]; rb[dy][3] = wv.q[dy][3];
          rb[dy][4] = wv.s[dy].x; rb[dy][5] = wv.s[dy].y;
        }
        float sa[4], sb[4];               // tap-major over the 8 independent chains (see S1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sa[e] = w1[0] * ra[0][e]; sb[e] = w2[0] * rb[0][e]; }
#pragma unroll
        for (int k = 1; k < 9; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sa[e] = fmaf(w1[k], ra[k / 3][e + k % 3], sa[e]);
            sb[e] = fmaf(w2[k], rb[k / 3][e + k % 3], sb[e]);
          }
        f32x4 o, c1, c2;
        float ta[4], tb[4];
#pragma unroll
#ifdef IEL_EXP1
        for (int e = 0; e < 4; ++e) { ta[e] = sa[e]; tb[e] = sb[e]; }
#else
        for (int e = 0; e < 4; ++e) { ta[e] = tanh_fast(sa[e]); tb[e] = tanh_fast(sb[e]); }
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          c1[e] = ra[1][e + 1];
          c2[e] = rb[1][e + 1];
          o[e] = (ta[e] + c1[e]) * (tb[e] + c2[e]);
        }
        *reinterpret_cast<f32x4*>(gc + ry * TW + 4 * qx) = o;
        const int y = y0 + ry, x = x0 + 4 * qx;
        if (a.u != nullptr && pr < h && y < H && x < W) {
          float* u1 = a.u + ((long)b * 2 * h + pr) * HW + (long)y * W + x;
          float* u2 = u1 + (long)h * HW;
          if (x + 3 < W) {
            store4u(u1, c1);
            store4u(u2, c2);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (x + e < W) { u1[e] = c1[e]; u2[e] = c2[e]; }
          }
        }
      };
      Win ua0, ub0, ua1, ub1;
      load(t2, ua0, ub0);
      for (int it = t2; it < NIT; it += 2 * L2) {
        load(it + L2, ua1, ub1);
        comp(it, ua0, ub0);
        load(it + 2 * L2, ua0, ub0);
        comp(it + L2, ua1, ub1);
      }
    };
  // ======================= MFMA stages =======================
  float areg[KS];                           // W_in fragment of the chunk S0 is about to run
  float aw[T::MT][2];                       // W_out fragments of the chunk S3 is about to run (per output-channel tile)
  f32x4 qa[KS];                             // operand quads of the p group S0 runs next
  const float* xb = a.xn + (long)b * C * HW;       // wave-uniform base; lanes add a 32-bit offset
  const float* wr = nullptr;
  bool wok = false;
  auto ld_areg = [&](int ch) {
    const int pr = kCP * ch + (c & 7);
    wok = pr < h;
    wr = a.w_in + (long)((c < 8 ? 0 : h) + (wok ? pr : 0)) * C + j;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float v = wr[4 * ks];
      areg[ks] = wok ? v : 0.f;
    }
  };
  auto ld_aw = [&](int ch) {
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) {
      const int m = mt * 16 + c;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k = kCP * ch + 4 * ks + j;
        const bool ok = m < C && k < h;
        const float v = a.w_out[ok ? (long)m * h + k : 0];
        aw[mt][ks] = ok ? v : 0.f;
      }
    }
  };
  auto ld_q = [&](int i) {          // operand quads of p group i (safe groups; the others load in place)
    const int off = poff[i];
    if (psafe[i]) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qa[ks] = load4u(xb + (long)(4 * ks) * HW + off);
    }
  };
  // S0: p = W_in[chunk] * xn on the halo region -> pbuf.  Group 0's quads and areg are already in flight; the next
  // group's quads are requested right after a group's MFMAs have been issued (the matrix pipe then works through them
  // for ~32 cycles each while the loads travel) and before its results are masked and written.
  auto s0 = [&]() {
#pragma unroll
    for (int i = 0; i < T::PPW; ++i) {
      const int g = wave + kNGW * i;
      if (g >= T::NGP) break;
      f32x4 d[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (psafe[i]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[ks], qa[ks][e], d[e], 0, 0, 0);
      } else {
        // first / last row of the tensor: a raw quad would start before the allocation or end after it.  Per-element
        // loads at addresses clamped into the tensor (the clamped ones are masked elements), rolled loop, weights
        // re-read from memory: this path runs in a handful of waves per launch and must not cost registers.
        const long lo = -(long)b * C * HW, hi = (long)(a.B - b) * C * HW - 1;
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
          const float wv = wr[4 * ks];
          const float av = wok ? wv : 0.f;
          const long o = (long)(4 * ks) * HW + poff[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = xb[min(max(o + e, lo), hi)];
            d[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xe, d[e], 0, 0, 0);
          }
        }
      }
      if (i + 1 < T::PPW && wave + kNGW * (i + 1) < T::NGP) ld_q(i + 1);
      if (!pin[i]) {          // a select, not a product: whatever was read for the masked elements may be anything
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool keep = (pmask[i] >> e) & 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) d[e][r] = keep ? d[e][r] : 0.f;
        }
      }
      float* dst = pbuf + (4 * j) * PS + 64 * g + 4 * c;
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(dst + r * PS) = f32x4{d[0][r], d[1][r], d[2][r], d[3][r]};
    }
  };
  // S3: acc += W_out[:, chunk] * g
  auto s3 = [&]() {
#pragma unroll
    for (int u = 0; u < T::UPW; ++u) {
      const int unit = wave + kNGW * u;
      if (unit >= T::NU) break;
      const int g = unit % T::NGO, mt = unit / T::NGO;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gbuf + (4 * ks + j) * GS + 64 * g + 4 * c);
        float av = aw[0][ks];
#pragma unroll
        for (int q = 1; q < T::MT; ++q) av = mt == q ? aw[q][ks] : av;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[u][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, gv[e], acc[u][e], 0, 0, 0);
      }
    }
  };

  // ======================= schedules =======================
  if constexpr (SPEC) {
    // Two roles, separate loops with the same barrier sequence (s_barrier only counts arriving waves):
    //     part A:  S3(i-1) || S1(i)        part B:  S0(i+1) || S2(i)
    // Measured on MI355X (tools/iel_phases.py): the fp32 MFMA and the fp32 VALU do NOT overlap -- a stencil wave that
    // issues an instruction every ~4 cycles alone needs ~10 next to a wave streaming v_mfma_f32_16x16x4_f32 (both run
    // at 64 FLOP/clk/SIMD: the same lanes) -- so this schedule only pays with a matrix pipe that is separate (bf16).
    if (!gemm) {
      __builtin_amdgcn_s_setprio(2);
      ld_w(0);
      __syncthreads();
      IEL_T0();
      for (int i = 0; i < nchunk; ++i) {
        ld_w12(i);
        s1();
        IEL_TICK(4);
        __syncthreads();
        IEL_TICK(5);
        if (i + 1 < nchunk) ld_w(i + 1);
        s2(i);
        IEL_TICK(6);
        __syncthreads();
        IEL_TICK(7);
      }
      __syncthreads();
      __syncthreads();
      return;
    }
    ld_areg(0);
    ld_q(0);
    s0();
    __syncthreads();
    IEL_T0();
    for (int i = 0; i <= nchunk; ++i) {
      if (i + 1 < nchunk) {          // operands of S0(i+1): in flight while S3 runs and the barrier waits
        ld_areg(i + 1);
        ld_q(0);
      }
      if (i >= 1) s3();
      IEL_TICK(0);
      __syncthreads();
      IEL_TICK(1);
      if (i < nchunk) ld_aw(i);      // W_out fragments of S3(i), used after the next barrier
      if (i + 1 < nchunk) s0();
      IEL_TICK(2);
      __syncthreads();
      IEL_TICK(3);
    }
  } else {
    // Uniform: every wave runs S0 -> S1 -> S2 -> S3 of a chunk in turn (three barriers per chunk).  In a stencil stage
    // both waves of a SIMD issue vector instructions (2 cycles per wave-instruction instead of one wave's 4), in a GEMM
    // stage both feed the MFMA; each stage's weights are requested one stage ahead.
    ld_areg(0);
    ld_q(0);
    ld_w(0);
    s0();
    __syncthreads();
    IEL_T0();
    for (int i = 0; i < nchunk; ++i) {
      ld_w12(i);
      s1();
      IEL_TICK(4);
      __syncthreads();
      IEL_TICK(5);
      ld_aw(i);
      if (i + 1 < nchunk) { ld_w(i + 1); ld_areg(i + 1); }
      s2(i);
      IEL_TICK(6);
      __syncthreads();
      IEL_TICK(7);
      if (i + 1 < nchunk) ld_q(0);
      s3();
      IEL_TICK(0);
      if (i + 1 < nchunk) s0();      // pbuf was last read by S1(i), two barriers ago
      IEL_TICK(2);
      __syncthreads();
      IEL_TICK(3);
    }
  }

  // ---------------- epilogue: out = acc (+ res) ----------------
  {
#pragma unroll
    for (int u = 0; u < T::UPW; ++u) {
      const int unit = wave + kNGW * u;
      if (unit >= T::NU) break;
      const int g = unit % T::NGO, mt = unit / T::NGO;
      const int q0 = 64 * g + 4 * c;
      const int ry = q0 / TW, rx = q0 - ry * TW;
      const int y = y0 + ry, x = x0 + rx;
      if (q0 >= T::GN || y >= H || x >= W) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt * 16 + 4 * j + r;
        if (m >= C) continue;
        f32x4 v = {acc[u][0][r], acc[u][1][r], acc[u][2][r], acc[u][3][r]};
        const long o = ((long)b * C + m) * HW + (long)y * W + x;
        if (x + 3 < W) {
          if (a.res) v += load4u(a.res + o);
          store4u(a.out + o, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (x + e < W) a.out[o + e] = v[e] + (a.res ? a.res[o + e] : 0.f);
        }
      }
    }
  }
}

// blocks of this instantiation one CU can hold (registers, LDS and the 32-wave limit together): asked once
template <class T>
int iel_resident() {
  static int cached = 0;                  // idempotent query result, not configuration state
  static LdsLimit limit;                  // the attribute itself is per device
  constexpr size_t lds = sizeof(float) * T::LDS_FLOATS;
  (void)limit.raise(reinterpret_cast<const void*>(&iel_fwd_kernel<T>), (int)lds);
  if (cached == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, iel_fwd_kernel<T>, kIelThreads, lds) != hipSuccess || n < 1) n = 1;
    cached = n;
  }
  return cached;
}

// cost of a launch in arbitrary units: rounds of resident blocks x MFMA work per block (project_in on the halo region
// + project_out on the tile, output channels padded to 16)
template <class T>
double iel_cost(int B, int H, int W) {
  const long blocks = (long)B * ((W + T::TW - 1) / T::TW) * ((H + T::TH - 1) / T::TH);
  const double per_block = (double)T::NGP * 64 * 2 * T::C + (double)T::NGO * 64 * T::MT * 16;
  const long slots = 256L * iel_resident<T>();
  return (double)((blocks + slots - 1) / slots) * per_block;
}

template <class T>
int launch_iel_fwd(const IelArgs& a0, hipStream_t st) {
  IelArgs a = a0;
  a.tiles_x = (a.W + T::TW - 1) / T::TW;
  a.tiles_y = (a.H + T::TH - 1) / T::TH;
  constexpr size_t lds = sizeof(float) * T::LDS_FLOATS;
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  (void)iel_resident<T>();                // also raises the kernel's dynamic-LDS limit (once)
  const long blocks = (long)a.B * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(iel_fwd_kernel<T>, dim3((unsigned)blocks), dim3(kIelThreads), lds, st, a);
  return 0;
}

}  // namespace
}  // namespace cidnet

using namespace cidnet;

#define IEL_TRY(Cc, TH, TW)                                                     \
  {                                                                             \
    using T = IelT<Cc, TH, TW>;                                                 \
    const double cst = iel_cost<T>(B, H, W);                                    \
    if (best < 0 || cst < best_cost) { best = idx; best_cost = cst; }           \
    ++idx;                                                                      \
  }
#define IEL_RUN(Cc, TH, TW)                                                     \
  {                                                                             \
    if (best == idx) launch_iel_fwd<IelT<Cc, TH, TW>>(a, (hipStream_t)stream);  \
    ++idx;                                                                      \
  }
// tile shapes per channel count: the accumulators of project_out (UPW x 16 registers per GEMM lane) bound the tile
#define IEL_CONFIGS_S(X, Cc) X(Cc, 16, 32) X(Cc, 8, 60) X(Cc, 8, 32) X(Cc, 8, 16)
#define IEL_CONFIGS_M(X, Cc) X(Cc, 8, 32) X(Cc, 8, 16)
#define IEL_CONFIGS_L(X, Cc) X(Cc, 8, 16)      // K >= 72: 4 KS operand registers per lane leave room for the small tile only

extern "C" {

#ifdef IEL_TIMING
int cidnet_debug_iel_phases(unsigned long long* host, int nblocks) {
  (void)hipDeviceSynchronize();
  const size_t n = sizeof(unsigned long long) * 8 * (nblocks < 4096 ? nblocks : 4096);
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_iel_phase), n);
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_iel_phase)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 8 * 4096);
  return rc;
}
#endif

int cidnet_iel_fwd_supported(int C, int h) { return (C == 36 || C == 72 || C == 12 || C == 24 || C == 48) && h >= 1; }

int cidnet_iel_fwd(const float* xn, const float* res, const float* w_in, const float* w_dw, const float* w_dw1,
                   const float* w_dw2, const float* w_out, float* u, float* out, int B, int C, int h, int H, int W,
                   void* stream) {
  CIDNET_CHECK_ARG(xn && w_in && w_dw && w_dw1 && w_dw2 && w_out && out && B > 0 && C > 0 && h > 0 && H > 0 && W > 0);
  if (!cidnet_iel_fwd_supported(C, h)) return CIDNET_ERR_SHAPE;
  IelArgs a{xn, res, w_in, w_dw, w_dw1, w_dw2, w_out, u, out, B, h, H, W, 0, 0};
  int best = -1, idx = 0;
  double best_cost = 0.0;
  switch (C) {
    case 36: IEL_CONFIGS_S(IEL_TRY, 36) idx = 0; IEL_CONFIGS_S(IEL_RUN, 36) break;
    case 72: IEL_CONFIGS_L(IEL_TRY, 72) idx = 0; IEL_CONFIGS_L(IEL_RUN, 72) break;
    case 12: IEL_CONFIGS_S(IEL_TRY, 12) idx = 0; IEL_CONFIGS_S(IEL_RUN, 12) break;
    case 24: IEL_CONFIGS_M(IEL_TRY, 24) idx = 0; IEL_CONFIGS_M(IEL_RUN, 24) break;
    case 48: IEL_CONFIGS_L(IEL_TRY, 48) idx = 0; IEL_CONFIGS_L(IEL_RUN, 48) break;
    default: return CIDNET_ERR_SHAPE;
  }
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
