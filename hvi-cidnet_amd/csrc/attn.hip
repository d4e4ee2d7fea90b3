// K7: LCA channel attention (CAB, net/LCA.py:26-38).  Per (sample, head) the attention matrix is
// only (c/head x c/head) = 18x18; the cost is streaming q,k,v (C x HW each), so the work is split as
//   gram        : S = q k^T, |q_i|^2, |k_j|^2 over HW   -- fp32 MFMA, split-K over pixels into slabs
//   softmax_fwd : fixed-order slab sum, L2 normalisation folded into the logits, * temperature,
//                 softmax (one row per lane group), then M_b = W_proj * blockdiag(attn_b) so that
//                 "attn @ v" and project_out collapse into ONE per-sample 1x1 conv (pw.hip)
//   softmax_bwd : from dM_b: d W_proj, d temperature, and the per-sample (2C x 2C) matrix that maps
//                 [q;k] to [dq;dk] (softmax + normalisation backward folded), applied by pw.hip.
// q, k, v live in one (B, 3C, HW) tensor: channels [0,C) q, [C,2C) k, [2C,3C) v.
#include "common.h"
#include "cidnet_hip.h"

namespace cidnet {
namespace {

constexpr int kThreads = 256;
constexpr float kNormEps = 1e-12f;   // F.normalize eps

// typed row access (DT = CIDNET_F32 / CIDNET_BF16, a compile-time constant): offsets in elements from the tensor's base
template <int DT>
__device__ __forceinline__ f32x4 ld_px4(const void* base, long row, long p, long pend, bool ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok && p < pend) {
    if (p + 3 < pend) v = ld4t(base, row + p, DT);
    else
      for (int e = 0; e < 4; ++e)
        if (p + e < pend) v[e] = ld1t(base, row + p + e, DT);
  }
  return v;
}

// slab layout per wave: [ch*ch S][ch nq2][ch nk2]
template <int TI, int DT = 0>           // DT: storage type of qkv (bf16 in the bf16 mode; the products stay fp32)
__global__ __launch_bounds__(kThreads) void gram_kernel(const void* __restrict__ qkv, float* __restrict__ slabs, int C,
                                                        int heads, long HW, int pch) {
  const int ch = C / heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, j = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const long pbeg = (long)blockIdx.x * pch;
  const long pend = (pbeg + pch < HW) ? pbeg + pch : HW;
  const long qb = ((long)b * 3 * C + (long)head * ch) * HW;    // element offsets into qkv
  const long kb = qb + (long)C * HW;

  f32x4 acc[TI][TI];
  float nq[TI], nk[TI];
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    nq[a] = 0.f; nk[a] = 0.f;
#pragma unroll
    for (int c = 0; c < TI; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Straight-line loads: rows past ch are clamped to a valid row (they only reach accumulator entries that are
  // never stored), and the step that crosses the end of the chunk is pulled back to pend-8 with its first `dup`
  // pixels (owned by the previous k-slot) masked out of q.  The next step's loads are issued before this step's
  // MFMAs and consumed one iteration later, so their latency hides behind the burst.
  long qrow[TI], krow[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    const long row = min(t * 16 + r, ch - 1);
    qrow[t] = qb + row * HW;
    krow[t] = kb + row * HW;
  }
  const bool wide = pend - pbeg >= 8;              // always, except degenerate planes: then the checked loader
  f32x4 qn[TI][2], kn[TI][2];
  auto issue = [&](long p) {
    const long pl = min(p + 8 * j, pend - 8);
#pragma unroll
    for (int t = 0; t < TI; ++t) {
      qn[t][0] = ld4t(qkv, qrow[t] + pl, DT); qn[t][1] = ld4t(qkv, qrow[t] + pl + 4, DT);
      kn[t][0] = ld4t(qkv, krow[t] + pl, DT); kn[t][1] = ld4t(qkv, krow[t] + pl + 4, DT);
    }
  };
  long p = pbeg + wave * 32;
  if (wide && p < pend) issue(p);
  for (; p < pend; p += 128) {
    f32x4 qa[TI][2], ka[TI][2];
    if (wide) {
      const long pl = p + 8 * j;
      const int dup = (int)(pl - min(pl, pend - 8));
#pragma unroll
      for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool live = h * 4 + e >= dup;
            qa[t][h][e] = live ? qn[t][h][e] : 0.f;
            ka[t][h][e] = live ? kn[t][h][e] : 0.f;
          }
      if (p + 128 < pend) issue(p + 128);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const long pl = p + 8 * j;
#pragma unroll
      for (int t = 0; t < TI; ++t) {
        const bool ok = t * 16 + r < ch;
        qa[t][0] = ld_px4<DT>(qkv, qrow[t], pl, pend, ok); qa[t][1] = ld_px4<DT>(qkv, qrow[t], pl + 4, pend, ok);
        ka[t][0] = ld_px4<DT>(qkv, krow[t], pl, pend, ok); ka[t][1] = ld_px4<DT>(qkv, krow[t], pl + 4, pend, ok);
      }
    }
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { nq[t] += qa[t][h][e] * qa[t][h][e]; nk[t] += ka[t][h][e] * ka[t][h][e]; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int tj = 0; tj < TI; ++tj)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[ti][h][e], ka[tj][h][e], acc[ti][tj], 0, 0, 0);
  }
  float* slab = slabs + ((((long)b * heads + head) * gridDim.x + blockIdx.x) * 4 + wave) * (long)(ch * ch + 2 * ch);
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TI; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = ti * 16 + j * 4 + reg, jj = tj * 16 + r;
        if (i < ch && jj < ch) slab[i * ch + jj] = acc[ti][tj][reg];
      }
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    float a = nq[t], c = nk[t];
    a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
    c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
    const int row = t * 16 + r;
    if (j == 0 && row < ch) { slab[ch * ch + row] = a; slab[ch * ch + ch + row] = c; }
  }
}

// one block per (head, sample)
__global__ __launch_bounds__(kThreads) void softmax_fwd_kernel(const float* __restrict__ slabs, int n_red,
                                                               const float* __restrict__ temperature,
                                                               const float* __restrict__ Wp, float* __restrict__ attn,
                                                               float* __restrict__ shat, float* __restrict__ nqo,
                                                               float* __restrict__ nko, float* __restrict__ Mout, int C,
                                                               int heads, int normalize) {
  extern __shared__ float sm[];          // [ch*ch + 2ch] sums, attn [ch*ch], Wp head slice [C*ch]
  const int ch = C / heads;
  const int head = blockIdx.x, b = blockIdx.y;
  const int ne = ch * ch + 2 * ch;
  float* S = sm;
  float* A = sm + ne;
  float* Wl = A + ch * ch;               // Wl[co*ch + k] = Wp[co][head*ch + k]
  const float* base = slabs + (((long)b * heads + head) * n_red) * (long)ne;
  // this block is one of only heads*B: everything below is a chain of global-memory round trips unless the loads
  // are independent and in flight together, so the slab sum keeps four partial sums per element (fixed order)
  for (int i = threadIdx.x; i < ne; i += blockDim.x) {
    float t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t[q] = 0.f;
    int k = 0;
    for (; k + 16 <= n_red; k += 16) {                          // sixteen loads in flight per lane (four left this kernel at ~24 us)
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] += base[(long)(k + q) * ne + i];
    }
    for (; k < n_red; ++k) t[0] += base[(long)k * ne + i];
    S[i] = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
           (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
  }
  for (int i = threadIdx.x; i < C * ch; i += blockDim.x) {
    const int co = i / ch, k = i - co * ch;
    Wl[i] = Wp[(long)co * C + head * ch + k];
  }
  __syncthreads();
  const float T = temperature[head];
  for (int i = threadIdx.x; i < 2 * ch; i += blockDim.x) {
    const float nv = normalize ? fmaxf(sqrtf(S[ch * ch + i]), kNormEps) : 1.0f;   // TNSM attention: raw dot products
    S[ch * ch + i] = nv;
    if (i < ch) nqo[(long)b * C + head * ch + i] = nv; else nko[(long)b * C + head * ch + (i - ch)] = nv;
  }
  __syncthreads();
  const long ho = ((long)b * heads + head) * ch * ch;
  for (int i = threadIdx.x; i < ch * ch; i += blockDim.x) {
    const int row = i / ch, col = i - row * ch;
    const float v = S[i] / (S[ch * ch + row] * S[ch * ch + ch + col]);
    S[i] = v;
    shat[ho + i] = v;
  }
  __syncthreads();
  for (int row = threadIdx.x; row < ch; row += blockDim.x) {   // ch <= 32 rows: one lane per row
    float mx = -INFINITY;
    for (int c = 0; c < ch; ++c) mx = fmaxf(mx, S[row * ch + c] * T);
    float den = 0.f;
    for (int c = 0; c < ch; ++c) { const float e = expf(S[row * ch + c] * T - mx); A[row * ch + c] = e; den += e; }
    const float inv = 1.f / den;
    for (int c = 0; c < ch; ++c) { const float v = A[row * ch + c] * inv; A[row * ch + c] = v; attn[ho + row * ch + c] = v; }
  }
  __syncthreads();
  // M_b[co][head*ch + jj] = sum_i Wp[co][head*ch + i] * attn[i][jj]
  for (int i = threadIdx.x; i < C * ch; i += blockDim.x) {
    const int co = i / ch, jj = i - co * ch;
    const float* wrow = Wl + co * ch;
    float t = 0.f;
    for (int k = 0; k < ch; ++k) t += wrow[k] * A[k * ch + jj];
    Mout[((long)b * C + co) * C + head * ch + jj] = t;
  }
}

// one block per (head, sample).  Outputs: dWp_b (B,C,C) [this head's columns], dT_b (B,heads),
// Wqk (B,2C,2C) rows of this head.
__global__ __launch_bounds__(kThreads) void softmax_bwd_kernel(const float* __restrict__ dM, const float* __restrict__ Wp,
                                                               const float* __restrict__ attn, const float* __restrict__ shat,
                                                               const float* __restrict__ nq, const float* __restrict__ nk,
                                                               const float* __restrict__ temperature, float* __restrict__ dWp_b,
                                                               float* __restrict__ dT_b, float* __restrict__ Wqk, int C,
                                                               int heads, int normalize) {
  extern __shared__ float sm[];    // A[ch*ch], G[ch*ch] (dattn -> dShat), a[ch], e[ch], Sh[ch*ch], Wl[C*ch], Dl[C*ch]
  __shared__ float red[kThreads / 64];
  const int ch = C / heads;
  const int head = blockIdx.x, b = blockIdx.y;
  float* A = sm;
  float* G = sm + ch * ch;
  float* ai = G + ch * ch;      // a_i
  float* ej = ai + ch;          // e_j
  float* Sh = ej + ch;          // shat of this (sample, head)
  float* Wl = Sh + ch * ch;     // Wl[co*ch + i] = Wp[co][hc+i]
  float* Dl = Wl + C * ch;      // Dl[co*ch + j] = dM[co][hc+j]
  const long ho = ((long)b * heads + head) * ch * ch;
  const float* dMb = dM + (long)b * C * C;
  const float T = temperature[head];
  // only heads*B of these blocks exist: stage every operand with independent, coalesced loads first, so the
  // products below run out of LDS instead of being chains of global-memory round trips
  for (int i = threadIdx.x; i < ch * ch; i += blockDim.x) { A[i] = attn[ho + i]; Sh[i] = shat[ho + i]; }
  for (int i = threadIdx.x; i < C * ch; i += blockDim.x) {
    const int co = i / ch, k = i - co * ch;
    Wl[i] = Wp[(long)co * C + head * ch + k];
    Dl[i] = dMb[(long)co * C + head * ch + k];
  }
  __syncthreads();
  // dattn[i][jj] = sum_co Wp[co][hc+i] * dM[co][hc+jj]
  for (int i = threadIdx.x; i < ch * ch; i += blockDim.x) {
    const int row = i / ch, col = i - row * ch;
    float t = 0.f;
    for (int co = 0; co < C; ++co) t += Wl[co * ch + row] * Dl[co * ch + col];
    G[i] = t;
  }
  // dWp_b[co][hc+i] = sum_jj dM[co][hc+jj] * attn[i][jj]
  for (int i = threadIdx.x; i < C * ch; i += blockDim.x) {
    const int co = i / ch, row = i - co * ch;
    float t = 0.f;
    for (int c = 0; c < ch; ++c) t += Dl[co * ch + c] * A[row * ch + c];
    dWp_b[((long)b * C + co) * C + head * ch + row] = t;
  }
  __syncthreads();
  // softmax backward per row, then dShat = T * dlogit; accumulate dT
  float dt_acc = 0.f;
  for (int row = threadIdx.x; row < ch; row += blockDim.x) {
    float dot = 0.f;
    for (int c = 0; c < ch; ++c) dot += A[row * ch + c] * G[row * ch + c];
    for (int c = 0; c < ch; ++c) {
      const float dl = A[row * ch + c] * (G[row * ch + c] - dot);
      dt_acc += dl * Sh[row * ch + c];
      G[row * ch + c] = T * dl;
    }
  }
  const float dts = block_sum(dt_acc, red);
  if (threadIdx.x == 0) dT_b[(long)b * heads + head] = dts;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * ch; i += blockDim.x) {
    float t = 0.f;
    if (i < ch) { for (int c = 0; c < ch; ++c) t += G[i * ch + c] * Sh[i * ch + c]; ai[i] = t; }
    else { const int c = i - ch; for (int rr = 0; rr < ch; ++rr) t += G[rr * ch + c] * Sh[rr * ch + c]; ej[c] = t; }
  }
  __syncthreads();
  // rows of the (2C x 2C) map [q;k] -> [dq;dk] owned by this head
  const float* nqb = nq + (long)b * C + head * ch;
  const float* nkb = nk + (long)b * C + head * ch;
  float* Wb = Wqk + (long)b * 4 * C * C;
  const int C2 = 2 * C;
  for (int i = threadIdx.x; i < 2 * ch * C2; i += blockDim.x) {
    const int lr = i / C2, col = i - lr * C2;          // local row (0..ch-1: dq rows, ch..2ch-1: dk rows)
    float v = 0.f;
    if (lr < ch) {
      const int row = lr;
      if (col == head * ch + row) {
        v = (normalize && nqb[row] > kNormEps) ? -ai[row] / (nqb[row] * nqb[row]) : 0.f;
      } else if (col >= C + head * ch && col < C + head * ch + ch) {
        const int c = col - C - head * ch;
        v = G[row * ch + c] / (nqb[row] * nkb[c]);
      }
      Wb[(long)(head * ch + row) * C2 + col] = v;
    } else {
      const int c = lr - ch;
      if (col == C + head * ch + c) {
        v = (normalize && nkb[c] > kNormEps) ? -ej[c] / (nkb[c] * nkb[c]) : 0.f;
      } else if (col >= head * ch && col < head * ch + ch) {
        const int row = col - head * ch;
        v = G[row * ch + c] / (nqb[row] * nkb[c]);
      }
      Wb[(long)(C + head * ch + c) * C2 + col] = v;
    }
  }
}

// out[i] = sum_r in[r*n + i]
__global__ void sum_rows_kernel(const float* __restrict__ in, int n_red, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int r = 0; r < n_red; ++r) t += in[(long)r * n + i];
  out[i] = t;
}

// the same sum for MANY rows (hundreds of per-block partials, e.g. cidnet_modulate_bwd's): one thread per output walking all
// rows is a chain of n_red dependent-latency loads (944 rows x 36 outputs: 218 us).  Here a block owns 8 outputs and 32 lanes
// per output stride over the rows, two partial sums each; the 32 partials are added through LDS in fixed order.
__global__ __launch_bounds__(256) void sum_rows_wide_kernel(const float* __restrict__ in, int n_red, long n, float* __restrict__ out) {
  __shared__ float part[32][9];
  const int ex = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const long i = (long)blockIdx.x * 8 + ex;
  float t0 = 0.f, t1 = 0.f;
  if (i < n) {
    int r = sl;
    for (; r + 32 < n_red; r += 64) { t0 += in[(long)r * n + i]; t1 += in[(long)(r + 32) * n + i]; }
    if (r < n_red) t0 += in[(long)r * n + i];
  }
  part[sl][ex] = t0 + t1;
  __syncthreads();
  if (sl == 0 && i < n) {
    float t = part[0][ex];
#pragma unroll
    for (int q = 1; q < 32; ++q) t += part[q][ex];
    out[i] = t;
  }
}

inline int gram_pch(long HW) { return HW >= 16384 ? 2048 : 512; }

}  // namespace
}  // namespace cidnet

using namespace cidnet;

extern "C" {

long cidnet_attn_gram_ws_floats(int B, int C, int heads, long HW) {
  const int ch = C / heads;
  const long chunks = (HW + gram_pch(HW) - 1) / gram_pch(HW);
  return (long)B * heads * chunks * 4 * (ch * ch + 2 * ch);
}

/* fwd: gram + softmax + fold with project_out.  Outputs attn/shat (B,heads,ch,ch), nq/nk (B,C), M (B,C,C). */
int cidnet_attn_fwd(const float* qkv, const float* temperature, const float* Wp, float* attn, float* shat, float* nq,
                    float* nk, float* M, float* ws, long ws_floats, int B, int C, int heads, long HW, int normalize, void* stream) {
  return cidnet_attn_fwd_t(qkv, CIDNET_F32, temperature, Wp, attn, shat, nq, nk, M, ws, ws_floats, B, C, heads, HW, normalize, stream);
}

/* qkv stored as fp32 or bf16 (qkv_dt); everything else as cidnet_attn_fwd */
int cidnet_attn_fwd_t(const void* qkv, int qkv_dt, const float* temperature, const float* Wp, float* attn, float* shat, float* nq,
                      float* nk, float* M, float* ws, long ws_floats, int B, int C, int heads, long HW, int normalize, void* stream) {
  CIDNET_CHECK_ARG(qkv && temperature && Wp && attn && shat && nq && nk && M && ws && B > 0 && C > 0 && heads > 0 && HW > 0 && (qkv_dt | 1) == 1);
  if (C % heads != 0 || C / heads > 32) return CIDNET_ERR_SHAPE;
  if (ws_floats < cidnet_attn_gram_ws_floats(B, C, heads, HW)) return CIDNET_ERR_WS;
  const int ch = C / heads;
  const int pch = gram_pch(HW);
  const int chunks = (int)((HW + pch - 1) / pch);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)chunks, (unsigned)heads, (unsigned)B);
  if (qkv_dt) {
    if (ch <= 16) hipLaunchKernelGGL((gram_kernel<1, 1>), grid, dim3(kThreads), 0, s, qkv, ws, C, heads, HW, pch);
    else hipLaunchKernelGGL((gram_kernel<2, 1>), grid, dim3(kThreads), 0, s, qkv, ws, C, heads, HW, pch);
  } else {
    if (ch <= 16) hipLaunchKernelGGL((gram_kernel<1, 0>), grid, dim3(kThreads), 0, s, qkv, ws, C, heads, HW, pch);
    else hipLaunchKernelGGL((gram_kernel<2, 0>), grid, dim3(kThreads), 0, s, qkv, ws, C, heads, HW, pch);
  }
  CIDNET_LAUNCH_STATUS();
  const size_t lds = (size_t)(2 * ch * ch + 2 * ch + C * ch) * sizeof(float);
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)heads, (unsigned)B), dim3(kThreads), lds, s, ws, chunks * 4,
                     temperature, Wp, attn, shat, nq, nk, M, C, heads, normalize);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_attn_bwd(const float* dM, const float* Wp, const float* attn, const float* shat, const float* nq, const float* nk,
                    const float* temperature, float* dWp_b, float* dT_b, float* Wqk, int B, int C, int heads, int normalize,
                    void* stream) {
  CIDNET_CHECK_ARG(dM && Wp && attn && shat && nq && nk && temperature && dWp_b && dT_b && Wqk && B > 0 && C > 0 && heads > 0);
  if (C % heads != 0 || C / heads > 32) return CIDNET_ERR_SHAPE;
  const int ch = C / heads;
  const size_t lds = (size_t)(3 * ch * ch + 2 * ch + 2 * C * ch) * sizeof(float);
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)heads, (unsigned)B), dim3(kThreads), lds, (hipStream_t)stream, dM, Wp,
                     attn, shat, nq, nk, temperature, dWp_b, dT_b, Wqk, C, heads, normalize);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

int cidnet_sum_rows(const float* in, int n_red, long n, float* out, void* stream) {
  CIDNET_CHECK_ARG(in && out && n_red > 0 && n > 0);
  if (n_red > 64)
    hipLaunchKernelGGL(sum_rows_wide_kernel, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, in, n_red, n, out);
  else
    hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, n_red, n, out);
  CIDNET_LAUNCH_STATUS();
  return CIDNET_OK;
}

}  // extern "C"
