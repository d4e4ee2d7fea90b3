/* cidnet_hip.h -- C ABI of libcidnet_hip.so: the MI355X (gfx950) kernels of the CIDNet
 * forward/backward hot path.
 *
 * The reference (KitaharaH/HVI-CIDNet) has no FFI: its boundary is the Python nn.Module API of
 * net/CIDNet.py.  Each entry point below replaces the group of ATen eager ops that one reference
 * call site executes (cited per function); hvi-cidnet_amd/ops.py wraps them as
 * torch.autograd.Function and hvi-cidnet_amd/{hvi_transform,transformer_utils,lca,cidnet}.py
 * re-expose the reference's module names on top.
 *
 * Conventions
 *   - all tensors are fp32, NCHW-contiguous device memory unless a stride argument says otherwise;
 *     "HW" planes are H*W floats; pointers are plain device pointers owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream; NULL = default);
 *   - functions are stateless and re-entrant, never allocate, never synchronise and never copy to
 *     the host, so a caller may capture them into a hipGraph;  scratch memory comes in through
 *     `ws` / `ws_floats` arguments whose required size the matching *_ws_floats() reports;
 *   - return 0 on success, a negative CIDNET_ERR_* for rejected arguments, or a positive
 *     hipError_t if the launch failed.  Kernels never abort the process.
 */
#ifndef CIDNET_HIP_H
#define CIDNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIDNET_ABI_VERSION 6

int cidnet_abi_version(void);

/* element-type codes of the typed (`_t`) entry points: a tensor passed as `void*` with an `int <name>_dt` is stored as fp32
 * or bfloat16; strides and offsets stay in ELEMENTS */
#define CIDNET_F32 0
#define CIDNET_BF16 1

/* ---- K1: RGB -> HVI  (RGB_HVI.HVIT, net/HVI_transform.py:16-47; CIDNet.HVIT, net/CIDNet.py:124) --
 * rgb, hvi: (B,3,H,W).  density_k: device pointer to the scalar parameter (read on the device, so
 * the reference's k.item() host sync at :38 disappears).  branch_code (optional, B*H*W bytes):
 * bits0-1 hue branch (0 gray,1 R,2 G,3 B), bits2-3 arg-max channel, bits4-5 arg-min channel,
 * bit6 value==0 -- the mask decisions of :23-27,31, exported for the bit-exactness test. */
int cidnet_hvit_fwd(const float* rgb, const float* density_k, float* hvi, uint8_t* branch_code,
                    int B, int H, int W, void* stream);

/* Backward of HVIT: g_rgb (B,3,H,W, optional) and g_k (1 float, optional; fixed-order reduction,
 * bitwise reproducible).  ws: >= cidnet_hvit_bwd_ws_floats() floats when g_k != NULL. */
long cidnet_hvit_bwd_ws_floats(void);
int cidnet_hvit_bwd(const float* rgb, const float* density_k, const float* g_hvi, float* g_rgb,
                    float* g_k, float* ws, long ws_floats, int B, int H, int W, void* stream);

/* ---- K2: HVI -> RGB  (RGB_HVI.PHVIT, net/HVI_transform.py:49-122) ---------------------------
 * Input is hvi (B,3,H,W); when hv (B,2,H,W) and iv (B,1,H,W) are non-NULL the residual form of
 * net/CIDNet.py:119, cat([hv, iv], 1) + hvi, is evaluated in registers.  k = *k_dev if k_dev !=
 * NULL else k_host (`this_k`, a python float in the reference: no gradient reaches density_k).
 * gated/alpha_s, gated2/alpha: the inference-time scales of :69-70,120-121.  sextant (optional,
 * B*H*W bytes): floor(6h) per pixel, 6 = the black-pixel case. */
int cidnet_phvit_fwd(const float* hv, const float* iv, const float* hvi, const float* k_dev,
                     float k_host, int gated, float alpha_s, int gated2, float alpha, float* rgb,
                     uint8_t* sextant, int B, int H, int W, void* stream);

/* Backward of PHVIT wrt its (summed) input: g_hvi (B,3,H,W) and/or the same gradient split as
 * g_hv (B,2,H,W) + g_iv (B,1,H,W) for the two decoder heads. */
int cidnet_phvit_bwd(const float* hv, const float* iv, const float* hvi, const float* k_dev,
                     float k_host, int gated, float alpha_s, int gated2, float alpha,
                     const float* g_rgb, float* g_hvi, float* g_hv, float* g_iv, int B, int H, int W,
                     void* stream);

/* ---- K3: channels-first LayerNorm  (LayerNorm.forward, net/transformer_utils.py:24-29) ---------
 * x,y: (B,C,HW).  mean/rstd (B,HW each, optional pair) are saved for the backward. */
int cidnet_ln_cf_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean,
                     float* rstd, int B, int C, long HW, float eps, void* stream);
long cidnet_ln_cf_bwd_ws_floats(int C);
/* gx optional (NULL: parameter gradients only); gw, gb: (C) each, overwritten. */
int cidnet_ln_cf_bwd(const float* x, const float* weight, const float* gy, const float* mean,
                     const float* rstd, float* gx, float* gw, float* gb, float* ws, long ws_floats,
                     int B, int C, long HW, void* stream);
/* Typed forms for the bf16 mode: the LayerNorm OUTPUT (it only feeds 1x1 convs, net/LCA.py:22-23,60) and the incoming
 * gradient (an output of those convs' data-gradient launches) may be stored as bf16; x, the statistics, gx and the
 * parameter gradients stay fp32.  bf16 only for cidnet_ln_cf_typed_supported shapes (C = 36 / 72 / 144), else
 * CIDNET_ERR_SHAPE. */
int cidnet_ln_cf_typed_supported(int B, int C, long HW);
int cidnet_ln_cf_fwd_t(const float* x, const float* weight, const float* bias, void* y, int y_dt, float* mean, float* rstd,
                       int B, int C, long HW, float eps, void* stream);
int cidnet_ln_cf_bwd_res_t(const float* x, const float* weight, const void* gy, int gy_dt, const float* mean,
                           const float* rstd, const float* addend, float* gx, float* gw, float* gb, int accumulate,
                           float* ws, long ws_floats, int B, int C, long HW, void* stream);
/* same, with gx += addend (the gradient that reaches x through the residual branch of the pre-norm block,
 * net/LCA.py:79,80,91,92): saves the separate accumulation pass.  addend may be NULL.  accumulate != 0: gw / gb are
 * added to instead of overwritten -- one LayerNorm module is applied two or three times per LCA (LCA.py:79-80,91-92)
 * and its uses then sum their parameter gradients in place, in backward order. */
int cidnet_ln_cf_bwd_res(const float* x, const float* weight, const float* gy, const float* mean, const float* rstd,
                         const float* addend, float* gx, float* gw, float* gb, int accumulate, float* ws,
                         long ws_floats, int B, int C, long HW, void* stream);

/* Two LayerNorm MODULES applied to one tensor (the x-norm of an LCA block and the y-norm of its partner block read the same
 * input, net/LCA.py:79-80,91-92 with net/CIDNet.py:83-84): one pass reads x and writes both outputs; one backward pass reads
 * x once and returns gx = LN'(gy*w + gy2*w2) (+ addend) and both modules' parameter gradients (each with its own
 * accumulate flag).  _supported: the shapes of the register-resident kernels (C = 36 with HW % 4 == 0, C = 72 with HW % 2 == 0,
 * C = 144 on planes of at most 2^18 pixels per batch); otherwise CIDNET_ERR_SHAPE -- call the single-module entry points. */
int cidnet_ln_cf_dual_supported(int B, int C, long HW);
long cidnet_ln_cf_bwd2_ws_floats(int C);
int cidnet_ln_cf_fwd2(const float* x, const float* weight, const float* bias, float* y, const float* weight2,
                      const float* bias2, float* y2, float* mean, float* rstd, int B, int C, long HW, float eps,
                      void* stream);
int cidnet_ln_cf_bwd2(const float* x, const float* weight, const float* gy, const float* weight2, const float* gy2,
                      const float* mean, const float* rstd, const float* addend, float* gx, float* gw, float* gb,
                      int accumulate, float* gw2, float* gb2, int accumulate2, float* ws, long ws_floats, int B, int C,
                      long HW, void* stream);

/* ---- K4: pointwise (1x1) convolution on the fp32 MFMA -----------------------------------------
 * (nn.Conv2d(k=1): net/LCA.py:13,15,17,51,57; net/transformer_utils.py:60)
 * Per sample b:  Y[b] (M x HW) = A_b (M x K) * X[b] (K x HW)  [+ R[b]],
 *   A_b[m][k] = Wt[b*w_bs + m*w_ms + k*w_ks]   (w_bs = 0: shared weights; forward: w_ms=K,w_ks=1;
 *   data gradient: w_ms=1, w_ks=<Cin>), X[b] at X + b*x_bs with channel stride HW (same for Y, R),
 *   so channel slices of wider tensors can be read / written in place.  R may alias Y. */
#ifdef CIDNET_DEBUG
/* timing-study switches for cidnet_pw_conv (bit0: no stores, bit1: no K loop; bit7: force the weight-gradient
 * kernel to (flags >> 8) * 128 pixels per block); 0 = production.  Only in -DCIDNET_DEBUG builds. */
void cidnet_debug_pw_flags(int flags);
#endif
/* Element-type codes of the `_t` entry points (row J1, bf16 storage mode): a tensor argument declared `void*` with an
 * `int <name>_dt` is fp32 (CIDNET_F32) or bfloat16 (CIDNET_BF16); strides stay in ELEMENTS.  Arithmetic is fp32 in
 * every kernel: bf16 is a storage format of activations / saved tensors (round to nearest even on store). */
int cidnet_pw_conv_t(const void* X, int x_dt, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks, void* Y,
                     int y_dt, long y_bs, const float* R, long r_bs, int B, int M, int K, long HW, void* stream);
int cidnet_pw_conv(const float* X, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks,
                   float* Y, long y_bs, const float* R, long r_bs, int B, int M, int K, long HW,
                   void* stream);
/* The same product on the BF16 matrix cores with exact three-way split operands (csrc/pwx.hip; fp32 tensors only): for
 * launches bound by the fp32 MFMA rate or by that kernel's instruction issue.  _supported: M > 16, HW >= 64 (any HW
 * from there: the last 64-pixel group of a plane is pulled back to the plane's end), K * HW and M * HW below 2^31.
 * ws: cidnet_pw_conv_bf16x3_ws_floats(B, M, K, w_bs != 0) floats, 16-byte aligned (the weights, split once per call into
 * MFMA fragment order). */
int cidnet_pw_conv_bf16x3_supported(int M, int K, long HW);
long cidnet_pw_conv_bf16x3_ws_floats(int B, int M, int K, int per_sample);
int cidnet_pw_conv_bf16x3(const float* X, long x_bs, const float* Wt, long w_bs, long w_ms, long w_ks, float* Y, long y_bs,
                          const float* R, long r_bs, float* ws, long ws_floats, int B, int M, int K, long HW, void* stream);
/* The same call in two halves, for callers that keep the prepared weight operand across calls (weights change once per
 * optimizer step, but are used by a forward and -- transposed -- a backward launch, and by every inference call):
 * _prep splits Wt into ws (nb sets: B for per-sample weights, w_bs != 0; else 1; ws as above with per_sample = 1 and
 * B = nb), _pre runs the product from it.  cidnet_pw_conv_bf16x3 == _prep + _pre.  _prep_batch prepares n shared (not
 * per-sample) weight tensors in ONE launch: `table` is a device array of n rows of 8 x int64 {Wt pointer, ws pointer,
 * M, K, w_ms, w_ks, first block of the row, 0}; a row takes cidnet_pw_conv_bf16x3_prep_blocks(M, K) blocks, rows in
 * ascending block order, total_blocks = their sum. */
int cidnet_pw_conv_bf16x3_prep(const float* Wt, long w_bs, long w_ms, long w_ks, float* ws, long ws_floats, int nb, int M,
                               int K, void* stream);
int cidnet_pw_conv_bf16x3_pre(const float* X, long x_bs, const float* Wprep, int per_sample, float* Y, long y_bs,
                              const float* R, long r_bs, int B, int M, int K, long HW, void* stream);
/* w_levels / x_levels: how many bf16 levels of the weights / activations enter the products (all pairs i + j <= 2):
 * (3, 3) = _pre, the fp32-exact six products; (1, 1) both operands rounded to nearest bf16 and ONE product -- the
 * arithmetic of a bf16 autocast convolution with fp32 accumulation and fp32 output (BASELINE.json configs[2]'s "bf16"
 * mode; the activation split disappears and the matrix-core work drops to a sixth).  Other pairs: CIDNET_ERR_ARG (the
 * 3x3 conv below also takes (3, 1): exact weights, rounded activations, three products).  The prepared operand is the
 * same for every pair. */
int cidnet_pw_conv_bf16x3_pre_lv(const float* X, long x_bs, const float* Wprep, int per_sample, float* Y, long y_bs,
                                 const float* R, long r_bs, int B, int M, int K, long HW, int w_levels, int x_levels,
                                 void* stream);
/* ... with typed activations / output (x_dt, y_dt: CIDNET_F32 / CIDNET_BF16, defined below; strides in elements): a tensor
 * STORED as bf16 is read without conversion (its one level) / written rounded to nearest; bf16 types need (1, 1) levels.
 * R stays fp32 (the residual stream). */
int cidnet_pw_conv_bf16x3_pre_t(const void* X, int x_dt, long x_bs, const float* Wprep, int per_sample, void* Y, int y_dt,
                                long y_bs, const float* R, long r_bs, int B, int M, int K, long HW, int w_levels,
                                int x_levels, void* stream);
long cidnet_pw_conv_bf16x3_prep_blocks(int M, int K);
int cidnet_pw_conv_bf16x3_prep_batch(const long long* table, int n, long total_blocks, void* stream);
/* Backward of a 1x1 convolution Y = W X (W: (M, N) contiguous) in one kernel: gX (B, N, HW) = W^T gY and dW (M, N) = sum over
 * samples and pixels of gY X^T (overwritten; partial sums combined in fixed order).  gY is read from HBM once instead of once
 * by the data-gradient launch (cidnet_pw_conv* with transposed strides) and once by cidnet_pw_wgrad*.  fp32 tensors, split
 * bf16 products as cidnet_pw_conv_bf16x3.  _supported: the IEL / CAB layer shapes of the 36-channel level (M x N tiles
 * 12x3, 3x6, 3x3, 5x3) with HW a multiple of 4; other shapes CIDNET_ERR_SHAPE -- use the two separate entry points. */
int cidnet_pw_bwd_fused_supported(int M, int N, long HW);
long cidnet_pw_bwd_fused_ws_floats(int B, int M, int N, long HW);
int cidnet_pw_bwd_fused(const float* gY, long gy_bs, const float* X, long x_bs, const float* Wt, float* gX, long gx_bs,
                        float* dW, float* ws, long ws_floats, int B, int M, int N, long HW, void* stream);
/* NormUpsample tail (net/transformer_utils.py:64-66): pre = Wt*X + bilinear_x2(Z), Y = PReLU(pre).
 * Z: (B,M,zh,zw) low-resolution, X: skip tensor (B,K,2zh,2zw), Y/Ypre: (B,M,2zh,2zw); Ypre may be NULL (inference). */
int cidnet_pw_conv_up_prelu(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks,
                            const float* Z, const float* slope, float* Y, float* Ypre, int B, int M,
                            int K, int zh, int zw, void* stream);
/* Weight gradient dW[m][n] = sum_{b,p} dY[b][m][p] X[b][n][p]; per_sample: dW is (B,M,N) without
 * the batch sum.  dw_ld = row stride of dW (>= N).  Fixed-order reduction (reproducible).
 * flags: CIDNET_WGRAD_ACCUMULATE adds to dW instead of overwriting it; with fp32 operands the products run on the
 * BF16 matrix cores as six exact bf16 cross products each (results within fp32 rounding, see cidnet_conv3x3_bf16x3)
 * unless CIDNET_WGRAD_FP32_MFMA is set. */
#define CIDNET_WGRAD_ACCUMULATE 1
#define CIDNET_WGRAD_FP32_MFMA 2
/* both operands rounded to nearest bf16, one product per term (the bf16 mode; any storage types) */
#define CIDNET_WGRAD_BF16_1LEVEL 4
long cidnet_pw_wgrad_ws_floats(int B, int M, int N, long HW);
int cidnet_pw_wgrad_t(const void* dY, int dy_dt, long dy_bs, const void* X, int x_dt, long x_bs, float* dW, long dw_ld,
                      int per_sample, int flags, float* ws, long ws_floats, int B, int M, int N, long HW,
                      void* stream);
int cidnet_pw_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, long dw_ld,
                    int per_sample, int flags, float* ws, long ws_floats, int B, int M, int N,
                    long HW, void* stream);

/* ---- K5 / K8: depthwise 3x3 (zero pad) and the IEL gate  (net/LCA.py:14,16,53-55,62-65) --------
 * out = dw3x3(in) [+ addend]; channel c uses w1[c] if c < csplit else w2[c-csplit] (weights (.,1,3,3));
 * flip != 0 applies the 180-degree rotated taps (= data gradient of the forward). */
#ifdef CIDNET_DEBUG
/* timing-study switch: force the strip height of the depthwise / gate kernels (0 = automatic) */
void cidnet_debug_dw_rows(int rows);
#endif
int cidnet_dw3x3(const float* in, const float* w1, const float* w2, int csplit, const float* addend,
                 float* out, int flip, int B, int C, int H, int W, void* stream);
/* in / addend / out stored as fp32 or bf16 (one type, dt): the CAB's q / kv depthwise convs in the bf16 mode */
int cidnet_dw3x3_t(const void* in, const float* w1, const float* w2, int csplit, const void* addend, void* out, int dt,
                   int flip, int B, int C, int H, int W, void* stream);
/* ---- K8 fused: tile-resident IEL forward (net/LCA.py:60-67 [+ the residual of I_LCA, LCA.py:92]) -------------
 * out = [res +] W_out * ((tanh(dw1 u1) + u1) * (tanh(dw2 u2) + u2)),  [u1; u2] = dw(W_in * xn), in ONE kernel: the
 * hidden tensors live in LDS only (csrc/iel.hip).  xn, res, out: (B,C,H,W); w_in (2h,C), w_dw (2h,1,3,3),
 * w_dw1 / w_dw2 (h,1,3,3), w_out (C,h).  u (B,2h,H,W) is written when non-NULL (the backward reads it); res may be
 * NULL.  cidnet_iel_fwd_supported: channel counts the kernel is instantiated for (others: CIDNET_ERR_SHAPE). */
int cidnet_iel_fwd_supported(int C, int h);
int cidnet_iel_fwd(const float* xn, const float* res, const float* w_in, const float* w_dw, const float* w_dw1,
                   const float* w_dw2, const float* w_out, float* u, float* out, int B, int C, int h, int H, int W,
                   void* stream);
long cidnet_dw3x3_wgrad_ws_floats(int B, int C, int H, int W);
int cidnet_dw3x3_wgrad(const float* in, const float* gout, float* gw1, float* gw2, int csplit,
                       float* ws, long ws_floats, int B, int C, int H, int W, void* stream);
/* fused backward: gin = dw3x3(gout, flipped) [+ addend] and gw in ONE pass over (in, gout) */
int cidnet_dw3x3_bwd(const float* in, const float* gout, const float* w1, const float* w2, int csplit,
                     const float* addend, float* gin, float* gw1, float* gw2, float* ws,
                     long ws_floats, int B, int C, int H, int W, void* stream);
/* g = (tanh(dw1(u1)) + u1) * (tanh(dw2(u2)) + u2);  u: (B,2h,H,W) = [u1;u2], g: (B,h,H,W). */
/* u = dw3x3(pin, wdw) (2h channels) and g = gate(u) in one pass (net/LCA.py:61-65): u is written once and never re-read.
 * u may be NULL: inference, where only the backward would read u -- the kernel then stores g alone. */
int cidnet_iel_dw_gate_fwd(const float* pin, const float* wdw, const float* w1, const float* w2, float* u, float* g,
                           int B, int h, int H, int W, void* stream);
/* bf16 storage mode (row J1): the same three kernels with the IEL's hidden tensors (pin, u, g; dg, du; in, gout, gin)
 * stored in the element type `dt` (CIDNET_F32 / CIDNET_BF16); weights, weight gradients and arithmetic stay fp32. */
int cidnet_iel_dw_gate_fwd_t(const void* pin, const float* wdw, const float* w1, const float* w2, void* u, void* g, int dt,
                             int B, int h, int H, int W, void* stream);
int cidnet_iel_gate_dw_bwd_t(const void* u, const float* w1, const float* w2, const void* dg, void* du, int dt, float* gw1,
                             float* gw2, float* ws, long ws_floats, int B, int h, int H, int W, void* stream);
int cidnet_dw3x3_bwd_t(const void* in, const void* gout, const float* w1, const float* w2, int csplit, const void* addend,
                       void* gin, int dt, float* gw1, float* gw2, float* ws, long ws_floats, int B, int C, int H, int W,
                       void* stream);
int cidnet_iel_gate_fwd(const float* u, const float* w1, const float* w2, float* g, int B, int h,
                        int H, int W, void* stream);
/* da = d(dw1/2 output), ds = d(s1/s2) both (B,2h,H,W); du = ds + dw3x3(da, flipped) by the caller. */
int cidnet_iel_gate_bwd(const float* u, const float* w1, const float* w2, const float* dg, float* da,
                        float* ds, int B, int h, int H, int W, void* stream);
/* fused: du = ds + dw^T(da) with da, ds recomputed on the fly, plus the dwconv1/dwconv2 weight gradients;
 * HBM traffic: read dg (h) + u (2h), write du (2h) -- the unfused pair moved 15h channel passes */
long cidnet_iel_gate_dw_bwd_ws_floats(int B, int h, int H, int W);
int cidnet_iel_gate_dw_bwd(const float* u, const float* w1, const float* w2, const float* dg, float* du,
                           float* gw1, float* gw2, float* ws, long ws_floats, int B, int h, int H, int W,
                           void* stream);

/* ---- K9/K10/K11: dense 3x3 convolution, pad 1, fp32 MFMA implicit GEMM -------------------------
 * (net/transformer_utils.py:39,58 zero pad; net/CIDNet.py:21-24,32-35,39-42,50-53 replicate pad)
 * Y[b][m] = sum_{k,tap} Wt[m*w_ms + k*w_ks + tap'] * Xpad[b][k] ; forward: w_ms=9K, w_ks=9, flip=0;
 * data gradient of the zero-pad conv: X=dY, M=Cin, K=Cout, w_ms=9, w_ks=9*Cin, flip=1. */
#ifdef CIDNET_DEBUG
/* timing-study switches for cidnet_conv3x3 (1 no stores, 2 no window prefetch loads, 4 constant weights,
 * 8 MFMA path even for thin (<= 4 channel) layers, 16 padded 16-row tiles instead of 4x4x1 row groups,
 * 128 weight gradient: input-channel remainder as a padded 16-column tile instead of the 4-column-group launch) */
void cidnet_debug_c3_flags(int flags);
#endif
int cidnet_conv3x3(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip,
                   int replicate, float* Y, long y_bs, int B, int M, int K, int H, int W, void* stream);
/* The same zero-pad convolution (net/transformer_utils.py:39,58 and its data gradient) with its fp32 operands split
 * into three bf16 values each and the six significant cross products run on the BF16 matrix cores (csrc/conv3x.hip):
 * results within fp32 rounding of cidnet_conv3x3 (products exact, fp32 accumulation, dropped terms <= 2^-25 relative),
 * on a pipe the VALU does not contend for.  R optional addend (batch stride r_bs).  ws (cidnet_conv3x3_bf16x3_ws_floats
 * floats, 16-byte aligned) receives the split weights in MFMA fragment order.  _supported: K a multiple of 36. */
int cidnet_conv3x3_bf16x3_supported(int M, int K);
long cidnet_conv3x3_bf16x3_ws_floats(int M, int K);
int cidnet_conv3x3_bf16x3(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip, const float* R,
                          long r_bs, float* Y, long y_bs, float* ws, long ws_floats, int B, int M, int K, int H,
                          int W, void* stream);
/* In two halves (see cidnet_pw_conv_bf16x3_prep): _prep writes the prepared weights (flip folded in) to ws, _pre convolves
 * from them; _prep_batch: rows {Wt pointer, ws pointer, M, K, w_ms, w_ks, first block, flip},
 * cidnet_conv3x3_bf16x3_prep_blocks(M, K) blocks per row. */
int cidnet_conv3x3_bf16x3_prep(const float* Wt, long w_ms, long w_ks, int flip, float* ws, long ws_floats, int M, int K,
                               void* stream);
int cidnet_conv3x3_bf16x3_pre(const float* X, long x_bs, const float* Wprep, const float* R, long r_bs, float* Y, long y_bs,
                              int B, int M, int K, int H, int W, void* stream);
int cidnet_conv3x3_bf16x3_pre_lv(const float* X, long x_bs, const float* Wprep, const float* R, long r_bs, float* Y,
                                 long y_bs, int B, int M, int K, int H, int W, int w_levels, int x_levels, void* stream);
long cidnet_conv3x3_bf16x3_prep_blocks(int M, int K);
int cidnet_conv3x3_bf16x3_prep_batch(const long long* table, int n, long total_blocks, void* stream);
/* Y = conv3x3(X) + R (R of Y's shape, batch stride r_bs; NULL = plain conv).  Used by the data gradient of
 * NormDownsample when its input also feeds a skip connection (net/CIDNet.py:80-81,85-86): the skip's gradient is
 * added in the epilogue instead of by a separate pass over the tensor.  Layers with <= 4 channels on a side
 * (streaming kernels) do not take an addend: CIDNET_ERR_ARG. */
int cidnet_conv3x3_add(const float* X, long x_bs, const float* Wt, long w_ms, long w_ks, int flip,
                       int replicate, const float* R, long r_bs, float* Y, long y_bs, int B, int M, int K,
                       int H, int W, void* stream);
long cidnet_conv3x3_wgrad_ws_floats(int B, int M, int N, int H, int W);
int cidnet_conv3x3_wgrad(const float* dY, long dy_bs, const float* X, long x_bs, int replicate,
                         float* dW, float* ws, long ws_floats, int B, int M, int N, int H, int W,
                         void* stream);
/* The zero-pad weight gradient with split operands on the BF16 matrix cores (csrc/conv3xw.hip; arithmetic as
 * cidnet_conv3x3_bf16x3, partial sums combined in fixed order: run-to-run identical).  dW (M, N, 3, 3) contiguous is
 * overwritten.  _supported: N (input channels) a multiple of 36, W >= 4. */
int cidnet_conv3x3_wgrad_bf16x3_supported(int M, int N, int H, int W);
long cidnet_conv3x3_wgrad_bf16x3_ws_floats(int B, int M, int N, int H, int W);
int cidnet_conv3x3_wgrad_bf16x3(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, float* ws,
                                long ws_floats, int B, int M, int N, int H, int W, void* stream);
/* levels: bf16 levels of BOTH operands that enter the products: 3 = the call above (six products, fp32-exact); 1 = operands
 * rounded to nearest bf16, one product (the bf16 mode, see cidnet_pw_conv_bf16x3_pre_lv) */
int cidnet_conv3x3_wgrad_bf16x3_lv(const float* dY, long dy_bs, const float* X, long x_bs, float* dW, float* ws,
                                   long ws_floats, int B, int M, int N, int H, int W, int levels, void* stream);
#ifdef CIDNET_DEBUG
/* timing-study switches for cidnet_conv3x3_wgrad_bf16x3 (1 stage only a block's first tile, 2 no fragment reads / MFMAs) */
void cidnet_debug_c3xw_flags(int flags);
#endif
/* Adds to gX (computed by the zero-pad data gradient) the taps that read replicated border pixels. */
int cidnet_conv3x3_replicate_dgrad_fix(const float* gY, const float* Wt, float* gX, int B, int Co,
                                       int Ci, int H, int W, void* stream);

/* ---- resampling / PReLU pieces of NormDownsample, NormUpsample ---------------------------------
 * (nn.UpsamplingBilinear2d == bilinear, align_corners=True; nn.PReLU with one slope) */
/* pre (the pre-activation, read only by cidnet_prelu_bwd) may be NULL: inference. */
int cidnet_down_prelu_fwd(const float* t, const float* slope, float* pre, float* out, int B, int C,
                          int H, int W, void* stream);
long cidnet_prelu_bwd_ws_floats(void);
int cidnet_prelu_bwd(const float* dout, const float* pre, const float* slope, float* dpre,
                     float* dslope, float* ws, long ws_floats, long n, void* stream);
/* adjoint of bilinear (Hi,Wi)->(Ho,Wo): din (B,C,Hi,Wi) from dout (B,C,Ho,Wo), gather form. */
long cidnet_bilinear_bwd_ws_floats(int Hi, int Wi);
int cidnet_bilinear_bwd(const float* dout, float* din, float* ws, long ws_floats, int B, int C, int Hi,
                        int Wi, int Ho, int Wo, void* stream);
/* In two halves: the per-axis tap tables depend on the four sizes only (cidnet_bilinear_bwd_ws_floats(Hi, Wi) floats), so a
 * caller may compute them once per shape (_tabs) and run the adjoint from them (_pre). */
int cidnet_bilinear_bwd_tabs(float* tabs, long tabs_floats, int Hi, int Wi, int Ho, int Wo, void* stream);
int cidnet_bilinear_bwd_pre(const float* dout, float* din, const float* tabs, int B, int C, int Hi, int Wi, int Ho, int Wo,
                            void* stream);
int cidnet_add(const float* a, const float* b, float* y, long n, void* stream);

/* ---- K7: channel attention of CAB  (net/LCA.py:26-38) -------------------------------------------
 * qkv: (B,3C,HW) = [q;k;v] after the depthwise convs.  Produces attn = softmax(normalize(q)
 * normalize(k)^T * temperature) per (b,head), the normalised logits shat, the row norms nq,nk (B,C)
 * and M[b] = Wp * blockdiag(attn[b]) (B,C,C) so that project_out(attn @ v) = M[b] * v[b].
 * normalize = 0 gives the TNSM attention (net/TNSM.py:98-104): raw q k^T logits, no L2 normalisation. */
long cidnet_attn_gram_ws_floats(int B, int C, int heads, long HW);
int cidnet_attn_fwd(const float* qkv, const float* temperature, const float* Wp, float* attn,
                    float* shat, float* nq, float* nk, float* M, float* ws, long ws_floats, int B,
                    int C, int heads, long HW, int normalize, void* stream);
/* qkv stored as fp32 or bf16 (qkv_dt: CIDNET_F32 / CIDNET_BF16); the Gram products stay on the fp32 MFMA */
int cidnet_attn_fwd_t(const void* qkv, int qkv_dt, const float* temperature, const float* Wp, float* attn, float* shat,
                      float* nq, float* nk, float* M, float* ws, long ws_floats, int B, int C, int heads, long HW,
                      int normalize, void* stream);
/* From dM (B,C,C): per-sample dWp_b (B,C,C), dT_b (B,heads), and Wqk (B,2C,2C) with
 * [dq;dk][b] = Wqk[b] * [q;k][b]  (softmax + L2-normalisation backward folded). */
int cidnet_attn_bwd(const float* dM, const float* Wp, const float* attn, const float* shat,
                    const float* nq, const float* nk, const float* temperature, float* dWp_b,
                    float* dT_b, float* Wqk, int B, int C, int heads, int normalize, void* stream);
/* out[i] = sum_r in[r*n + i] (fixed order) */
int cidnet_sum_rows(const float* in, int n_red, long n, float* out, void* stream);

/* ---- training-step pieces (train.py:56-73) ------------------------------------------------------
 * loss = mean(|out - gt|) (L1Loss, loss/losses.py:10-20) and grad = sign(out - gt)/n in one pass. */
long cidnet_l1_loss_ws_floats(void);
int cidnet_l1_loss(const float* out, const float* gt, float* grad, float* loss, float* ws,
                   long ws_floats, long n, void* stream);
/* y = x * s[0] * mult (s: device scalar, may be null = 1): the upstream factor d(total)/d(loss) of a loss's
 * backward, and the sign flip that gives the gradient wrt the target (train.py:62: gt_hvi = model.HVIT(gt_rgb)
 * is NOT detached, so density_k receives gradient through the target of every HVI-space term) */
int cidnet_scale(const float* x, const float* s, float mult, float* y, long n, void* stream);
/* Gradient of the HVI image at its three-way fan-out (net/CIDNet.py:73-77: `i = hvi[:,2,:,:].unsqueeze(1)` feeds the I stem,
 * hvi the HV stem; :119 `+ hvi` the output residual): out (B,3,HW) = ga + gc, plane 2 also + gi (B,1,HW).  Null inputs
 * count as zero.  Replaces autograd's two accumulation passes and the slice backward (fill + copy). */
int cidnet_hvi_grad_sum(const float* ga, const float* gc, const float* gi, float* out, int B, long HW, void* stream);
/* SSIM loss ("next" row f1): SSIM.forward, loss/losses.py:166-190 with map_ssim, loss/loss_utils.py:125-145:
 * 11x11 Gaussian window (sigma 1.5), zero padding 5, depthwise; loss = (1 - mean(ssim_map)) * weight (1 float on the
 * device).  dA/dB/dC (B,C,H,W each) are the per-pixel derivative maps the backward filters; ws: block partials.
 * The backward gives d(total)/d(img1) for a device scalar gloss = d(total)/d(loss); img2 (the ground truth) gets none. */
long cidnet_ssim_ws_floats(int B, int C, int H, int W);
int cidnet_ssim_fwd(const float* img1, const float* img2, float weight, float* loss, float* dA, float* dB,
                    float* dC, float* ws, long ws_floats, int B, int C, int H, int W, void* stream);
int cidnet_ssim_bwd(const float* img1, const float* img2, const float* dA, const float* dB, const float* dC,
                    const float* gloss, float weight, float* gimg1, int B, int C, int H, int W, void* stream);
/* EdgeLoss ("next" row f1): EdgeLoss.forward, loss/losses.py:41-65: laplacian(z) = z - G(U(G(z))) with G the 5x5
 * replicate-padded blur outer([.05 .25 .4 .25 .05]) and U = even pixels x4; loss = mean((lap(x) - lap(y))^2) * weight.
 * fwd saves lap = laplacian(x - y) (B,C,H,W); bwd gives d(total)/dx = gloss * 2 weight / n * (lap - G^T U G^T lap).
 * ws: B*C*H*W + 2048 floats. */
long cidnet_edge_ws_floats(int B, int C, int H, int W);
int cidnet_edge_fwd(const float* x, const float* y, float weight, float* loss, float* lap, float* ws,
                    long ws_floats, int B, int C, int H, int W, void* stream);
int cidnet_edge_bwd(const float* lap, const float* gloss, float weight, float* gx, float* ws, long ws_floats,
                    int B, int C, int H, int W, void* stream);
/* ---- "next" row f4: VGG19 perceptual loss pieces (loss/vgg_arch.py:217-239, loss/losses.py:126-142; the 3x3
 * convolutions are cidnet_conv3x3).  The VGG weights are frozen (requires_grad=False, vgg_arch.py:203-206): data
 * gradients only.
 * normalize: y = ((range_norm ? (x + 1) / 2 : x) - mean[c]) / std[c], x (B,3,HW), ImageNet mean / std (:211-214).
 * bias_relu: x += bias[c] in place (the 'convN_M' feature, taken BEFORE the ReLU), act = max(x, 0) (act may be NULL,
 *   or x itself for an in-place ReLU).  relu_bwd: gx = act > 0 ? g : 0.
 * maxpool2: nn.MaxPool2d(2, 2) (:196), output floor(H/2) x floor(W/2); bwd routes to the first maximum in scan order.
 * mse_loss: loss (+)= weight * mean((a - b)^2), grad = weight * 2 (a - b) / n (criterion 'mse', train.py:192). */
int cidnet_vgg_normalize(const float* x, float* y, int range_norm, int B, long HW, void* stream);
int cidnet_vgg_normalize_bwd(const float* g, float* gx, int range_norm, int B, long HW, void* stream);
int cidnet_bias_relu(float* x, const float* bias, float* act, int B, int C, long HW, void* stream);
int cidnet_relu_bwd(const float* g, const float* act, float* gx, long n, void* stream);
int cidnet_maxpool2_fwd(const float* x, float* y, long planes, int H, int W, void* stream);
int cidnet_maxpool2_bwd(const float* x, const float* gy, float* gx, long planes, int H, int W, void* stream);
long cidnet_mse_ws_floats(void);
int cidnet_mse_loss(const float* a, const float* b, float* grad, float* loss, float weight, int accumulate, float* ws,
                    long ws_floats, long n, void* stream);
/* torch.optim.Adam step (train.py:166) over one flat buffer; g is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce).  step = 1-based update count. */
int cidnet_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, float grad_scale,
                     void* stream);

/* ---- K13: SpatialAttention of the MSSA variant (net/CIDNet_MSSA.py:10-25) ------------------------
 * out = x * sigmoid(conv7x7([mean_c x, max_c x])), w: (1,2,7,7).  stats (B,2,H,W), amax (B,H,W int32)
 * and att (B,1,H,W) are saved for the backward. */
int cidnet_sa_fwd(const float* x, const float* w, float* stats, int* amax, float* att, float* out,
                  int B, int C, int H, int W, void* stream);
long cidnet_sa_bwd_ws_floats(int B, int H, int W);
int cidnet_sa_bwd(const float* x, const float* w, const float* stats, const int* amax,
                  const float* att, const float* g, float* gx, float* gw, float* ws, long ws_floats,
                  int B, int C, int H, int W, void* stream);

/* ---- K14: pieces of the TNSM variant (net/TNSM.py) ----------------------------------------------
 * adaptive avg + max pool to 1x1 (:39-40); amax = flat index of the maximum per (b,c) plane */
int cidnet_global_pool_fwd(const float* x, float* avg, float* mx, int* amax, int B, int C, long HW,
                           void* stream);
int cidnet_global_pool_bwd(const float* gavg, const float* gmx, const int* amax, float* gx, int B, int C,
                           long HW, void* stream);
/* DynamicNoiseMap global branch (:43-47) folded with noise_branch[2] (Wn) and final_conv (wf):
 * gf = sigmoid(W2 (relu(W1 avg) + relu(W1 mx))), vrow[k] = sum_c wf[c] gf[c] Wn[c][k], so that
 * noise_map = sigmoid(vrow . leaky(dw3x3(x))).  hsum: (B,R,3) saved for the backward. */
int cidnet_noise_global_fwd(const float* avg, const float* mx, const float* W1, const float* W2,
                            const float* Wn, const float* wf, float* hsum, float* gf, float* vrow,
                            int B, int C, int R, void* stream);
/* per-sample parameter-gradient partials (sum over B with cidnet_sum_rows) + gavg, gmx (B,C) */
int cidnet_noise_global_bwd(const float* avg, const float* mx, const float* W1, const float* W2,
                            const float* Wn, const float* wf, const float* hsum, const float* gf,
                            const float* gvrow, float* gW1_b, float* gW2_b, float* gWn_b, float* gwf_b,
                            float* gavg, float* gmx, int B, int C, int R, void* stream);
/* mode 0: y = leaky_relu(a, 0.2); 1: y = a * (b > 0 ? 1 : 0.2) (a = grad, b = forward output);
 * 2: y = sigmoid(a); 3: y = a * b * (1 - b) (a = grad, b = forward output) */
int cidnet_elementwise(int mode, const float* a, const float* b, float* y, long n, void* stream);
/* nm[b][p] = sigmoid(sum_c v[b][c] t[b][c][p]) and its backward (gt (B,C,HW), gv (B,C)) */
int cidnet_rowdot_sigmoid_fwd(const float* t, const float* v, float* nm, int B, int C, long HW,
                              void* stream);
int cidnet_rowdot_sigmoid_bwd(const float* gnm, const float* nm, const float* t, const float* v,
                              float* gt, float* gv, int B, int C, long HW, void* stream);
/* v' = v * sigmoid(ws[c] * nm) (:103-112); vin may be a channel slice (batch stride vin_bs) */
int cidnet_modulate_fwd(const float* vin, long vin_bs, const float* nm, const float* ws, float* vout,
                        int B, int C, long HW, void* stream);
long cidnet_modulate_bwd_ws_floats(int B, int C, long HW);
int cidnet_modulate_bwd(const float* vin, long vin_bs, const float* nm, const float* ws,
                        const float* gvout, float* gvin, long gvin_bs, float* gnm, float* gws_part,
                        int B, int C, long HW, void* stream);
/* out = nm * a + (1 - nm) * d (AdaptiveFilter :165-166) and its backward */
int cidnet_blend_fwd(const float* a, const float* d, const float* nm, float* out, int B, int C, long HW,
                     void* stream);
int cidnet_blend_bwd(const float* a, const float* d, const float* nm, const float* g, float* ga,
                     float* gd, float* gnm, int B, int C, long HW, void* stream);
/* F.interpolate(bilinear, align_corners=False) (CIDNet_TNSM.py:258); dst / gdst may be channel
 * slices of a wider tensor (batch stride in floats) */
int cidnet_resize_bilinear_fwd(const float* src, float* dst, long dst_bs, int B, int C, int Hi, int Wi,
                               int Ho, int Wo, void* stream);
int cidnet_resize_bilinear_bwd(const float* gdst, long gdst_bs, float* gsrc, int B, int C, int Hi,
                               int Wi, int Ho, int Wo, void* stream);

/* ---- the TNSM variant's extra objective (train_tnsm.py:68-72) ---------------------------------------------------
 * loss = weight * (mean|noise_map - (1 - sigmoid(mean_c|out_rgb - im|))| + mean|d_x noise_map| + mean|d_y noise_map|),
 * noise_map (B,C,H,W) = the fused noise map CIDNet_TNSM.forward returns in train mode (C = 3), out_rgb / im (B,3,H,W).
 * One pass: loss (device scalar) and the gradients wrt noise_map and out_rgb (either may be NULL).
 * ws: cidnet_tnsm_noise_loss_ws_floats() floats. */
long cidnet_tnsm_noise_loss_ws_floats(void);
int cidnet_tnsm_noise_loss(const float* noise_map, const float* out_rgb, const float* im, float weight, float* loss,
                           float* g_noise, float* g_out, float* ws, long ws_floats, int B, int C, int H, int W,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIDNET_HIP_H */
