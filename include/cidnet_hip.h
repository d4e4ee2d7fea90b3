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

#define CIDNET_ABI_VERSION 1

int cidnet_abi_version(void);

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

#ifdef __cplusplus
}
#endif
#endif /* CIDNET_HIP_H */
