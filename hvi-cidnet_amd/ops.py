"""torch.autograd.Function wrappers over the C ABI (include/cidnet_hip.h).

Every op launches hand-written HIP kernels on the caller's current stream through ctypes; torch only
provides device memory, the stream and the autograd tape.  Inputs must be fp32 tensors on a ROCm
device -- anything else raises (no CPU / ATen fallback exists in this package).
"""
import ctypes

import torch

from ._lib import lib

_vp = ctypes.c_void_p


def _check(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("hvi-cidnet_amd ops run only on a ROCm device (got a CPU tensor); "
                               "there is no CPU fallback -- use the oracle under oracle/ for CPU checks")
        if t.dtype != torch.float32:
            raise RuntimeError(f"hvi-cidnet_amd ops are fp32 (got {t.dtype})")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return _vp(t.data_ptr()) if t is not None else None


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _f(x):
    return ctypes.c_float(float(x))


# --------------------------------------------------------------------------------------------
# K1 / K2: HVI colour transform
# --------------------------------------------------------------------------------------------
class HVITFn(torch.autograd.Function):
    """RGB -> HVI, reference RGB_HVI.HVIT (net/HVI_transform.py:16-47)."""

    @staticmethod
    def forward(ctx, img, density_k):
        _check(img, density_k)
        img = _c(img)
        B, C, H, W = img.shape
        if C != 3:
            raise RuntimeError("HVIT expects (B,3,H,W)")
        out = torch.empty_like(img)
        lib().call("cidnet_hvit_fwd", _p(img), _p(density_k), _p(out), None, B, H, W, _stream())
        ctx.save_for_backward(img, density_k)
        return out

    @staticmethod
    def backward(ctx, g):
        img, k = ctx.saved_tensors
        B, _, H, W = img.shape
        g = _c(g)
        need_x, need_k = ctx.needs_input_grad
        gx = torch.empty_like(img) if need_x else None
        gk = torch.empty_like(k) if need_k else None
        n = lib().raw("cidnet_hvit_bwd_ws_floats")()
        ws = torch.empty(n, device=img.device, dtype=torch.float32) if need_k else None
        lib().call("cidnet_hvit_bwd", _p(img), _p(k), _p(g), _p(gx), _p(gk), _p(ws), n if need_k else 0, B, H, W,
                   _stream())
        return gx, gk


def hvit_branch_code(img, density_k):
    """uint8 mask-decision code per pixel (see cidnet_hvit_fwd); test/diagnostic helper."""
    _check(img, density_k)
    img = _c(img)
    B, _, H, W = img.shape
    out = torch.empty_like(img)
    code = torch.empty((B, H, W), device=img.device, dtype=torch.uint8)
    lib().call("cidnet_hvit_fwd", _p(img), _p(density_k), _p(out), _p(code), B, H, W, _stream())
    return code


class PHVITFn(torch.autograd.Function):
    """HVI -> RGB, reference RGB_HVI.PHVIT (net/HVI_transform.py:49-122), optionally with the
    residual cat([hv, iv]) + hvi of net/CIDNet.py:119 fused.  k: python float or 1-element device
    tensor (never differentiated, as in the reference)."""

    @staticmethod
    def forward(ctx, hvi, hv, iv, k, gated, alpha_s, gated2, alpha):
        _check(hvi, hv, iv)
        hvi = _c(hvi)
        hv = _c(hv) if hv is not None else None
        iv = _c(iv) if iv is not None else None
        B, C, H, W = hvi.shape
        if C != 3:
            raise RuntimeError("PHVIT expects (B,3,H,W)")
        k_dev = k if isinstance(k, torch.Tensor) else None
        k_host = 0.0 if k_dev is not None else float(k)
        out = torch.empty_like(hvi)
        lib().call("cidnet_phvit_fwd", _p(hv), _p(iv), _p(hvi), _p(k_dev), _f(k_host), int(bool(gated)), _f(alpha_s),
                   int(bool(gated2)), _f(alpha), _p(out), None, B, H, W, _stream())
        ctx.save_for_backward(hvi, hv, iv, k_dev)
        ctx.cfg = (k_host, bool(gated), float(alpha_s), bool(gated2), float(alpha))
        return out

    @staticmethod
    def backward(ctx, g):
        hvi, hv, iv, k_dev = ctx.saved_tensors
        k_host, gated, alpha_s, gated2, alpha = ctx.cfg
        B, _, H, W = hvi.shape
        g = _c(g)
        g_hvi = torch.empty_like(hvi) if ctx.needs_input_grad[0] else None
        fused = hv is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        g_hv = torch.empty_like(hv) if fused else None
        g_iv = torch.empty_like(iv) if fused else None
        if g_hvi is None and not fused:
            return (None,) * 8
        lib().call("cidnet_phvit_bwd", _p(hv), _p(iv), _p(hvi), _p(k_dev), _f(k_host), int(gated), _f(alpha_s),
                   int(gated2), _f(alpha), _p(g), _p(g_hvi), _p(g_hv), _p(g_iv), B, H, W, _stream())
        return g_hvi, g_hv, g_iv, None, None, None, None, None


def phvit_sextant(hvi, k):
    _check(hvi)
    hvi = _c(hvi)
    B, _, H, W = hvi.shape
    out = torch.empty_like(hvi)
    sx = torch.empty((B, H, W), device=hvi.device, dtype=torch.uint8)
    lib().call("cidnet_phvit_fwd", None, None, _p(hvi), None, _f(k), 0, _f(1.3), 0, _f(1.0), _p(out), _p(sx), B, H, W,
               _stream())
    return sx
