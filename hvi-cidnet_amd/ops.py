"""torch.autograd.Function wrappers over the C ABI (include/cidnet_hip.h).

Every op launches hand-written HIP kernels on the caller's current stream through ctypes; torch only
provides device memory, the stream and the autograd tape.  Inputs must be fp32 tensors on a ROCm
device -- anything else raises (no CPU / ATen fallback exists in this package).
"""
import ctypes
import os

import torch

from ._lib import lib

_vp = ctypes.c_void_p


def needs_grad(*ts):
    """True when a backward pass can follow: decided by the MODULE before .apply(), because inside
    Function.forward grad mode is always off and ctx.needs_input_grad stays True for requires_grad
    parameters even under torch.no_grad()."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _check(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("hvi-cidnet_amd ops run only on a ROCm device (got a CPU tensor); "
                               "there is no CPU fallback -- use the oracle under oracle/ for CPU checks")
        if t.dtype != torch.float32:
            raise RuntimeError(f"hvi-cidnet_amd ops are fp32 (got {t.dtype})")


def _check_act(*ts):
    """activations inside an LCA block: fp32, or bf16 in the bf16 mode (STORAGE)"""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("hvi-cidnet_amd ops run only on a ROCm device (got a CPU tensor); "
                               "there is no CPU fallback -- use the oracle under oracle/ for CPU checks")
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"hvi-cidnet_amd activations are fp32 or bf16 (got {t.dtype})")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return _vp(t.data_ptr()) if t is not None else None


# torch.cuda.current_stream() builds a Stream object through three Python layers (~8 us; ~500 calls per training step = a
# quarter of the host's enqueue time).  The raw handle comes from one C call; the Python form stays as the fallback.
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) if os.environ.get("CIDNET_PY_STREAM") != "1" else None
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream_handle():
    """hipStream_t of the current stream of the current device, as an int"""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return _vp(_stream_handle())


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
        gk = grad_like(k) if need_k else None                  # the arena slot when density_k has one use per step
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


# --------------------------------------------------------------------------------------------
# shared plumbing for the conv / attention ops
# --------------------------------------------------------------------------------------------
_WS = {}


def _ws(n_floats, device):
    """Stream-ordered scratch buffer (grown on demand, reused by consecutive launches); one per stream,
    because the two CIDNet branches may run on two streams concurrently."""
    key = (device.type, device.index, _stream_handle())
    t = _WS.get(key)
    if t is None or t.numel() < n_floats:
        t = torch.empty(max(int(n_floats), 1 << 20), device=device, dtype=torch.float32)
        _WS[key] = t
    return t


# Weight-gradient GEMMs feed nothing but the optimizer, so they may run on a third stream and overlap the
# data-gradient chain.  Only safe when the caller (dp.DataParallelTrainer) waits for that stream before it
# touches the gradients, and only for single-use parameters written in place into the gradient arena.
_WG = {"on": False, "streams": {}}


def enable_wgrad_stream(on=True):
    _WG["on"] = bool(on)


def wgrad_stream(device):
    key = (device.type, device.index)
    st = _WG["streams"].get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _WG["streams"][key] = st
    return st


def _is_wgrad_out(t):
    """heuristic-free marker: grad_like() tags the tensors it hands out"""
    return getattr(t, "_cidnet_wgrad", False)


def _in_arena(t):
    base, n, flat_g, _ = _ARENA
    gb = flat_g.data_ptr()
    return gb <= t.data_ptr() < gb + 4 * n


def _offload_wgrad(tensors, fn):
    """tensors = (inputs..., outputs...) of fn; outputs are the trailing grad_like() results.  Falls back to
    the caller's stream unless every weight-gradient output lives in the arena (otherwise autograd would
    read it on the caller's stream before this stream has written it)."""
    if not _WG["on"] or _ARENA is None or not all(_in_arena(t) for t in tensors if t.dim() >= 1 and _is_wgrad_out(t)):
        fn()
        return
    main = torch.cuda.current_stream()
    st = wgrad_stream(tensors[0].device)
    st.wait_stream(main)
    with torch.cuda.stream(st):
        fn()
    for t in tensors:
        t.record_stream(st)


def _po(t, off_floats=0):
    return _vp(t.data_ptr() + 4 * int(off_floats))


def _raw(name, *a):
    return lib().raw(name)(*a)


# ---- bf16 storage mode (row J1 / BASELINE.json configs[2]) ------------------------------------------------------------
# "f32": every tensor fp32 (the parity mode).  "bf16": the hidden tensors of the IEL chain -- project_in's output, u, the
# gate and their three gradients, 32 of the 80 GB a step moves -- are STORED as bfloat16 (round to nearest even); every
# kernel still computes in fp32, weights, weight gradients and all other activations stay fp32.  Results then differ from
# the reference by bf16 rounding of those tensors (~4e-3 relative on a hidden value): a separate, looser tolerance tier.
STORAGE = {"hidden": torch.float32}


def lca_dtype(x):
    """storage type of the tensors that live INSIDE an LCA block (LayerNorm outputs, the CAB's q / k / v before and after
    their depthwise convs, the IEL hidden tensors, and all their gradients) for a block whose input is x: bf16 in the bf16
    mode where the typed LayerNorm kernels cover the shape (CIDNet's 36 / 72 / 144-channel levels), else fp32.  The
    residual stream between blocks, the conv / resampling activations and every parameter / gradient stay fp32."""
    # (set_storage_dtype("bf16") alone -- bf16 storage with fp32-exact products -- keeps to the IEL hidden tensors: the fp32-MFMA
    # 1x1 kernel it then runs on has no bf16-in / bf16-out form)
    if STORAGE["hidden"] == torch.float32 or MATH["levels"] != 1 or x.dim() != 4 or x.dtype != torch.float32:
        return torch.float32
    B, C, H, W = x.shape
    return torch.bfloat16 if _raw("cidnet_ln_cf_typed_supported", B, C, H * W) else torch.float32


def set_storage_dtype(name):
    if name not in ("f32", "bf16"):
        raise ValueError("storage dtype must be 'f32' or 'bf16'")
    STORAGE["hidden"] = torch.bfloat16 if name == "bf16" else torch.float32


# ---- bf16 arithmetic mode -------------------------------------------------------------------------------------------------
# MATH["levels"] = how many bf16 levels of each fp32 operand enter the matrix-core products of the convolutions (1x1, dense
# 3x3, their weight gradients).  3: the exact three-way split, six products per fp32 product -- results within fp32
# rounding of an fp32 convolution: the PARITY mode and the default.  1: operands rounded to nearest bf16, one product, fp32
# accumulation and fp32 output -- the arithmetic of a torch.autocast(bfloat16) convolution (BASELINE.json configs[2]): the
# activation split in the kernels disappears (11 VALU ops per element pair -> 1) and the matrix-core work drops to a sixth.
MATH = {"levels": 3}


def set_math_levels(n):
    if n not in (1, 3):
        raise ValueError("operand levels must be 3 (fp32-exact split products) or 1 (bf16 operands)")
    MATH["levels"] = int(n)


def set_precision(name):
    """'f32': the parity mode (fp32 storage, fp32-exact products).  'bf16': BASELINE.json configs[2] -- bf16 matrix-core
    operands (one product per term, fp32 accumulation) and bf16 storage of the tensors listed at STORAGE; its own
    tolerance tier (tests/test_fullsize_gpu.py)."""
    if name not in ("f32", "bf16"):
        raise ValueError("precision must be 'f32' or 'bf16'")
    set_storage_dtype(name)
    set_math_levels(1 if name == "bf16" else 3)


def _dt(t):
    """element-type code of the C ABI (CIDNET_F32 = 0, CIDNET_BF16 = 1)"""
    return 1 if t.dtype == torch.bfloat16 else 0


def _pe(t, off_elems=0):
    return _vp(t.data_ptr() + t.element_size() * int(off_elems))


# ---- prepared weight operands kept across calls ------------------------------------------------------------------------
# The split-product kernels (csrc/pwx.hip, csrc/conv3x.hip) read their weights split into bf16 levels in MFMA fragment
# order.  Preparing them per call costs a 5 us launch in front of every conv: ~125 per training step (98 shared-weight
# 1x1 convs, 24 dense 3x3 convs; forward and backward use different orders of the same weights), latency the branch
# streams cannot hide behind anything.  With the cache ON a prepared operand lives in its own buffer, keyed by (weight
# address, strides, shape, orientation): the first use prepares it, and whoever changes the weights re-prepares ALL of them
# in two launches (refresh_prepared_weights: one batched kernel per family) -- dp.DataParallelTrainer does after its fused
# Adam update, which is the only writer of the flat parameter buffer.  An entry also remembers the tensor's autograd
# version counter, so in-place updates through torch (load_state_dict, torch.optim) are seen and re-prepared on use;
# writes through `.data` or raw pointers are not -- hence OFF unless a caller that owns the weight updates turns it on
# (enable_prepared_weights).  Per-sample operands (the attention maps) are never cached.
class _Prepared:
    __slots__ = ("buf", "kind", "row", "version", "tensor")


_PREP = {"on": False, "entries": {}, "tables": {}, "stats": [0, 0]}
# kernel families with a prepared weight operand: cache kind -> prefix of their _prep / _prep_blocks / _prep_batch entry points
_PREP_FAMILY = {"pw": "cidnet_pw_conv_bf16x3", "c3": "cidnet_conv3x3_bf16x3"}
_PREP_MAX_ENTRIES = 1024


def clear_prepared_weights():
    if _PREP["entries"] and torch.cuda.is_initialized():
        torch.cuda.synchronize()                                 # a refresh or a reader may still be in flight
    _PREP["entries"].clear()
    _PREP["tables"].clear()


def enable_prepared_weights(on=True):
    """global switch; switching (on or off) drops every cached operand"""
    clear_prepared_weights()
    _PREP["on"] = bool(on)


class prepared_weights:
    """`with ops.prepared_weights(True): ...` -- the cache is consulted only inside the block (entries survive it): the
    trainer wraps its own forward+backward passes, so a model(x) call outside them (validation, a user's own weight
    surgery between steps) prepares per call as ever."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _PREP["on"]
        _PREP["on"] = self.on
        return self

    def __exit__(self, *exc):
        _PREP["on"] = self.prev
        return False


def _prepared(kind, w, w_off, w_ms, w_ks, flip, M, K, n_floats, prepare):
    """-> buffer holding the prepared operand of weight view (w, w_off, strides), or None with the cache off"""
    if not _PREP["on"]:
        return None
    src = w.data_ptr() + 4 * int(w_off)
    key = (kind, src, int(w_ms), int(w_ks), int(flip), M, K)
    e = _PREP["entries"].get(key)
    ver = w._version
    if e is not None and e.version == ver:
        _PREP["stats"][0] += 1
        return e.buf
    _PREP["stats"][1] += 1
    if e is None:
        if len(_PREP["entries"]) >= _PREP_MAX_ENTRIES:          # weights re-homed over and over: start again
            _PREP["entries"].clear()
        e = _Prepared()
        e.buf = torch.empty(int(n_floats), device=w.device, dtype=torch.float32)
        e.kind = kind
        e.row = (src, e.buf.data_ptr(), M, K, int(w_ms), int(w_ks), 0, int(flip))
        e.tensor = w
        _PREP["entries"][key] = e
        _PREP["tables"].pop((kind, w.device), None)
    e.version = ver
    prepare(e.buf)
    return e.buf


def refresh_prepared_weights(device=None):
    """Re-prepare every cached operand from the current weights: one launch per kernel family on the current stream.  For
    the owner of the weight updates (dp.DataParallelTrainer calls it right after the optimizer step)."""
    if not _PREP["entries"]:
        return
    by = {}
    for e in _PREP["entries"].values():
        if device is None or e.buf.device == device:
            by.setdefault((e.kind, e.buf.device), []).append(e)
    for (kind, dev), es in by.items():
        tab = _PREP["tables"].get((kind, dev))
        if tab is None:
            blocks_of = lib().raw(_PREP_FAMILY[kind] + "_prep_blocks")
            rows, first = [], 0
            for e in es:
                r = list(e.row)
                r[6] = first
                first += int(blocks_of(r[2], r[3]))
                rows.append(r)
            tab = (torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), first)
            _PREP["tables"][(kind, dev)] = tab
        t, n, total = tab
        with torch.cuda.device(dev):
            lib().call(_PREP_FAMILY[kind] + "_prep_batch", _p(t), n, total, _stream())
        for e in es:
            e.version = e.tensor._version


_BILINEAR_TABS = {}


def _bilinear_tabs(device, Hi, Wi, Ho, Wo):
    """per-axis tap tables of the bilinear adjoint: a function of the four sizes only, computed once per shape and device
    (they were two 5 us launches in front of each of the 12 adjoint launches of a step)"""
    key = (device.type, device.index, Hi, Wi, Ho, Wo)
    t = _BILINEAR_TABS.get(key)
    if t is None:
        n = _raw("cidnet_bilinear_bwd_ws_floats", Hi, Wi)
        t = torch.empty(int(n), device=device, dtype=torch.float32)
        lib().call("cidnet_bilinear_bwd_tabs", _p(t), t.numel(), Hi, Wi, Ho, Wo, _stream())
        # the creating stream may not be the one of later users: make the tables globally visible once
        torch.cuda.current_stream().synchronize()
        _BILINEAR_TABS[key] = t
    return t


# The 1x1 convs run on the BF16 matrix cores with exact three-way split operands (csrc/pwx.hip, third version: every wave
# independent, operands split in registers).  It is faster than the fp32-MFMA kernel (pw.hip) on every 1x1 shape of the
# step but one (tools/sweep_pw_step.py, profiles/r04_b_sweep_pw_step.txt: 1.03 - 2.0x; 3.9 - 4.6 TB/s on the 200x300 planes
# where pw.hip reaches 2.2 - 4.0): the fan-out from 36 to 190 channels (IEL project_in at the 36-channel level, 0.94x), whose
# two k-blocks of weights per 12 output tiles cost more than the fp32 kernel's register-resident panel.
# CIDNET_PW_BF16X3=0 switches it off.
PW_BF16X3 = {"on": os.environ.get("CIDNET_PW_BF16X3", "1") == "1"}


def pw_bf16x3_wins(M, K, HW=0):
    return MATH["levels"] == 1 or not (K <= 40 and M >= 150)


def pw_conv_bf16x3(x, x_off, x_bs, w, w_off, w_bs, w_ms, w_ks, y, y_off, y_bs, B, M, K, HW, res=None, r_off=0, r_bs=0):
    per_sample = int(w_bs != 0)
    nb = B if per_sample else 1
    n = _raw("cidnet_pw_conv_bf16x3_ws_floats", B, M, K, per_sample)

    def prepare(buf):
        lib().call("cidnet_pw_conv_bf16x3_prep", _po(w, w_off), w_bs, w_ms, w_ks, _p(buf), buf.numel(), nb, M, K, _stream())
    pre = _prepared("pw", w, w_off, w_ms, w_ks, 0, M, K, n, prepare) if not per_sample else None
    if pre is None:
        pre = _ws(n, x.device)
        prepare(pre)
    lv = MATH["levels"]
    # x / y stored as bf16 (bf16 mode, levels == 1): read without conversion / rounded on store; offsets and strides in elements
    lib().call("cidnet_pw_conv_bf16x3_pre_t", _pe(x, x_off), _dt(x), x_bs, _p(pre), per_sample, _pe(y, y_off), _dt(y), y_bs,
               _po(res, r_off) if res is not None else None, r_bs, B, M, K, HW, lv, lv, _stream())


def pw_conv(x, x_off, x_bs, w, w_off, w_bs, w_ms, w_ks, y, y_off, y_bs, B, M, K, HW, res=None, r_off=0, r_bs=0):
    """x / y may be bf16 tensors (offsets and strides in elements)"""
    if PW_BF16X3["on"] and ((x.dtype == torch.float32 and y.dtype == torch.float32) or MATH["levels"] == 1) \
            and pw_bf16x3_wins(M, K, HW) \
            and _raw("cidnet_pw_conv_bf16x3_supported", M, K, HW):
        return pw_conv_bf16x3(x, x_off, x_bs, w, w_off, w_bs, w_ms, w_ks, y, y_off, y_bs, B, M, K, HW, res, r_off, r_bs)
    lib().call("cidnet_pw_conv_t", _pe(x, x_off), _dt(x), x_bs, _po(w, w_off), w_bs, w_ms, w_ks, _pe(y, y_off), _dt(y), y_bs,
               _po(res, r_off) if res is not None else None, r_bs, B, M, K, HW, _stream())


# The 1x1 weight gradients of fp32 operands run on the BF16 matrix cores as exact split products (csrc/pw.hip, pw_wgrad_kernel
# <.., BF3>); CIDNET_PW_WGRAD_BF16X3=0 keeps them on the fp32 MFMA (flag CIDNET_WGRAD_FP32_MFMA of the C ABI).
PW_WGRAD_BF16X3 = {"on": os.environ.get("CIDNET_PW_WGRAD_BF16X3", "1") == "1"}


def pw_wgrad(dy, dy_off, dy_bs, x, x_off, x_bs, dw, dw_off, dw_ld, B, M, N, HW, per_sample=False):
    n = _raw("cidnet_pw_wgrad_ws_floats", B, M, N, HW)
    ws = _ws(n, dy.device)
    flags = 4 if MATH["levels"] == 1 else (0 if PW_WGRAD_BF16X3["on"] else 2)      # CIDNET_WGRAD_BF16_1LEVEL / _FP32_MFMA
    lib().call("cidnet_pw_wgrad_t", _pe(dy, dy_off), _dt(dy), dy_bs, _pe(x, x_off), _dt(x), x_bs, _po(dw, dw_off), dw_ld,
               int(per_sample), flags, _p(ws), ws.numel(), B, M, N, HW, _stream())


def dw3x3(inp, w1, w2, csplit, out, B, C, H, W, flip=False, addend=None):
    """inp / addend / out share one storage type (fp32 or bf16)"""
    lib().call("cidnet_dw3x3_t", _p(inp), _p(w1), _p(w2), csplit, _p(addend), _p(out), _dt(inp), int(flip), B, C, H, W, _stream())


def dw3x3_wgrad(inp, gout, gw1, gw2, csplit, B, C, H, W):
    n = _raw("cidnet_dw3x3_wgrad_ws_floats", B, C, H, W)
    ws = _ws(n, inp.device)
    lib().call("cidnet_dw3x3_wgrad", _p(inp), _p(gout), _p(gw1), _p(gw2), csplit, _p(ws), ws.numel(), B, C, H, W, _stream())


def dw3x3_bwd(inp, gout, w1, w2, csplit, gin, gw1, gw2, B, C, H, W, addend=None):
    """data gradient (+ addend) and weight gradient of a depthwise 3x3 in one pass; inp / gout / gin (/ addend) share one
    storage type (fp32 or bf16)"""
    n = _raw("cidnet_dw3x3_wgrad_ws_floats", B, C, H, W)
    ws = _ws(n, inp.device)
    lib().call("cidnet_dw3x3_bwd_t", _p(inp), _p(gout), _p(w1), _p(w2), csplit, _p(addend), _p(gin), _dt(inp), _p(gw1), _p(gw2),
               _p(ws), ws.numel(), B, C, H, W, _stream())


# Dense 3x3 convs whose input-channel count is a multiple of 36 (all of CIDNet's 36 / 72 / 144-channel layers) run on the
# BF16 matrix cores with exact three-way split operands (csrc/conv3x.hip: results within fp32 rounding of the fp32-MFMA
# kernel, error against fp64 equal or smaller).  CIDNET_CONV3_BF16X3=0 selects the fp32-MFMA kernel (csrc/conv3.hip).
CONV3_BF16X3 = {"on": os.environ.get("CIDNET_CONV3_BF16X3", "1") == "1"}


def conv3x3(x, w, y, B, M, K, H, W, w_ms, w_ks, flip=False, replicate=False, addend=None, x_bs=None):
    """y = conv3x3(x) (+ addend, in the kernel's epilogue; only for layers with more than 4 channels on both sides).
    x_bs: batch stride of x in floats when x is a plane slice of a wider tensor (the I stem reads plane 2 of hvi)"""
    x_bs = K * H * W if x_bs is None else int(x_bs)
    if CONV3_BF16X3["on"] and not replicate and x_bs == K * H * W and min(M, K) > 4 and _raw("cidnet_conv3x3_bf16x3_supported", M, K) \
            and CONV3_BF16X3.get("filter", lambda *a: True)(M, K, H, W):
        n = _raw("cidnet_conv3x3_bf16x3_ws_floats", M, K)

        def prepare(buf):
            lib().call("cidnet_conv3x3_bf16x3_prep", _p(w), w_ms, w_ks, int(flip), _p(buf), buf.numel(), M, K, _stream())
        pre = _prepared("c3", w, 0, w_ms, w_ks, int(flip), M, K, n, prepare)
        if pre is None:
            pre = _ws(n, x.device)
            prepare(pre)
        lv = MATH["levels"]
        lib().call("cidnet_conv3x3_bf16x3_pre_lv", _p(x), K * H * W, _p(pre), _p(addend), M * H * W, _p(y), M * H * W, B, M, K, H, W,
                   lv, lv, _stream())
        return
    if addend is None:
        lib().call("cidnet_conv3x3", _p(x), x_bs, _p(w), w_ms, w_ks, int(flip), int(replicate), _p(y), M * H * W, B, M, K,
                   H, W, _stream())
    else:
        lib().call("cidnet_conv3x3_add", _p(x), x_bs, _p(w), w_ms, w_ks, int(flip), int(replicate), _p(addend), M * H * W,
                   _p(y), M * H * W, B, M, K, H, W, _stream())


# weight gradient of the same layers: csrc/conv3xw.hip (operands split once per staged tile, fragments by transposing LDS
# reads); CIDNET_CONV3_WGRAD_BF16X3=0 selects the fp32-MFMA kernel (csrc/conv3.hip)
CONV3_WGRAD_BF16X3 = {"on": os.environ.get("CIDNET_CONV3_WGRAD_BF16X3", "1") == "1"}


def conv3x3_wgrad(dy, x, dw, B, M, N, H, W, replicate=False, x_bs=None):
    x_bs = N * H * W if x_bs is None else int(x_bs)
    if CONV3_WGRAD_BF16X3["on"] and not replicate and x_bs == N * H * W and M > 4 and _raw("cidnet_conv3x3_wgrad_bf16x3_supported", M, N, H, W):
        n = _raw("cidnet_conv3x3_wgrad_bf16x3_ws_floats", B, M, N, H, W)
        ws = _ws(n, dy.device)
        lib().call("cidnet_conv3x3_wgrad_bf16x3_lv", _p(dy), M * H * W, _p(x), N * H * W, _p(dw), _p(ws), ws.numel(), B, M, N, H, W,
                   MATH["levels"], _stream())
        return
    n = _raw("cidnet_conv3x3_wgrad_ws_floats", B, M, N, H, W)
    ws = _ws(n, dy.device)
    lib().call("cidnet_conv3x3_wgrad", _p(dy), M * H * W, _p(x), x_bs, int(replicate), _p(dw), _p(ws), ws.numel(), B, M,
               N, H, W, _stream())


def bilinear_bwd(dout, din, B, C, Hi, Wi, Ho, Wo):
    if _BILINEAR_TABS.get((dout.device.type, dout.device.index, Hi, Wi, Ho, Wo)) is None and torch.cuda.is_current_stream_capturing():
        # first use of a shape inside a capture: no synchronisation possible there, tables per call
        n = _raw("cidnet_bilinear_bwd_ws_floats", Hi, Wi)
        ws = _ws(n, dout.device)
        lib().call("cidnet_bilinear_bwd", _p(dout), _p(din), _p(ws), ws.numel(), B, C, Hi, Wi, Ho, Wo, _stream())
        return
    tabs = _bilinear_tabs(dout.device, Hi, Wi, Ho, Wo)
    lib().call("cidnet_bilinear_bwd_pre", _p(dout), _p(din), _p(tabs), B, C, Hi, Wi, Ho, Wo, _stream())


def prelu_bwd(go, pre, slope):
    dpre = torch.empty_like(pre)
    dslope = grad_like(slope)
    n = _raw("cidnet_prelu_bwd_ws_floats")
    ws = _ws(n, go.device)
    lib().call("cidnet_prelu_bwd", _p(go), _p(pre), _p(slope), _p(dpre), _p(dslope), _p(ws), ws.numel(), go.numel(), _stream())
    return dpre, dslope


# --------------------------------------------------------------------------------------------
# K3: LayerNorm (channels first)
# --------------------------------------------------------------------------------------------
# Backward of a 1x1 conv as ONE kernel (csrc/pwb.hip): data gradient and weight gradient from one read of gy.
# CIDNET_PW_BWD_FUSED=0 keeps the two separate launches (data gradient on the main stream, weight gradient on its own).
# It pays where gy is much larger than x (the IEL project_in: 190 against 36 channels, 168 vs 241 us); for the layers with
# M <= N or M = 2 N the two launches are as fast or faster (131 vs 124 us at 36 x 95) -- measured in the step: +1.2 % with
# both IEL convs fused, +1.7 % with project_in only.
PW_BWD_FUSED = {"on": os.environ.get("CIDNET_PW_BWD_FUSED", "1") == "1", "min_ratio": float(os.environ.get("CIDNET_PW_BWD_FUSED_MIN_RATIO", "4"))}


def pw_bwd_fused_ok(gy, x, w, M, N, HW):
    return (PW_BWD_FUSED["on"] and MATH["levels"] == 3 and M >= PW_BWD_FUSED["min_ratio"] * N and gy.dtype == torch.float32 and x.dtype == torch.float32 and w.dtype == torch.float32
            and gy.is_contiguous() and x.is_contiguous() and w.is_contiguous()
            and bool(_raw("cidnet_pw_bwd_fused_supported", M, N, HW)))


def pw_bwd_fused(gy, x, w, gx, dw, B, M, N, HW):
    """gx (B, N, HW) = W^T gy and dw (M, N) = sum gy x^T for y = W x with W (M, N); fp32, contiguous"""
    n = _raw("cidnet_pw_bwd_fused_ws_floats", B, M, N, HW)
    ws = _ws(n, gy.device)
    lib().call("cidnet_pw_bwd_fused", _p(gy), M * HW, _p(x), N * HW, _p(w), _p(gx), N * HW, _p(dw), _p(ws), ws.numel(), B, M, N, HW,
               _stream())


class LNUse:
    """Per-module bookkeeping for a LayerNorm that is applied several times per step (net/LCA.py:79-80,91-92: one `norm`
    per LCA, used on x, on y and on the CAB output).  With `acc` set (by dp.DataParallelTrainer, after its probing step has
    seen every forward use come back in the backward) the uses sum their weight / bias gradients IN PLACE in the gradient
    arena -- the first backward use overwrites, the others add (cidnet_ln_cf_bwd_res, accumulate) -- and only the last one
    hands the arena views to autograd: no separate accumulation launches (44 per step) and no copies into the arena (22)."""

    def __init__(self):
        self.fwd = 0          # uses in the current forward pass
        self.bwd = 0          # of which have run their backward
        self.acc = False
        self.probe = [0, 0]   # forward / backward uses counted while not in acc mode (the trainer's probe reads them)


def _ln_param_grads(ctx, weight, bias):
    """-> (gw, gb, accumulate, hand_over): the tensors the kernel writes, whether it adds, whether autograd gets them"""
    st = ctx.ln_state
    if st is not None and st.acc and _ARENA is not None:
        gw, gb = grad_like(weight), grad_like(bias)
        if _in_arena(gw) and _in_arena(gb):
            first = st.bwd == 0
            st.bwd += 1
            last = st.bwd >= st.fwd
            if last:
                st.fwd = st.bwd = 0
            return gw, gb, (not first), last
    if st is not None:
        st.probe[1] += 1
    return grad_like(weight), torch.empty_like(weight), False, True


def _ln_forward_count(ctx, state):
    """`state` is None unless a backward will follow (decided by the module: ops.needs_grad)"""
    ctx.ln_state = state
    if state is not None:
        if state.acc:
            state.fwd += 1
        else:
            state.probe[0] += 1


class LayerNormCFFn(torch.autograd.Function):
    """Reference: LayerNorm.forward (channels_first), net/transformer_utils.py:24-29."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, state=None, out_dtype=torch.float32):
        _check(x, weight, bias)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x, dtype=out_dtype)
        mean = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lib().call("cidnet_ln_cf_fwd_t", _p(x), _p(weight), _p(bias), _p(y), _dt(y), _p(mean), _p(rstd), B, C, H * W, _f(eps), _stream())
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        _ln_forward_count(ctx, state)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        gy = _c(gy)
        gw, gb, acc, hand = _ln_param_grads(ctx, weight, bias)
        # a bf16 gradient (bf16 mode) runs on the fused kernel, which always produces gx
        gx = torch.empty_like(x) if (ctx.needs_input_grad[0] or gy.dtype == torch.bfloat16) else None
        n = _raw("cidnet_ln_cf_bwd_ws_floats", C)
        ws = _ws(n, x.device)
        lib().call("cidnet_ln_cf_bwd_res_t", _p(x), _p(weight), _p(gy), _dt(gy), _p(mean), _p(rstd), None, _p(gx), _p(gw), _p(gb), int(acc),
                   _p(ws), ws.numel(), B, C, H * W, _stream())
        return (gx if ctx.needs_input_grad[0] else None), (gw if hand else None), (gb if hand else None), None, None, None


class LayerNormResFn(torch.autograd.Function):
    """LayerNorm of a pre-norm residual block (net/LCA.py:79,80,91,92): returns (norm(x), x).  The second output is x
    itself, to be used as the block's residual input; its gradient then arrives HERE instead of being summed with the
    LayerNorm's input gradient by a separate autograd accumulation pass, and the backward kernel adds it in place
    (cidnet_ln_cf_bwd_res)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, state=None, out_dtype=torch.float32):
        _check(x, weight, bias)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x, dtype=out_dtype)
        mean = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lib().call("cidnet_ln_cf_fwd_t", _p(x), _p(weight), _p(bias), _p(y), _dt(y), _p(mean), _p(rstd), B, C, H * W, _f(eps), _stream())
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        _ln_forward_count(ctx, state)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gres):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        gw, gb, acc, hand = _ln_param_grads(ctx, weight, bias)
        if gy is None:                      # only the residual path was used
            if not acc:
                gw.zero_(); gb.zero_()
            return gres, (gw if hand else None), (gb if hand else None), None, None, None
        gy = _c(gy)
        gres = _c(gres) if gres is not None else None
        gx = torch.empty_like(x)
        n = _raw("cidnet_ln_cf_bwd_ws_floats", C)
        ws = _ws(n, x.device)
        lib().call("cidnet_ln_cf_bwd_res_t", _p(x), _p(weight), _p(gy), _dt(gy), _p(mean), _p(rstd), _p(gres), _p(gx), _p(gw), _p(gb),
                   int(acc), _p(ws), ws.numel(), B, C, H * W, _stream())
        return gx, (gw if hand else None), (gb if hand else None), None, None, None


LN_DUAL = {"on": os.environ.get("CIDNET_LN_DUAL", "1") == "1"}


def ln_dual_supported(x):
    B, C, H, W = x.shape
    return LN_DUAL["on"] and x.is_cuda and x.dtype == torch.float32 and bool(_raw("cidnet_ln_cf_dual_supported", B, C, H * W))


class LayerNormDualFn(torch.autograd.Function):
    """Two LayerNorm modules on one tensor: (norm_a(x), norm_b(x), x).  In an LCA pair (net/CIDNet.py:83-84) every input is
    the x of its own block (norm_a, and the block's residual: third output, as in LayerNormResFn) and the y of the partner
    block (norm_b): one forward pass reads x once and writes both outputs, one backward pass reads x once and adds the
    three gradients.  Module a's parameter gradients follow the in-place protocol of LNUse (`state_a`); module b lives on the
    OTHER branch's stream, so its gradients are returned to autograd as fresh tensors (its own uses keep their protocol)."""

    @staticmethod
    def forward(ctx, x, w_a, b_a, w_b, b_b, eps, state_a=None):
        _check(x, w_a, b_a)
        _check(x, w_b, b_b)
        x = _c(x)
        B, C, H, W = x.shape
        ya, yb = torch.empty_like(x), torch.empty_like(x)
        mean = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lib().call("cidnet_ln_cf_fwd2", _p(x), _p(w_a), _p(b_a), _p(ya), _p(w_b), _p(b_b), _p(yb), _p(mean), _p(rstd), B, C, H * W,
                   _f(eps), _stream())
        ctx.save_for_backward(x, w_a, b_a, w_b, b_b, mean, rstd)
        _ln_forward_count(ctx, state_a)
        return ya, yb, x.view_as(x)

    @staticmethod
    def backward(ctx, gya, gyb, gres):
        x, w_a, b_a, w_b, b_b, mean, rstd = ctx.saved_tensors
        B, C, H, W = x.shape
        gw, gb, acc, hand = _ln_param_grads(ctx, w_a, b_a)
        gres = _c(gres) if gres is not None else None
        n1 = _raw("cidnet_ln_cf_bwd_ws_floats", C)
        if gya is not None and gyb is not None:
            gya, gyb = _c(gya), _c(gyb)
            gx = torch.empty_like(x)
            gw2, gb2 = torch.empty_like(w_b), torch.empty_like(b_b)
            ws = _ws(max(n1, _raw("cidnet_ln_cf_bwd2_ws_floats", C)), x.device)
            lib().call("cidnet_ln_cf_bwd2", _p(x), _p(w_a), _p(gya), _p(w_b), _p(gyb), _p(mean), _p(rstd), _p(gres), _p(gx), _p(gw),
                       _p(gb), int(acc), _p(gw2), _p(gb2), 0, _p(ws), ws.numel(), B, C, H * W, _stream())
            return gx, (gw if hand else None), (gb if hand else None), gw2, gb2, None, None
        # one of the two outputs was not used: the single-module kernel (or nothing) does the work
        ws = _ws(n1, x.device)
        gx, gw2, gb2 = gres, None, None
        if gya is not None:
            gx = torch.empty_like(x)
            lib().call("cidnet_ln_cf_bwd_res", _p(x), _p(w_a), _p(_c(gya)), _p(mean), _p(rstd), _p(gres), _p(gx), _p(gw), _p(gb),
                       int(acc), _p(ws), ws.numel(), B, C, H * W, _stream())
        elif not acc:
            gw.zero_(); gb.zero_()
        if gyb is not None:
            gx2 = torch.empty_like(x)
            gw2, gb2 = torch.empty_like(w_b), torch.empty_like(b_b)
            lib().call("cidnet_ln_cf_bwd_res", _p(x), _p(w_b), _p(_c(gyb)), _p(mean), _p(rstd), _p(gx), _p(gx2), _p(gw2), _p(gb2), 0,
                       _p(ws), ws.numel(), B, C, H * W, _stream())
            gx = gx2
        return gx, (gw if hand else None), (gb if hand else None), gw2, gb2, None, None


# --------------------------------------------------------------------------------------------
# K6/K7: cross-attention block with the residual:  out = x_res + CAB(xn, yn)
# --------------------------------------------------------------------------------------------
class CABResidualFn(torch.autograd.Function):
    """Reference: CAB.forward (net/LCA.py:19-41) plus the residual add of LCA.py:79 / :91."""

    @staticmethod
    def forward(ctx, x_res, xn, yn, temperature, wq, wq_dw, wkv, wkv_dw, wp, heads):
        _check(x_res, temperature, wq, wq_dw, wkv, wkv_dw, wp)
        _check_act(xn, yn)
        if xn.dtype != yn.dtype:
            raise RuntimeError("CAB: the two normalised inputs must share a storage type")
        x_res, xn, yn = _c(x_res), _c(xn), _c(yn)
        B, C, H, W = xn.shape
        HW = H * W
        dev = xn.device
        ch = C // heads
        # q | k | v before and after their depthwise convs take the storage type of the normalised inputs (bf16 in the bf16 mode)
        qkv0 = torch.empty((B, 3 * C, H, W), device=dev, dtype=xn.dtype)
        pw_conv(xn, 0, C * HW, wq, 0, 0, C, 1, qkv0, 0, 3 * C * HW, B, C, C, HW)
        pw_conv(yn, 0, C * HW, wkv, 0, 0, C, 1, qkv0, C * HW, 3 * C * HW, B, 2 * C, C, HW)
        qkv = torch.empty_like(qkv0)
        dw3x3(qkv0, wq_dw, wkv_dw, C, qkv, B, 3 * C, H, W)
        attn = torch.empty((B, heads, ch, ch), device=dev, dtype=torch.float32)
        shat = torch.empty_like(attn)
        nq = torch.empty((B, C), device=dev, dtype=torch.float32)
        nk = torch.empty_like(nq)
        M = torch.empty((B, C, C), device=dev, dtype=torch.float32)
        n = _raw("cidnet_attn_gram_ws_floats", B, C, heads, HW)
        ws = _ws(n, dev)
        lib().call("cidnet_attn_fwd_t", _p(qkv), _dt(qkv), _p(temperature), _p(wp), _p(attn), _p(shat), _p(nq), _p(nk), _p(M), _p(ws),
                   ws.numel(), B, C, heads, HW, 1, _stream())
        out = torch.empty_like(x_res)
        pw_conv(qkv, 2 * C * HW, 3 * C * HW, M, 0, C * C, C, 1, out, 0, C * HW, B, C, C, HW, res=x_res, r_bs=C * HW)
        ctx.save_for_backward(xn, yn, qkv0, qkv, attn, shat, nq, nk, M, temperature, wq, wq_dw, wkv, wkv_dw, wp)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        xn, yn, qkv0, qkv, attn, shat, nq, nk, M, temperature, wq, wq_dw, wkv, wkv_dw, wp = ctx.saved_tensors
        heads = ctx.heads
        B, C, H, W = xn.shape
        HW = H * W
        dev = xn.device
        g = _c(g)
        dqkv = torch.empty_like(qkv)
        # dv = M^T g ; dM = g v^T
        pw_conv(g, 0, C * HW, M, 0, C * C, 1, C, dqkv, 2 * C * HW, 3 * C * HW, B, C, C, HW)
        dM = torch.empty_like(M)
        pw_wgrad(g, 0, C * HW, qkv, 2 * C * HW, 3 * C * HW, dM, 0, C, B, C, C, HW, per_sample=True)
        dwp_b = torch.empty_like(M)
        dT_b = torch.empty((B, heads), device=dev, dtype=torch.float32)
        wqk = torch.empty((B, 2 * C, 2 * C), device=dev, dtype=torch.float32)
        lib().call("cidnet_attn_bwd", _p(dM), _p(wp), _p(attn), _p(shat), _p(nq), _p(nk), _p(temperature), _p(dwp_b), _p(dT_b),
                   _p(wqk), B, C, heads, 1, _stream())
        g_wp = grad_like(wp)
        lib().call("cidnet_sum_rows", _p(dwp_b), B, C * C, _p(g_wp), _stream())
        g_T = grad_like(temperature)
        lib().call("cidnet_sum_rows", _p(dT_b), B, heads, _p(g_T), _stream())
        # [dq;dk] = Wqk_b [q;k]
        pw_conv(qkv, 0, 3 * C * HW, wqk, 0, 4 * C * C, 2 * C, 1, dqkv, 0, 3 * C * HW, B, 2 * C, 2 * C, HW)
        # depthwise backward
        g_wq_dw = grad_like(wq_dw)
        g_wkv_dw = grad_like(wkv_dw)
        dqkv0 = torch.empty_like(qkv0)
        dw3x3_bwd(qkv0, dqkv, wq_dw, wkv_dw, C, dqkv0, g_wq_dw, g_wkv_dw, B, 3 * C, H, W)
        # pointwise backward
        g_wq = grad_like(wq)
        g_wkv = grad_like(wkv)
        _offload_wgrad((dqkv0, xn, yn, g_wq, g_wkv), lambda: (
            pw_wgrad(dqkv0, 0, 3 * C * HW, xn, 0, C * HW, g_wq, 0, C, B, C, C, HW),
            pw_wgrad(dqkv0, C * HW, 3 * C * HW, yn, 0, C * HW, g_wkv, 0, C, B, 2 * C, C, HW)))
        dxn = dyn = None
        if ctx.needs_input_grad[1]:
            dxn = torch.empty_like(xn)
            pw_conv(dqkv0, 0, 3 * C * HW, wq, 0, 0, 1, C, dxn, 0, C * HW, B, C, C, HW)
        if ctx.needs_input_grad[2]:
            dyn = torch.empty_like(yn)
            pw_conv(dqkv0, C * HW, 3 * C * HW, wkv, 0, 0, 1, C, dyn, 0, C * HW, B, C, 2 * C, HW)
        return g, dxn, dyn, g_T, g_wq, g_wq_dw, g_wkv, g_wkv_dw, g_wp, None


# --------------------------------------------------------------------------------------------
# K8: IEL gated FFN:  out = [res +] Wout * gate(dw(Win * xn))
# --------------------------------------------------------------------------------------------
# tile-resident IEL kernels (csrc/iel.hip): "fwd" the fused forward, "bwd" the fused backward pair; a training
# forward only takes the fused path when the fused backward exists (it saves xn and u, not pin / gate)
# Off by default: at fp32 the fused forward measures 500-570 us against the chain's 460 us at 8x36x200x300 (csrc/iel.hip).
IEL_FUSED = {"fwd": os.environ.get("CIDNET_IEL_FUSED", "0") == "1", "bwd": False}


class IELFn(torch.autograd.Function):
    """Reference: IEL.forward (net/LCA.py:60-67); `res` is the residual of I_LCA (LCA.py:92)."""

    @staticmethod
    def forward(ctx, xn, res, w_in, w_dw, w_dw1, w_dw2, w_out, train=True):
        _check(res, w_in, w_dw, w_dw1, w_dw2, w_out)
        _check_act(xn)
        xn = _c(xn)
        res = _c(res) if res is not None else None
        B, C, H, W = xn.shape
        HW = H * W
        h = w_dw1.shape[0]
        dev = xn.device
        ctx.fused = False
        if IEL_FUSED["fwd"] and xn.dtype == torch.float32 and (not train or IEL_FUSED["bwd"]) and _raw("cidnet_iel_fwd_supported", C, h):
            # tile-resident kernel (csrc/iel.hip): the hidden tensors never reach HBM; u is written only for the backward
            u = torch.empty((B, 2 * h, H, W), device=dev, dtype=torch.float32) if train else None
            out = torch.empty_like(xn)
            lib().call("cidnet_iel_fwd", _p(xn), _p(res), _p(w_in), _p(w_dw), _p(w_dw1), _p(w_dw2), _p(w_out), _p(u), _p(out),
                       B, C, h, H, W, _stream())
            if train:
                ctx.fused = True
                ctx.save_for_backward(xn, u, w_in, w_dw, w_dw1, w_dw2, w_out)
            ctx.has_res = res is not None
            return out
        hd = STORAGE["hidden"] if STORAGE["hidden"] != torch.float32 else xn.dtype    # fp32, or bf16 in the bf16 mode
        pin = torch.empty((B, 2 * h, H, W), device=dev, dtype=hd)
        pw_conv(xn, 0, C * HW, w_in, 0, 0, C, 1, pin, 0, 2 * h * HW, B, 2 * h, C, HW)
        u = torch.empty_like(pin) if train else None      # inference: u (read only by the backward) is not stored
        gate = torch.empty((B, h, H, W), device=dev, dtype=hd)
        lib().call("cidnet_iel_dw_gate_fwd_t", _p(pin), _p(w_dw), _p(w_dw1), _p(w_dw2), _p(u) if train else None, _p(gate), _dt(pin),
                   B, h, H, W, _stream())
        out = torch.empty_like(xn, dtype=torch.float32)    # back on the fp32 residual stream
        pw_conv(gate, 0, h * HW, w_out, 0, 0, h, 1, out, 0, C * HW, B, C, h, HW, res=res, r_bs=C * HW)
        if train:
            ctx.save_for_backward(xn, pin, u, gate, w_in, w_dw, w_dw1, w_dw2, w_out)
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, go):
        xn, pin, u, gate, w_in, w_dw, w_dw1, w_dw2, w_out = ctx.saved_tensors
        B, C, H, W = xn.shape
        HW = H * W
        h = w_dw1.shape[0]
        go = _c(go)
        g_wout = grad_like(w_out)
        dg = torch.empty_like(gate)                     # hidden tensors keep the storage type they were saved in
        if pw_bwd_fused_ok(go, gate, w_out, C, h, HW):
            pw_bwd_fused(go, gate, w_out, dg, g_wout, B, C, h, HW)
        else:
            _offload_wgrad((go, gate, g_wout), lambda: pw_wgrad(go, 0, C * HW, gate, 0, h * HW, g_wout, 0, h, B, C, h, HW))
            pw_conv(go, 0, C * HW, w_out, 0, 0, 1, h, dg, 0, h * HW, B, h, C, HW)
        g_dw1 = grad_like(w_dw1)
        g_dw2 = grad_like(w_dw2)
        du = torch.empty_like(u)                        # gate backward + dwconv1/2 backward in one pass
        n = _raw("cidnet_iel_gate_dw_bwd_ws_floats", B, h, H, W)
        ws = _ws(n, u.device)
        lib().call("cidnet_iel_gate_dw_bwd_t", _p(u), _p(w_dw1), _p(w_dw2), _p(dg), _p(du), _dt(u), _p(g_dw1), _p(g_dw2), _p(ws),
                   ws.numel(), B, h, H, W, _stream())
        g_dw = grad_like(w_dw)
        dpin = torch.empty_like(u)
        dw3x3_bwd(pin, du, w_dw, None, 2 * h, dpin, g_dw, None, B, 2 * h, H, W)
        g_win = grad_like(w_in)
        dxn = None
        if ctx.needs_input_grad[0] and pw_bwd_fused_ok(dpin, xn, w_in, 2 * h, C, HW):
            dxn = torch.empty_like(xn)
            pw_bwd_fused(dpin, xn, w_in, dxn, g_win, B, 2 * h, C, HW)
        else:
            _offload_wgrad((dpin, xn, g_win), lambda: pw_wgrad(dpin, 0, 2 * h * HW, xn, 0, C * HW, g_win, 0, C, B, 2 * h, C, HW))
            if ctx.needs_input_grad[0]:
                dxn = torch.empty_like(xn)
                pw_conv(dpin, 0, 2 * h * HW, w_in, 0, 0, 1, C, dxn, 0, C * HW, B, C, 2 * h, HW)
        return dxn, (go if ctx.has_res else None), g_win, g_dw, g_dw1, g_dw2, g_wout, None


# --------------------------------------------------------------------------------------------
# K9/K10/K11: down / up blocks and the replicate-pad stem / head convs
# --------------------------------------------------------------------------------------------
class DownFn(torch.autograd.Function):
    """Reference: NormDownsample.forward, net/transformer_utils.py:38-43: conv3x3 -> bilinear x0.5
    (align_corners=True) -> PReLU."""

    @staticmethod
    def forward(ctx, x, w, slope, train=True):
        _check(x, w, slope)
        x = _c(x)
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        t = torch.empty((B, Co, H, W), device=x.device, dtype=torch.float32)
        conv3x3(x, w, t, B, Co, Ci, H, W, 9 * Ci, 9)
        # train=False (inference): the pre-activation (read only by the backward) is not stored
        out = torch.empty((B, Co, H // 2, W // 2), device=x.device, dtype=torch.float32)
        pre = torch.empty_like(out) if train else None
        lib().call("cidnet_down_prelu_fwd", _p(t), _p(slope), _p(pre), _p(out), B, Co, H, W, _stream())
        if train:
            ctx.save_for_backward(x, w, slope, pre)
        return out

    @staticmethod
    def backward(ctx, go):
        x, w, slope, pre = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        go = _c(go)
        dpre, dslope = prelu_bwd(go, pre, slope)
        dt = torch.empty((B, Co, H, W), device=x.device, dtype=torch.float32)
        bilinear_bwd(dpre, dt, B, Co, H, W, H // 2, W // 2)
        gw = grad_like(w)
        _offload_wgrad((dt, x, gw), lambda: conv3x3_wgrad(dt, x, gw, B, Co, Ci, H, W))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            conv3x3(dt, w, dx, B, Ci, Co, H, W, 9, 9 * Ci, flip=True)
        return dx, gw, dslope, None


class DownResFn(torch.autograd.Function):
    """NormDownsample whose input also feeds a skip connection (net/CIDNet.py:80-81,85-86: `i_jump0 = i_enc0`, ...):
    returns (down(x), x).  The skip consumer takes the second output, so its gradient arrives HERE and is added in the
    epilogue of the data-gradient conv instead of by autograd's separate accumulation pass over the tensor (two
    69M-element passes per step at 400x600, two more at 200x300)."""

    @staticmethod
    def forward(ctx, x, w, slope):
        out = DownFn.forward(ctx, x, w, slope, True)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, go, gres):
        x, w, slope, pre = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        if go is None:
            return gres, None, None
        if gres is None or min(Ci, Co) <= 4:
            dx, gw, dslope, _ = DownFn.backward(ctx, go)
            if gres is not None and dx is not None:
                dx = dx + gres
            return dx, gw, dslope
        dpre, dslope = prelu_bwd(_c(go), pre, slope)
        dt = torch.empty((B, Co, H, W), device=x.device, dtype=torch.float32)
        bilinear_bwd(dpre, dt, B, Co, H, W, H // 2, W // 2)
        gw = grad_like(w)
        _offload_wgrad((dt, x, gw), lambda: conv3x3_wgrad(dt, x, gw, B, Co, Ci, H, W))
        dx = torch.empty_like(x)
        conv3x3(dt, w, dx, B, Ci, Co, H, W, 9, 9 * Ci, flip=True, addend=_c(gres))
        return dx, gw, dslope


class UpFn(torch.autograd.Function):
    """Reference: NormUpsample.forward, net/transformer_utils.py:62-70: conv3x3 -> bilinear x2 ->
    cat(skip) -> conv1x1 -> PReLU.  The 1x1 conv is split over the concat and its up-sampled half is
    applied BEFORE the (linear) bilinear map, at a quarter of the pixels; no concat is materialised."""

    @staticmethod
    def forward(ctx, x, skip, w, w_up, slope, train=True):
        _check(x, skip, w, w_up, slope)
        x, skip = _c(x), _c(skip)
        B, Ci, h, wd = x.shape
        Co = w.shape[0]
        if skip.shape != (B, Co, 2 * h, 2 * wd):
            raise RuntimeError(f"NormUpsample: skip {tuple(skip.shape)} does not match upsampled {(B, Co, 2 * h, 2 * wd)}")
        dev = x.device
        t = torch.empty((B, Co, h, wd), device=dev, dtype=torch.float32)
        conv3x3(x, w, t, B, Co, Ci, h, wd, 9 * Ci, 9)
        z = torch.empty_like(t)
        pw_conv(t, 0, Co * h * wd, w_up, 0, 0, 2 * Co, 1, z, 0, Co * h * wd, B, Co, Co, h * wd)
        out = torch.empty_like(skip)
        pre = torch.empty_like(skip) if train else None
        lib().call("cidnet_pw_conv_up_prelu", _p(skip), Co * 4 * h * wd, _po(w_up, Co), 2 * Co, 1, _p(z), _p(slope), _p(out),
                   _p(pre), B, Co, Co, h, wd, _stream())
        if train:
            ctx.save_for_backward(x, skip, w, w_up, slope, t, pre)
        return out

    @staticmethod
    def backward(ctx, go):
        x, skip, w, w_up, slope, t, pre = ctx.saved_tensors
        B, Ci, h, wd = x.shape
        Co = w.shape[0]
        HWl, HWh = h * wd, 4 * h * wd
        go = _c(go)
        dpre, dslope = prelu_bwd(go, pre, slope)
        g_wup = grad_like(w_up)
        _offload_wgrad((dpre, skip, g_wup), lambda: pw_wgrad(dpre, 0, Co * HWh, skip, 0, Co * HWh, g_wup, Co, 2 * Co, B, Co, Co, HWh))
        dskip = None
        if ctx.needs_input_grad[1]:
            dskip = torch.empty_like(skip)
            pw_conv(dpre, 0, Co * HWh, w_up, Co, 0, 1, 2 * Co, dskip, 0, Co * HWh, B, Co, Co, HWh)
        dz = torch.empty_like(t)
        bilinear_bwd(dpre, dz, B, Co, h, wd, 2 * h, 2 * wd)
        _offload_wgrad((dz, t, g_wup), lambda: pw_wgrad(dz, 0, Co * HWl, t, 0, Co * HWl, g_wup, 0, 2 * Co, B, Co, Co, HWl))
        dt = torch.empty_like(t)
        pw_conv(dz, 0, Co * HWl, w_up, 0, 0, 1, 2 * Co, dt, 0, Co * HWl, B, Co, Co, HWl)
        gw = grad_like(w)
        _offload_wgrad((dt, x, gw), lambda: conv3x3_wgrad(dt, x, gw, B, Co, Ci, h, wd))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            conv3x3(dt, w, dx, B, Ci, Co, h, wd, 9, 9 * Ci, flip=True)
        return dx, dskip, gw, g_wup, dslope, None


class RepConv3x3Fn(torch.autograd.Function):
    """Reference: nn.ReplicationPad2d(1) + valid 3x3 conv (net/CIDNet.py:21-24,32-35,39-42,50-53)."""

    @staticmethod
    def forward(ctx, x, w):
        _check(x, w)
        B, Ci, H, W = x.shape
        # one plane of a wider tensor (the I stem reads hvi[:, 2:3], net/CIDNet.py:75): read in place through its batch stride
        if not (Ci == 1 and x.stride(3) == 1 and x.stride(2) == W and x.stride(0) >= H * W):
            x = _c(x)
        Co = w.shape[0]
        y = torch.empty((B, Co, H, W), device=x.device, dtype=torch.float32)
        conv3x3(x, w, y, B, Co, Ci, H, W, 9 * Ci, 9, replicate=True, x_bs=x.stride(0))
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        go = _c(go)
        gw = grad_like(w)
        _offload_wgrad((go, x, gw), lambda: conv3x3_wgrad(go, x, gw, B, Co, Ci, H, W, replicate=True, x_bs=x.stride(0)))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            conv3x3(go, w, dx, B, Ci, Co, H, W, 9, 9 * Ci, flip=True)
            lib().call("cidnet_conv3x3_replicate_dgrad_fix", _p(go), _p(w), _p(dx), B, Co, Ci, H, W, _stream())
        return dx, gw


class SpatialAttentionFn(torch.autograd.Function):
    """Reference: SpatialAttention.forward of the MSSA variant, net/CIDNet_MSSA.py:20-25."""

    @staticmethod
    def forward(ctx, x, w):
        _check(x, w)
        x = _c(x)
        B, C, H, W = x.shape
        if tuple(w.shape) != (1, 2, 7, 7):
            raise RuntimeError("SpatialAttention: only the 7x7 kernel of CIDNet_MSSA is implemented")
        dev = x.device
        stats = torch.empty((B, 2, H, W), device=dev, dtype=torch.float32)
        amax = torch.empty((B, H, W), device=dev, dtype=torch.int32)
        att = torch.empty((B, 1, H, W), device=dev, dtype=torch.float32)
        out = torch.empty_like(x)
        lib().call("cidnet_sa_fwd", _p(x), _p(w), _p(stats), _p(amax), _p(att), _p(out), B, C, H, W, _stream())
        ctx.save_for_backward(x, w, stats, amax, att)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, stats, amax, att = ctx.saved_tensors
        B, C, H, W = x.shape
        g = _c(g)
        gx = torch.empty_like(x)
        gw = grad_like(w)
        n = _raw("cidnet_sa_bwd_ws_floats", B, H, W)
        ws = _ws(n, x.device)
        lib().call("cidnet_sa_bwd", _p(x), _p(w), _p(stats), _p(amax), _p(att), _p(g), _p(gx), _p(gw), _p(ws), ws.numel(), B, C,
                   H, W, _stream())
        return gx, gw


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _check(a, b)
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        lib().call("cidnet_add", _p(a), _p(b), _p(y), a.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


class HviFanoutFn(torch.autograd.Function):
    """The HVI image feeds three consumers (net/CIDNet.py:73-77,119): the HV stem (whole image), the I stem (plane 2) and
    the residual of the output.  Returns (hvi, hvi[:, 2:3], hvi) as views; the backward sums the three gradients in ONE
    kernel (cidnet_hvi_grad_sum) instead of autograd's two accumulation passes plus the slice backward's fill and copy."""

    @staticmethod
    def forward(ctx, hvi):
        _check(hvi)
        hvi = _c(hvi)
        return hvi.view_as(hvi), hvi[:, 2:3], hvi.view_as(hvi)

    @staticmethod
    def backward(ctx, ga, gi, gc):
        ref = ga if ga is not None else gc
        if ref is None:
            if gi is None:
                return None
            B, _, H, W = gi.shape
            out = torch.empty((B, 3, H, W), device=gi.device, dtype=torch.float32)
        else:
            B, _, H, W = ref.shape
            out = torch.empty_like(ref, memory_format=torch.contiguous_format)
        ga, gi, gc = (_c(t) if t is not None else None for t in (ga, gi, gc))
        lib().call("cidnet_hvi_grad_sum", _p(ga), _p(gc), _p(gi), _p(out), B, H * W, _stream())
        return out


# --------------------------------------------------------------------------------------------
# training-step pieces: L1 loss (loss + gradient in one pass) and fused flat Adam
# --------------------------------------------------------------------------------------------
def _scaled(x, g, mult):
    """x * g * mult with the 0-dim upstream gradient g read on the device"""
    y = torch.empty_like(x)
    g = g.to(torch.float32).reshape(1).contiguous()
    lib().call("cidnet_scale", _p(x), _p(g), _f(mult), _p(y), x.numel(), _stream())
    return y


def snapshot(t):
    """detached device copy of a small fp32 tensor through cidnet_scale (keeps ATen's copy kernels out of the step)"""
    t = t.detach()
    _check(t)
    y = torch.empty_like(t)
    lib().call("cidnet_scale", _p(_c(t)), None, _f(1.0), _p(y), t.numel(), _stream())
    return y


class L1LossFn(torch.autograd.Function):
    """mean(|out - gt|): reference L1Loss (loss/losses.py:10-20, loss_utils.py l1_loss, reduction='mean').
    Differentiable in both arguments: train.py:62 does not detach gt_hvi = model.HVIT(gt_rgb), so the HVI-space
    terms send gradient to density_k through the target as well."""

    @staticmethod
    def forward(ctx, out, gt):
        _check(out, gt)
        out, gt = _c(out), _c(gt)
        grad = torch.empty_like(out) if any(ctx.needs_input_grad) else None
        loss = torch.empty((), device=out.device, dtype=torch.float32)
        n = _raw("cidnet_l1_loss_ws_floats")
        ws = _ws(n, out.device)
        lib().call("cidnet_l1_loss", _p(out), _p(gt), _p(grad), _p(loss), _p(ws), ws.numel(), out.numel(), _stream())
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        # g is the scalar d(total)/d(loss); 1.0 in the benchmark step
        return (_scaled(grad, g, 1.0) if ctx.needs_input_grad[0] else None,
                _scaled(grad, g, -1.0) if ctx.needs_input_grad[1] else None)


class SSIMLossFn(torch.autograd.Function):
    """(1 - mean(ssim_map(img1, img2))) * weight: reference SSIM.forward (loss/losses.py:166-190) with map_ssim
    (loss/loss_utils.py:125-145); 11x11 Gaussian window, zero padding.  The SSIM map is symmetric in its two images,
    so the gradient wrt img2 (needed when img2 = HVIT(gt) depends on density_k) is the img1 formula with the roles
    swapped: one more forward pass of the kernel with (img2, img1), only when that gradient is requested."""

    @staticmethod
    def _fwd(img1, img2, weight):
        B, C, H, W = img1.shape
        dA, dB, dC = torch.empty_like(img1), torch.empty_like(img1), torch.empty_like(img1)
        loss = torch.empty((), device=img1.device, dtype=torch.float32)
        n = _raw("cidnet_ssim_ws_floats", B, C, H, W)
        ws = _ws(n, img1.device)
        lib().call("cidnet_ssim_fwd", _p(img1), _p(img2), _f(weight), _p(loss), _p(dA), _p(dB), _p(dC), _p(ws), ws.numel(), B, C, H, W,
                   _stream())
        return loss, dA, dB, dC

    @staticmethod
    def forward(ctx, img1, img2, weight):
        _check(img1, img2)
        img1, img2 = _c(img1), _c(img2)
        if img1.shape != img2.shape or img1.dim() != 4:
            raise RuntimeError("SSIM: img1 and img2 must be (B,C,H,W) tensors of the same shape")
        loss, dA, dB, dC = SSIMLossFn._fwd(img1, img2, weight)
        ctx.save_for_backward(img1, img2, dA, dB, dC)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    def backward(ctx, g):
        img1, img2, dA, dB, dC = ctx.saved_tensors
        B, C, H, W = img1.shape
        g = g.to(torch.float32).reshape(1).contiguous()
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(img1)
            lib().call("cidnet_ssim_bwd", _p(img1), _p(img2), _p(dA), _p(dB), _p(dC), _p(g), _f(ctx.weight), _p(gx), B, C, H, W, _stream())
        if ctx.needs_input_grad[1]:
            _, eA, eB, eC = SSIMLossFn._fwd(img2, img1, ctx.weight)
            gy = torch.empty_like(img2)
            lib().call("cidnet_ssim_bwd", _p(img2), _p(img1), _p(eA), _p(eB), _p(eC), _p(g), _f(ctx.weight), _p(gy), B, C, H, W, _stream())
        return gx, gy, None


class EdgeLossFn(torch.autograd.Function):
    """mean((laplacian(x) - laplacian(y))^2) * weight: reference EdgeLoss.forward (loss/losses.py:41-65).  The loss
    depends on x - y only (the Laplacian is linear), so d/dy = -d/dx."""

    @staticmethod
    def forward(ctx, x, y, weight):
        _check(x, y)
        x, y = _c(x), _c(y)
        if x.shape != y.shape or x.dim() != 4:
            raise RuntimeError("EdgeLoss: x and y must be (B,C,H,W) tensors of the same shape")
        B, C, H, W = x.shape
        lap = torch.empty_like(x)
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        n = _raw("cidnet_edge_ws_floats", B, C, H, W)
        ws = _ws(n, x.device)
        lib().call("cidnet_edge_fwd", _p(x), _p(y), _f(weight), _p(loss), _p(lap), _p(ws), ws.numel(), B, C, H, W, _stream())
        ctx.save_for_backward(lap)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    def backward(ctx, g):
        (lap,) = ctx.saved_tensors
        B, C, H, W = lap.shape
        g = g.to(torch.float32).reshape(1).contiguous()
        gx = torch.empty_like(lap)
        n = _raw("cidnet_edge_ws_floats", B, C, H, W)
        ws = _ws(n, lap.device)
        lib().call("cidnet_edge_bwd", _p(lap), _p(g), _f(ctx.weight), _p(gx), _p(ws), ws.numel(), B, C, H, W, _stream())
        gy = None
        if ctx.needs_input_grad[1]:
            gy = torch.empty_like(gx)
            lib().call("cidnet_scale", _p(gx), None, _f(-1.0), _p(gy), gx.numel(), _stream())
        return (gx if ctx.needs_input_grad[0] else None), gy, None


class TNSMNoiseLossFn(torch.autograd.Function):
    """weight * (noise_consistency_loss + noise_smoothing_loss) of the TNSM training script (train_tnsm.py:68-72):
    consistency = mean|noise_map - (1 - sigmoid(mean_c|output_rgb - im1|))|, smoothing = mean|d_x| + mean|d_y| of the
    map.  Differentiable in noise_map and output_rgb (im1 is the network's input, never a leaf that needs it)."""

    @staticmethod
    def forward(ctx, noise_map, output_rgb, im1, weight):
        _check(noise_map, output_rgb, im1)
        noise_map, output_rgb, im1 = _c(noise_map), _c(output_rgb), _c(im1)
        B, C, H, W = noise_map.shape
        if output_rgb.shape != (B, 3, H, W) or im1.shape != output_rgb.shape:
            raise RuntimeError("TNSM noise loss: noise_map (B,C,H,W) with output_rgb / im1 (B,3,H,W)")
        need_n, need_o = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_n = torch.empty_like(noise_map) if need_n else None
        g_o = torch.empty_like(output_rgb) if need_o else None
        loss = torch.empty((), device=noise_map.device, dtype=torch.float32)
        n = _raw("cidnet_tnsm_noise_loss_ws_floats")
        ws = _ws(n, noise_map.device)
        lib().call("cidnet_tnsm_noise_loss", _p(noise_map), _p(output_rgb), _p(im1), _f(weight), _p(loss), _p(g_n), _p(g_o), _p(ws),
                   ws.numel(), B, C, H, W, _stream())
        ctx.save_for_backward(g_n, g_o)
        return loss

    @staticmethod
    def backward(ctx, g):
        g_n, g_o = ctx.saved_tensors
        return (_scaled(g_n, g, 1.0) if g_n is not None else None, _scaled(g_o, g, 1.0) if g_o is not None else None, None, None)


# VGG19 feature stack up to conv4_4 (loss/vgg_arch.py:105-110: names; torchvision cfg 'E'): (name, Cin, Cout) or "pool"
VGG19_LAYERS = (("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool",
                ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256), "pool",
                ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv4_4", 512, 512), "pool",
                ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), ("conv5_4", 512, 512))


def vgg_plan(layer_names):
    """the prefix of VGG19_LAYERS that reaches the deepest requested 'convN_M' feature"""
    names = [l[0] if l != "pool" else "pool" for l in VGG19_LAYERS]
    last = max(names.index(n) for n in layer_names)
    return VGG19_LAYERS[:last + 1]


class PerceptualLossFn(torch.autograd.Function):
    """perceptual_weight * sum_k layer_weight[k] * mse(vgg_k(x), vgg_k(gt)) with the features taken at the conv outputs
    BEFORE the ReLU: reference PerceptualLoss.forward (loss/losses.py:126-142, criterion 'mse', style_weight 0) over
    VGGFeatureExtractor.forward (loss/vgg_arch.py:217-239).  gt is detached there (:134) and the VGG weights are frozen,
    so the only gradient is d/dx.  params = (w, b) per conv layer of the plan, in order."""

    @staticmethod
    def forward(ctx, x, gt, layer_names, layer_weights, perceptual_weight, range_norm, use_input_norm, *params):
        _check(x, gt, *params)
        x, gt = _c(x), _c(gt)
        B, C, H, W = x.shape
        if C != 3 or gt.shape != x.shape:
            raise RuntimeError("PerceptualLoss expects two (B,3,H,W) images of the same shape")
        plan = vgg_plan(layer_names)
        dev = x.device
        loss = torch.zeros((), device=dev, dtype=torch.float32)
        nws = _raw("cidnet_mse_ws_floats")

        def run(img, feats_gt):
            """-> (features or None, saved); with feats_gt given, the loss and the feature gradients are formed on the way"""
            t = torch.empty_like(img)
            if use_input_norm:
                lib().call("cidnet_vgg_normalize", _p(img), _p(t), int(bool(range_norm)), B, H * W, _stream())
            elif range_norm:
                raise NotImplementedError("range_norm without use_input_norm is not used by the reference's training loop")
            else:
                t = img
            h, w, pi = H, W, 0
            feats, saved, first = {}, [], True
            for li, layer in enumerate(plan):
                if layer == "pool":
                    y = torch.empty((B, t.shape[1], h // 2, w // 2), device=dev, dtype=torch.float32)
                    lib().call("cidnet_maxpool2_fwd", _p(t), _p(y), B * t.shape[1], h, w, _stream())
                    saved.append(("pool", t, h, w))
                    t, h, w = y, h // 2, w // 2
                    continue
                name, ci, co = layer
                wt, bs = params[2 * pi], params[2 * pi + 1]
                pi += 1
                y = torch.empty((B, co, h, w), device=dev, dtype=torch.float32)
                conv3x3(t, wt, y, B, co, ci, h, w, 9 * ci, 9)
                is_feat, is_last = name in layer_names, li == len(plan) - 1
                if is_feat:
                    act = None if is_last else torch.empty_like(y)
                    lib().call("cidnet_bias_relu", _p(y), _p(bs), _p(act), B, co, h * w, _stream())
                    if feats_gt is None:
                        feats[name] = y
                        gf = None
                    else:
                        gf = torch.empty_like(y)
                        ws = _ws(nws, dev)
                        wk = float(perceptual_weight) * float(layer_weights[layer_names.index(name)])
                        lib().call("cidnet_mse_loss", _p(y), _p(feats_gt[name]), _p(gf), _p(loss), _f(wk), 0 if first else 1, _p(ws),
                                   ws.numel(), y.numel(), _stream())
                        first = False
                    saved.append(("conv", wt, ci, co, h, w, gf, act))
                    t = act
                else:
                    lib().call("cidnet_bias_relu", _p(y), _p(bs), _p(y), B, co, h * w, _stream())
                    saved.append(("conv", wt, ci, co, h, w, None, y))
                    t = y
            return feats, saved

        feats_gt, _ = run(gt, None)
        need = ctx.needs_input_grad[0]
        _, saved = run(x, feats_gt)
        del feats_gt
        ctx.saved = saved if need else None
        ctx.meta = (B, H, W, bool(range_norm), bool(use_input_norm))
        return loss

    @staticmethod
    def backward(ctx, g):
        B, H, W, range_norm, use_input_norm = ctx.meta
        saved = ctx.saved
        d = None                               # gradient wrt the OUTPUT (post-ReLU / pooled) of the layer being walked
        for idx in range(len(saved) - 1, -1, -1):
            rec = saved[idx]
            if rec[0] == "pool":
                _, tin, h, w = rec
                gx = torch.empty_like(tin)
                lib().call("cidnet_maxpool2_bwd", _p(tin), _p(d), _p(gx), tin.shape[0] * tin.shape[1], h, w, _stream())
                d = gx
                continue
            _, wt, ci, co, h, w, gf, act = rec
            # gradient wrt the pre-activation (conv output + bias): through the ReLU, plus the feature term
            if d is not None:
                dp = torch.empty_like(d)
                lib().call("cidnet_relu_bwd", _p(d), _p(act), _p(dp), d.numel(), _stream())
                if gf is not None:
                    lib().call("cidnet_add", _p(dp), _p(gf), _p(dp), dp.numel(), _stream())
            else:
                dp = gf                          # deepest layer: only its feature term
            dx = torch.empty((B, ci, h, w), device=dp.device, dtype=torch.float32)
            conv3x3(dp, wt, dx, B, ci, co, h, w, 9, 9 * ci, flip=True)
            d = dx
        if use_input_norm:
            gx = torch.empty_like(d)
            lib().call("cidnet_vgg_normalize_bwd", _p(d), _p(gx), int(range_norm), B, H * W, _stream())
            d = gx
        ctx.saved = None
        return (_scaled(d, g, 1.0), None, None, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 7)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _check(p, g, m, v)
    lib().call("cidnet_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), _f(lr), _f(beta1), _f(beta2), _f(eps),
               _f(weight_decay), int(step), _f(grad_scale), _stream())


# --------------------------------------------------------------------------------------------
# gradient arena: weight gradients are written straight into one flat buffer (the all-reduce /
# optimizer operand) instead of 191 separately allocated tensors
# --------------------------------------------------------------------------------------------
_ARENA = None


def set_grad_arena(flat_p, flat_g, exclude_ptrs=()):
    """Parameters must be views into `flat_p`; their gradients are then produced as the matching
    views of `flat_g`.  `exclude_ptrs`: data_ptr()s of parameters used more than once per step
    (their gradients must be accumulated by autograd instead)."""
    global _ARENA
    _ARENA = None if flat_p is None else (flat_p.data_ptr(), flat_p.numel(), flat_g, frozenset(exclude_ptrs))


_GL_COUNT = None      # {data_ptr: number of grad_like() calls}: probe for parameters used more than once


def start_grad_probe():
    global _GL_COUNT
    _GL_COUNT = {}


def stop_grad_probe():
    global _GL_COUNT
    c, _GL_COUNT = _GL_COUNT, None
    return c or {}


def grad_like(w):
    if _GL_COUNT is not None:
        _GL_COUNT[w.data_ptr()] = _GL_COUNT.get(w.data_ptr(), 0) + 1
    if _ARENA is not None:
        base, n, flat_g, excl = _ARENA
        off = (w.data_ptr() - base) // 4
        if 0 <= off < n and w.data_ptr() not in excl and w.is_contiguous():
            g = flat_g[off:off + w.numel()].view(w.shape)
            g._cidnet_wgrad = True
            return g
    g = torch.empty_like(w)
    g._cidnet_wgrad = True
    return g


# --------------------------------------------------------------------------------------------
# generic conv pieces used by the TNSM variant (net/TNSM.py)
# --------------------------------------------------------------------------------------------
class PwConvFn(torch.autograd.Function):
    """Y = W[:, col_off:col_off+K] * X (1x1 conv on a column slice of a wider weight, bias-free)."""

    @staticmethod
    def forward(ctx, x, w, col_off, K):
        _check(x, w)
        x = _c(x)
        B, Kx, H, W = x.shape
        M, Kw = w.shape[0], w.shape[1]
        if Kx != K:
            raise RuntimeError("PwConvFn: channel mismatch")
        y = torch.empty((B, M, H, W), device=x.device, dtype=torch.float32)
        pw_conv(x, 0, K * H * W, w, col_off, 0, Kw, 1, y, 0, M * H * W, B, M, K, H * W)
        ctx.save_for_backward(x, w)
        ctx.meta = (col_off, K)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        col_off, K = ctx.meta
        B, _, H, W = x.shape
        M, Kw = w.shape[0], w.shape[1]
        HW = H * W
        g = _c(g)
        full = (col_off == 0 and K == Kw)
        gw = grad_like(w) if full else torch.zeros_like(w)
        pw_wgrad(g, 0, M * HW, x, 0, K * HW, gw, col_off, Kw, B, M, K, HW)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            pw_conv(g, 0, M * HW, w, col_off, 0, 1, Kw, gx, 0, K * HW, B, K, M, HW)
        return gx, gw, None, None


class DwConvFn(torch.autograd.Function):
    """depthwise 3x3 (zero pad), optionally followed by LeakyReLU(0.2)"""

    @staticmethod
    def forward(ctx, x, w, leaky):
        _check(x, w)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        dw3x3(x, w, None, C, y, B, C, H, W)
        if leaky:
            lib().call("cidnet_elementwise", 0, _p(y), None, _p(y), y.numel(), _stream())
        ctx.save_for_backward(x, w, y if leaky else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        B, C, H, W = x.shape
        g = _c(g)
        if y is not None:
            gp = torch.empty_like(g)
            lib().call("cidnet_elementwise", 1, _p(g), _p(y), _p(gp), g.numel(), _stream())
            g = gp
        gw = grad_like(w)
        dw3x3_wgrad(x, g, gw, None, C, B, C, H, W)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            dw3x3(g, w, None, C, gx, B, C, H, W, flip=True)
        return gx, gw, None


class UnaryFn(torch.autograd.Function):
    """kind 'leaky': LeakyReLU(0.2); kind 'sigmoid'"""

    @staticmethod
    def forward(ctx, x, kind):
        _check(x)
        x = _c(x)
        y = torch.empty_like(x)
        lib().call("cidnet_elementwise", 0 if kind == "leaky" else 2, _p(x), None, _p(y), x.numel(), _stream())
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _c(g)
        gx = torch.empty_like(y)
        lib().call("cidnet_elementwise", 1 if ctx.kind == "leaky" else 3, _p(g), _p(y), _p(gx), y.numel(), _stream())
        return gx, None


class Conv3x3Fn(torch.autograd.Function):
    """dense 3x3, zero pad 1, bias-free (noise_fusion[0], net/CIDNet_TNSM.py:96-98)"""

    @staticmethod
    def forward(ctx, x, w):
        _check(x, w)
        x = _c(x)
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        y = torch.empty((B, Co, H, W), device=x.device, dtype=torch.float32)
        conv3x3(x, w, y, B, Co, Ci, H, W, 9 * Ci, 9)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        g = _c(g)
        gw = grad_like(w)
        conv3x3_wgrad(g, x, gw, B, Co, Ci, H, W)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            conv3x3(g, w, gx, B, Ci, Co, H, W, 9, 9 * Ci, flip=True)
        return gx, gw


# --------------------------------------------------------------------------------------------
# K14: TNSM-specific ops
# --------------------------------------------------------------------------------------------
class NoiseMapFn(torch.autograd.Function):
    """DynamicNoiseMap.forward (net/TNSM.py:37-57) -> (B,1,H,W): global avg/max pool, the two tiny FCs
    and sigmoid, folded with noise_branch[2] and final_conv into one per-sample row vector applied to
    t = leaky(dw3x3(x))."""

    @staticmethod
    def forward(ctx, x, w_fc1, w_fc2, w_dw, w_pw, w_final):
        _check(x, w_fc1, w_fc2, w_dw, w_pw, w_final)
        x = _c(x)
        B, C, H, W = x.shape
        HW = H * W
        R = w_fc1.shape[0]
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        avg, mx = torch.empty((B, C), **f32), torch.empty((B, C), **f32)
        amax = torch.empty((B, C), device=dev, dtype=torch.int32)
        lib().call("cidnet_global_pool_fwd", _p(x), _p(avg), _p(mx), _p(amax), B, C, HW, _stream())
        hsum, gf, vrow = torch.empty((B, R, 3), **f32), torch.empty((B, C), **f32), torch.empty((B, C), **f32)
        lib().call("cidnet_noise_global_fwd", _p(avg), _p(mx), _p(w_fc1), _p(w_fc2), _p(w_pw), _p(w_final), _p(hsum), _p(gf),
                   _p(vrow), B, C, R, _stream())
        t = torch.empty_like(x)
        dw3x3(x, w_dw, None, C, t, B, C, H, W)
        lib().call("cidnet_elementwise", 0, _p(t), None, _p(t), t.numel(), _stream())
        nm = torch.empty((B, 1, H, W), **f32)
        lib().call("cidnet_rowdot_sigmoid_fwd", _p(t), _p(vrow), _p(nm), B, C, HW, _stream())
        ctx.save_for_backward(x, w_fc1, w_fc2, w_dw, w_pw, w_final, avg, mx, amax, hsum, gf, vrow, t, nm)
        return nm

    @staticmethod
    def backward(ctx, gnm):
        x, w_fc1, w_fc2, w_dw, w_pw, w_final, avg, mx, amax, hsum, gf, vrow, t, nm = ctx.saved_tensors
        B, C, H, W = x.shape
        HW = H * W
        R = w_fc1.shape[0]
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        gnm = _c(gnm)
        gt = torch.empty_like(t)
        gv = torch.empty((B, C), **f32)
        lib().call("cidnet_rowdot_sigmoid_bwd", _p(gnm), _p(nm), _p(t), _p(vrow), _p(gt), _p(gv), B, C, HW, _stream())
        # t = leaky(dw(x))
        lib().call("cidnet_elementwise", 1, _p(gt), _p(t), _p(gt), gt.numel(), _stream())
        g_dw = grad_like(w_dw)
        dw3x3_wgrad(x, gt, g_dw, None, C, B, C, H, W)
        gx = torch.empty_like(x)
        dw3x3(gt, w_dw, None, C, gx, B, C, H, W, flip=True)
        # global branch
        gW1_b, gW2_b = torch.empty((B, R, C), **f32), torch.empty((B, C, R), **f32)
        gWn_b, gwf_b = torch.empty((B, C, C), **f32), torch.empty((B, C), **f32)
        gavg, gmx = torch.empty((B, C), **f32), torch.empty((B, C), **f32)
        lib().call("cidnet_noise_global_bwd", _p(avg), _p(mx), _p(w_fc1), _p(w_fc2), _p(w_pw), _p(w_final), _p(hsum), _p(gf),
                   _p(gv), _p(gW1_b), _p(gW2_b), _p(gWn_b), _p(gwf_b), _p(gavg), _p(gmx), B, C, R, _stream())
        g_fc1, g_fc2, g_pw, g_final = grad_like(w_fc1), grad_like(w_fc2), grad_like(w_pw), grad_like(w_final)
        for part, out, n in ((gW1_b, g_fc1, R * C), (gW2_b, g_fc2, C * R), (gWn_b, g_pw, C * C), (gwf_b, g_final, C)):
            lib().call("cidnet_sum_rows", _p(part), B, n, _p(out), _stream())
        gpool = torch.empty_like(x)
        lib().call("cidnet_global_pool_bwd", _p(gavg), _p(gmx), _p(amax), _p(gpool), B, C, HW, _stream())
        lib().call("cidnet_add", _p(gx), _p(gpool), _p(gx), gx.numel(), _stream())
        return gx, g_fc1, g_fc2, g_dw, g_pw, g_final


class NoiseCABResidualFn(torch.autograd.Function):
    """out = x_res + NoiseAwareAttentionCABStyle(xn, yn, noise_map) (net/TNSM.py:83-128): CAB without the
    L2 normalisation, value modulated by sigmoid(conv1x1(noise_map))."""

    @staticmethod
    def forward(ctx, x_res, xn, yn, nm, temperature, wq, wq_dw, wkv, wkv_dw, w_scaler, wp, heads):
        _check(x_res, xn, yn, nm, temperature, wq, wq_dw, wkv, wkv_dw, w_scaler, wp)
        x_res, xn, yn, nm = _c(x_res), _c(xn), _c(yn), _c(nm)
        B, C, H, W = xn.shape
        HW = H * W
        dev = xn.device
        ch = C // heads
        f32 = dict(device=dev, dtype=torch.float32)
        qkv0 = torch.empty((B, 3 * C, H, W), **f32)
        pw_conv(xn, 0, C * HW, wq, 0, 0, C, 1, qkv0, 0, 3 * C * HW, B, C, C, HW)
        pw_conv(yn, 0, C * HW, wkv, 0, 0, C, 1, qkv0, C * HW, 3 * C * HW, B, 2 * C, C, HW)
        qkv = torch.empty_like(qkv0)
        dw3x3(qkv0, wq_dw, wkv_dw, C, qkv, B, 3 * C, H, W)
        attn = torch.empty((B, heads, ch, ch), **f32)
        shat = torch.empty_like(attn)
        nq, nk = torch.empty((B, C), **f32), torch.empty((B, C), **f32)
        M = torch.empty((B, C, C), **f32)
        n = _raw("cidnet_attn_gram_ws_floats", B, C, heads, HW)
        ws = _ws(n, dev)
        lib().call("cidnet_attn_fwd", _p(qkv), _p(temperature), _p(wp), _p(attn), _p(shat), _p(nq), _p(nk), _p(M), _p(ws),
                   ws.numel(), B, C, heads, HW, 0, _stream())
        vmod = torch.empty((B, C, H, W), **f32)
        lib().call("cidnet_modulate_fwd", _po(qkv, 2 * C * HW), 3 * C * HW, _p(nm), _p(w_scaler), _p(vmod), B, C, HW, _stream())
        out = torch.empty_like(x_res)
        pw_conv(vmod, 0, C * HW, M, 0, C * C, C, 1, out, 0, C * HW, B, C, C, HW, res=x_res, r_bs=C * HW)
        ctx.save_for_backward(xn, yn, nm, qkv0, qkv, vmod, attn, shat, nq, nk, M, temperature, wq, wq_dw, wkv, wkv_dw, w_scaler, wp)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        (xn, yn, nm, qkv0, qkv, vmod, attn, shat, nq, nk, M, temperature, wq, wq_dw, wkv, wkv_dw, w_scaler,
         wp) = ctx.saved_tensors
        heads = ctx.heads
        B, C, H, W = xn.shape
        HW = H * W
        dev = xn.device
        f32 = dict(device=dev, dtype=torch.float32)
        g = _c(g)
        dqkv = torch.empty_like(qkv)
        dvm = torch.empty_like(vmod)
        pw_conv(g, 0, C * HW, M, 0, C * C, 1, C, dvm, 0, C * HW, B, C, C, HW)
        dM = torch.empty_like(M)
        pw_wgrad(g, 0, C * HW, vmod, 0, C * HW, dM, 0, C, B, C, C, HW, per_sample=True)
        # v' = v * sigmoid(ws * nm)
        gnm = torch.empty_like(nm)
        nws = _raw("cidnet_modulate_bwd_ws_floats", B, C, HW)
        part = torch.empty(nws, **f32)
        lib().call("cidnet_modulate_bwd", _po(qkv, 2 * C * HW), 3 * C * HW, _p(nm), _p(w_scaler), _p(dvm), _po(dqkv, 2 * C * HW),
                   3 * C * HW, _p(gnm), _p(part), B, C, HW, _stream())
        g_ws = grad_like(w_scaler)
        lib().call("cidnet_sum_rows", _p(part), nws // C, C, _p(g_ws), _stream())
        dwp_b = torch.empty_like(M)
        dT_b = torch.empty((B, heads), **f32)
        wqk = torch.empty((B, 2 * C, 2 * C), **f32)
        lib().call("cidnet_attn_bwd", _p(dM), _p(wp), _p(attn), _p(shat), _p(nq), _p(nk), _p(temperature), _p(dwp_b), _p(dT_b),
                   _p(wqk), B, C, heads, 0, _stream())
        g_wp = grad_like(wp)
        lib().call("cidnet_sum_rows", _p(dwp_b), B, C * C, _p(g_wp), _stream())
        g_T = grad_like(temperature)
        lib().call("cidnet_sum_rows", _p(dT_b), B, heads, _p(g_T), _stream())
        pw_conv(qkv, 0, 3 * C * HW, wqk, 0, 4 * C * C, 2 * C, 1, dqkv, 0, 3 * C * HW, B, 2 * C, 2 * C, HW)
        g_wq_dw, g_wkv_dw = grad_like(wq_dw), grad_like(wkv_dw)
        dw3x3_wgrad(qkv0, dqkv, g_wq_dw, g_wkv_dw, C, B, 3 * C, H, W)
        dqkv0 = torch.empty_like(qkv0)
        dw3x3(dqkv, wq_dw, wkv_dw, C, dqkv0, B, 3 * C, H, W, flip=True)
        g_wq, g_wkv = grad_like(wq), grad_like(wkv)
        pw_wgrad(dqkv0, 0, 3 * C * HW, xn, 0, C * HW, g_wq, 0, C, B, C, C, HW)
        pw_wgrad(dqkv0, C * HW, 3 * C * HW, yn, 0, C * HW, g_wkv, 0, C, B, 2 * C, C, HW)
        dxn = torch.empty_like(xn)
        pw_conv(dqkv0, 0, 3 * C * HW, wq, 0, 0, 1, C, dxn, 0, C * HW, B, C, C, HW)
        dyn = torch.empty_like(yn)
        pw_conv(dqkv0, C * HW, 3 * C * HW, wkv, 0, 0, 1, C, dyn, 0, C * HW, B, C, 2 * C, HW)
        return g, dxn, dyn, gnm, g_T, g_wq, g_wq_dw, g_wkv, g_wkv_dw, g_ws, g_wp, None


class BlendFn(torch.autograd.Function):
    """nm * a + (1 - nm) * d with nm (B,1,H,W) broadcast over channels (net/TNSM.py:165-166)."""

    @staticmethod
    def forward(ctx, a, d, nm):
        _check(a, d, nm)
        a, d, nm = _c(a), _c(d), _c(nm)
        B, C, H, W = a.shape
        out = torch.empty_like(a)
        lib().call("cidnet_blend_fwd", _p(a), _p(d), _p(nm), _p(out), B, C, H * W, _stream())
        ctx.save_for_backward(a, d, nm)
        return out

    @staticmethod
    def backward(ctx, g):
        a, d, nm = ctx.saved_tensors
        B, C, H, W = a.shape
        g = _c(g)
        ga, gd, gnm = torch.empty_like(a), torch.empty_like(d), torch.empty_like(nm)
        lib().call("cidnet_blend_bwd", _p(a), _p(d), _p(nm), _p(g), _p(ga), _p(gd), _p(gnm), B, C, H * W, _stream())
        return ga, gd, gnm


class ResizeCatFn(torch.autograd.Function):
    """cat([F.interpolate(m, (H,W), 'bilinear', align_corners=False) for m in maps], 1)
    (net/CIDNet_TNSM.py:252-262); each map is resized straight into its channel slice."""

    @staticmethod
    def forward(ctx, H, W, *maps):
        _check(*maps)
        maps = [_c(m) for m in maps]
        B = maps[0].shape[0]
        ctot = sum(m.shape[1] for m in maps)
        out = torch.empty((B, ctot, H, W), device=maps[0].device, dtype=torch.float32)
        off = 0
        for m in maps:
            c, hi, wi = m.shape[1:]
            lib().call("cidnet_resize_bilinear_fwd", _p(m), _po(out, off * H * W), ctot * H * W, B, c, hi, wi, H, W, _stream())
            off += c
        ctx.shapes = [tuple(m.shape) for m in maps]
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.hw
        g = _c(g)
        ctot = g.shape[1]
        outs, off = [], 0
        for shp in ctx.shapes:
            B, c, hi, wi = shp
            gm = torch.empty(shp, device=g.device, dtype=torch.float32)
            lib().call("cidnet_resize_bilinear_bwd", _po(g, off * H * W), ctot * H * W, _p(gm), B, c, hi, wi, H, W, _stream())
            outs.append(gm)
            off += c
        return (None, None, *outs)
