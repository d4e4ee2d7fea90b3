"""Micro-benchmark of the depthwise / IEL-gate kernels over strip heights (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib

dev = torch.device("cuda:0")

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def cases(B, h, H, W):
    u = torch.randn(B, 2 * h, H, W, device=dev); pin = torch.randn_like(u); du = torch.randn_like(u); out = torch.empty_like(u)
    w = torch.randn(2 * h, 1, 3, 3, device=dev); w1 = torch.randn(h, 1, 3, 3, device=dev); w2 = torch.randn(h, 1, 3, 3, device=dev)
    gate = torch.empty(B, h, H, W, device=dev); dg = torch.randn_like(gate)
    gw = torch.empty_like(w); g1 = torch.empty_like(w1); g2 = torch.empty_like(w2)
    n = lib().raw("cidnet_iel_gate_dw_bwd_ws_floats")(B, h, H, W)
    ws = torch.empty(max(n, 1 << 22), device=dev)
    px = B * h * H * W * 4
    return {
        "dw3x3 fwd (2h)": (lambda: ops.dw3x3(pin, w, None, 2 * h, u, B, 2 * h, H, W), 4 * px),
        "dw+gate fwd": (lambda: lib().call("cidnet_iel_dw_gate_fwd", ops._p(pin), ops._p(w), ops._p(w1), ops._p(w2), ops._p(u), ops._p(gate),
                                           B, h, H, W, ops._stream()), 5 * px),
        "gate fwd": (lambda: lib().call("cidnet_iel_gate_fwd", ops._p(u), ops._p(w1), ops._p(w2), ops._p(gate), B, h, H, W, ops._stream()), 3 * px),
        "gate+dw bwd": (lambda: lib().call("cidnet_iel_gate_dw_bwd", ops._p(u), ops._p(w1), ops._p(w2), ops._p(dg), ops._p(out), ops._p(g1),
                                           ops._p(g2), ops._p(ws), ws.numel(), B, h, H, W, ops._stream()), 5 * px),
        "dw3x3 bwd (2h)": (lambda: ops.dw3x3_bwd(pin, du, w, None, 2 * h, out, gw, None, B, 2 * h, H, W), 6 * px),
    }

try:
    _dbg = lib().raw("cidnet_debug_dw_rows")          # only in -DCIDNET_DEBUG builds
except AttributeError:
    _dbg = lambda r: None
rows_list = ([int(a) for a in sys.argv[1:]] or [0, 4, 6, 8, 10, 12, 16, 20, 24]) if _dbg.__class__.__name__ != "function" else [0]
for sh in [(8, 95, 200, 300), (8, 191, 100, 150), (8, 383, 50, 75)]:
    cs = cases(*sh)
    for name, (fn, nbytes) in cs.items():
        res = []
        for r in rows_list:
            _dbg(r)
            us = timeit(fn)
            res.append(f"{r}:{us:6.0f}us/{nbytes / us / 1e6:4.2f}TB/s")
        _dbg(0)
        print(f"{sh} {name:16s} " + "  ".join(res), flush=True)
