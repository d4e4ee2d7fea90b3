"""Two LayerNorm modules on one tensor: one dual pass against two single passes, forward and backward (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.ops import _p, _stream, lib, _raw
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, C, H, W in [(8, 36, 200, 300), (8, 72, 100, 150), (8, 144, 50, 75)]:
    HW = H * W
    x = torch.randn(B, C, H, W, device=dev); g1 = torch.randn_like(x); g2 = torch.randn_like(x); ad = torch.randn_like(x)
    w1, b1, w2, b2 = (torch.randn(C, device=dev) for _ in range(4))
    y1, y2, gx, gxa = (torch.empty_like(x) for _ in range(4))
    mean = torch.empty(B, H, W, device=dev); rstd = torch.empty_like(mean)
    gw1, gb1, gw2, gb2 = (torch.empty(C, device=dev) for _ in range(4))
    ws = torch.empty(max(_raw("cidnet_ln_cf_bwd_ws_floats", C), _raw("cidnet_ln_cf_bwd2_ws_floats", C)), device=dev)
    f1 = lambda: (lib().call("cidnet_ln_cf_fwd", _p(x), _p(w1), _p(b1), _p(y1), _p(mean), _p(rstd), B, C, HW, 1e-6, _stream()),
                  lib().call("cidnet_ln_cf_fwd", _p(x), _p(w2), _p(b2), _p(y2), _p(mean), _p(rstd), B, C, HW, 1e-6, _stream()))
    f2 = lambda: lib().call("cidnet_ln_cf_fwd2", _p(x), _p(w1), _p(b1), _p(y1), _p(w2), _p(b2), _p(y2), _p(mean), _p(rstd), B, C, HW, 1e-6, _stream())
    b1_ = lambda: (lib().call("cidnet_ln_cf_bwd_res", _p(x), _p(w1), _p(g1), _p(mean), _p(rstd), _p(ad), _p(gxa), _p(gw1), _p(gb1), 0, _p(ws), ws.numel(), B, C, HW, _stream()),
                   lib().call("cidnet_ln_cf_bwd_res", _p(x), _p(w2), _p(g2), _p(mean), _p(rstd), _p(gxa), _p(gx), _p(gw2), _p(gb2), 0, _p(ws), ws.numel(), B, C, HW, _stream()))
    b2_ = lambda: lib().call("cidnet_ln_cf_bwd2", _p(x), _p(w1), _p(g1), _p(w2), _p(g2), _p(mean), _p(rstd), _p(ad), _p(gx), _p(gw1), _p(gb1), 0, _p(gw2), _p(gb2), 0, _p(ws), ws.numel(), B, C, HW, _stream())
    byt = x.numel() * 4 / 1e3
    a, b, c, d = timeit(f1), timeit(f2), timeit(b1_), timeit(b2_)
    print(f"C={C:3d} {H}x{W}: forward 2 passes {a:6.1f} us ({4 * byt / a:5.0f} GB/s)  dual {b:6.1f} us ({3 * byt / b:5.0f} GB/s) | "
          f"backward 2 passes {c:6.1f} us ({8 * byt / c:5.0f} GB/s)  dual {d:6.1f} us ({5 * byt / d:5.0f} GB/s)", flush=True)
