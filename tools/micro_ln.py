"""Times the channels-first LayerNorm forward / backward at the step's shapes (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.ops import _p, _stream, lib, _ws, _raw
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
for C, HW in ((36, 60000), (72, 15000), (144, 3750), (36, 240000)):
    x = torch.randn(B, C, HW, device=dev); w = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    y = torch.empty_like(x); mean = torch.empty(B, HW, device=dev); rstd = torch.empty(B, HW, device=dev)
    gy = torch.randn_like(x); gres = torch.randn_like(x); gx = torch.empty_like(x); gw = torch.zeros(C, device=dev); gb = torch.zeros(C, device=dev)
    tf = timeit(lambda: lib().call("cidnet_ln_cf_fwd", _p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), B, C, HW, 1e-5, _stream()))
    n = _raw("cidnet_ln_cf_bwd_ws_floats", C); ws = _ws(n, dev)
    tb = timeit(lambda: lib().call("cidnet_ln_cf_bwd_res", _p(x), _p(w), _p(gy), _p(mean), _p(rstd), _p(gres), _p(gx), _p(gw), _p(gb), 0, _p(ws), ws.numel(), B, C, HW, _stream()))
    by = x.numel() * 4.0
    print(f"LN C={C:3d} HW={HW:6d}: fwd {tf:6.1f} us ({2 * by / tf / 1e3:5.0f} GB/s)   bwd+res {tb:6.1f} us ({4 * by / tb / 1e3:5.0f} GB/s)")
