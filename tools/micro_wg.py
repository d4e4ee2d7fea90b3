"""Sweep the pixels-per-block of cidnet_pw_wgrad over the step's shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib
dev = torch.device("cuda:0")

def run(B, M, N, HW, flags, iters=20):
    dy = torch.rand(B, M, HW, device=dev); x = torch.rand(B, N, HW, device=dev); dw = torch.empty(M, N, device=dev)
    lib().raw("cidnet_debug_pw_flags")(flags)
    for _ in range(3): ops.pw_wgrad(dy, 0, M * HW, x, 0, N * HW, dw, 0, N, B, M, N, HW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.pw_wgrad(dy, 0, M * HW, x, 0, N * HW, dw, 0, N, B, M, N, HW)
    e1.record(); torch.cuda.synchronize()
    lib().raw("cidnet_debug_pw_flags")(0)
    return e0.elapsed_time(e1) * 1e3 / iters

SHAPES = [(36, 36, 60000, 12), (190, 36, 60000, 4), (36, 36, 240000, 2), (382, 72, 15000, 3), (766, 144, 3750, 4), (72, 72, 15000, 8),
          (36, 95, 60000, 4), (72, 36, 60000, 4), (144, 144, 3750, 8), (144, 383, 3750, 4), (72, 191, 15000, 3), (144, 72, 15000, 3), (288, 144, 3750, 4)]
td = tb = 0.0
for M, N, HW, n in SHAPES:
    run(8, M, N, HW, 0)
    d = run(8, M, N, HW, 0)
    res = {p: run(8, M, N, HW, 128 | ((p // 128) << 8)) for p in (512, 1024, 2048, 4096)}
    best = min(res, key=res.get)
    td += d * n; tb += min(d, res[best]) * n
    print(f"M={M:4d} N={N:4d} HW={HW:6d} x{n:2d}: auto {d:6.1f} us | " + " ".join(f"{k}:{v:.0f}" for k, v in res.items()), flush=True)
print(f"per step: auto {td / 1e3:.2f} ms, best-of-sweep {tb / 1e3:.2f} ms")
