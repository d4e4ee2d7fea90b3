"""Micro-benchmark of the thin (<= 4 channel) 3x3 convolution paths vs the MFMA path (dev tool)."""
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

B, H, W = 8, 400, 600
for (Co, Ci) in [(36, 1), (36, 3), (1, 36), (2, 36)]:
    x = torch.randn(B, Ci, H, W, device=dev); w = torch.randn(Co, Ci, 3, 3, device=dev)
    y = torch.empty(B, Co, H, W, device=dev); gy = torch.randn_like(y); dx = torch.empty_like(x); gw = torch.empty_like(w)
    mb = (Ci + Co) * B * H * W * 4 / 1e6
    for flag, tag in ((0, "thin"), (8, "mfma")):
        lib().raw("cidnet_debug_c3_flags")(flag)
        t_f = timeit(lambda: ops.conv3x3(x, w, y, B, Co, Ci, H, W, 9 * Ci, 9, replicate=True))
        t_d = timeit(lambda: ops.conv3x3(gy, w, dx, B, Ci, Co, H, W, 9, 9 * Ci, flip=True))
        t_w = timeit(lambda: ops.conv3x3_wgrad(gy, x, gw, B, Co, Ci, H, W, replicate=True))
        print(f"{Ci:2d}->{Co:2d} {tag}: fwd {t_f:6.0f} us ({mb / t_f:5.2f} TB/s)  dgrad {t_d:6.0f} us ({mb / t_d:5.2f} TB/s)  wgrad {t_w:6.0f} us ({mb / t_w:5.2f} TB/s)", flush=True)
    lib().raw("cidnet_debug_c3_flags")(0)
