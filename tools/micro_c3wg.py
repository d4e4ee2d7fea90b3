"""Times the dense 3x3 weight gradient at the step's shapes and checks it against fp64 (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = [(36, 36, 400, 600), (144, 72, 100, 150), (72, 36, 200, 300), (36, 36, 200, 300), (36, 72, 100, 150), (72, 144, 50, 75)]
B = 8
tot = tot32 = gf_tot = 0.0
for M, N, H, W in SHAPES:
    dy = torch.randn(B, M, H, W, device=dev); x = torch.randn(B, N, H, W, device=dev); dw = torch.empty(M, N, 3, 3, device=dev)
    ops.CONV3_WGRAD_BF16X3["on"] = False
    us32 = timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, B, M, N, H, W))
    ops.CONV3_WGRAD_BF16X3["on"] = True
    us = timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, B, M, N, H, W))
    ref = torch.nn.grad.conv2d_weight(x[:1].double().cpu(), (M, N, 3, 3), dy[:1].double().cpu(), padding=1)
    dw1 = torch.empty(M, N, 3, 3, device=dev); ops.conv3x3_wgrad(dy[:1].contiguous(), x[:1].contiguous(), dw1, 1, M, N, H, W)
    err = (dw1.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    gf = 18.0 * M * N * H * W * B / 1e9
    tot += 2 * us; tot32 += 2 * us32; gf_tot += 2 * gf
    print(f"wgrad {M:3d}x{N:3d} @ {H}x{W}: {us:7.1f} us (fp32 MFMA {us32:7.1f})  {gf / us * 1e3:6.1f} TFLOP/s   rel err vs fp64 {err:.1e}")
print(f"family per step (each shape twice): {tot / 1e3:.3f} ms (fp32 MFMA {tot32 / 1e3:.3f}), {gf_tot / tot * 1e3:.1f} TFLOP/s")
