"""fp32-MFMA 1x1 conv (pw.hip) vs the bf16x3 split-product kernel (pwx.hip) at the step's shapes: time and error against
fp64 (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
shapes = [(144, 766, 3750), (766, 144, 3750), (288, 288, 3750), (144, 383, 3750), (383, 144, 3750), (144, 144, 3750), (144, 288, 3750),
          (288, 144, 3750), (72, 382, 15000), (72, 191, 15000), (382, 72, 15000), (144, 144, 15000), (72, 72, 15000), (72, 144, 15000),
          (144, 72, 15000), (191, 72, 15000), (72, 72, 60000), (36, 190, 60000), (190, 36, 60000), (36, 36, 60000), (36, 95, 60000),
          (95, 36, 60000), (72, 36, 60000), (36, 72, 60000), (36, 36, 240000)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
B = 8
torch.manual_seed(0)
for M, K, HW in shapes:
    x = torch.randn(B, K, HW, device=dev); w = torch.randn(M, K, device=dev) / K ** 0.5
    r = torch.randn(B, M, HW, device=dev)
    y0 = torch.empty(B, M, HW, device=dev); y1 = torch.empty_like(y0)
    ops.PW_BF16X3["on"] = False
    f0 = lambda: ops.pw_conv(x, 0, K * HW, w, 0, 0, K, 1, y0, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW)
    f1 = lambda: ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, 0, K, 1, y1, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW)
    t0, t1 = timeit(f0), timeit(f1)
    ref = torch.einsum("mk,bkp->bmp", w.double(), x[:2].double()) + r[:2].double()
    e0 = (y0[:2].double() - ref).abs().max().item(); e1 = (y1[:2].double() - ref).abs().max().item()
    gf = 2.0 * M * K * HW * B / 1e9
    by = (M + K + M) * 4.0 * HW * B
    print(f"M={M:4d} K={K:4d} HW={HW:6d}: fp32 {t0:7.1f} us ({gf / t0 * 1e3:6.1f} TF/s, {by / t0 / 1e3:5.0f} GB/s)   bf16x3 {t1:7.1f} us "
          f"({gf / t1 * 1e3:6.1f} TF/s, {by / t1 / 1e3:5.0f} GB/s)   err vs fp64: {e0:.2e} / {e1:.2e}")
