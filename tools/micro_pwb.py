"""Backward of a 1x1 conv: the fused kernel (csrc/pwb.hip) against data-gradient + weight-gradient launches (dev tool)."""
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
B = 8
for M, N, HW in [(190, 36, 60000), (36, 95, 60000), (36, 36, 60000), (72, 36, 60000)]:
    gy = torch.randn(B, M, HW, device=dev); x = torch.randn(B, N, HW, device=dev); w = torch.randn(M, N, device=dev) * 0.1
    gx, dw, gx2, dw2 = torch.empty(B, N, HW, device=dev), torch.empty(M, N, device=dev), torch.empty(B, N, HW, device=dev), torch.empty(M, N, device=dev)
    t_f = timeit(lambda: ops.pw_bwd_fused(gy, x, w, gx, dw, B, M, N, HW))
    t_d = timeit(lambda: ops.pw_conv(gy, 0, M * HW, w, 0, 0, 1, N, gx2, 0, N * HW, B, N, M, HW))
    t_w = timeit(lambda: ops.pw_wgrad(gy, 0, M * HW, x, 0, N * HW, dw2, 0, N, B, M, N, HW))
    by = (M + 2 * N) * 4.0 * HW * B
    print(f"M={M:3d} N={N:3d} HW={HW}: fused {t_f:6.1f} us ({by / t_f / 1e3:5.0f} GB/s)   separate {t_d:6.1f} + {t_w:6.1f} = {t_d + t_w:6.1f} us   "
          f"max |gx diff| {(gx - gx2).abs().max().item():.2e}  max |dw diff| / max |dw| {((dw - dw2).abs().max() / dw2.abs().max()).item():.2e}", flush=True)
