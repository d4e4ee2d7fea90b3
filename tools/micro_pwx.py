"""The bf16x3 split-product 1x1 conv (pwx.hip) at the step's shapes: time per call (a 1 GiB buffer is rewritten between
calls: cold caches, as inside the step; the rewrite's own time is subtracted) and error against fp64.  With --fp32 the
fp32-MFMA kernel (pw.hip) is timed next to it.  A/B of library builds: CIDNET_LIB_PATH=tools/bin/<lib> (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
flush = torch.empty(256 << 20, device=dev)
def timeit(f, n=10):
    for _ in range(2): f()
    def run(g):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            flush.add_(1.0); g()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    return run(f) - run(lambda: None)
# (M, K, HW, residual) as launched by one training step (bench.py --op-table)
shapes = [(36, 95, 60000, 1), (72, 72, 60000, 0), (72, 36, 60000, 0), (36, 190, 60000, 0), (36, 72, 60000, 0), (95, 36, 60000, 0),
          (72, 72, 15000, 0), (72, 72, 15000, 1), (144, 72, 15000, 0), (72, 144, 15000, 0), (144, 144, 15000, 0), (72, 382, 15000, 0),
          (72, 191, 15000, 1), (191, 72, 15000, 0), (382, 72, 15000, 0),
          (144, 766, 3750, 0), (766, 144, 3750, 0), (288, 288, 3750, 0), (144, 383, 3750, 1), (383, 144, 3750, 0), (288, 144, 3750, 0),
          (144, 288, 3750, 0), (144, 144, 3750, 1)]
B = 8
torch.manual_seed(0)
tot = 0.0
for M, K, HW, res in shapes:
    x = torch.randn(B, K, HW, device=dev); w = torch.randn(M, K, device=dev) / K ** 0.5
    r = torch.randn(B, M, HW, device=dev) if res else None
    y1 = torch.empty(B, M, HW, device=dev)
    f1 = lambda: ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, 0, K, 1, y1, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW)
    t1 = timeit(f1)
    ref = torch.einsum("mk,bkp->bmp", w.double(), x[:2].double()) + (r[:2].double() if res else 0)
    e1 = (y1[:2].double() - ref).abs().max().item()
    by = (M + K + (M if res else 0)) * 4.0 * HW * B
    line = f"M={M:4d} K={K:4d} HW={HW:6d} res={res}: bf16x3 {t1:7.1f} us ({by / t1 / 1e3:5.0f} GB/s) err {e1:.2e} wins={int(ops.pw_bf16x3_wins(M, K, HW))}"
    if "--fp32" in sys.argv:
        y0 = torch.empty_like(y1)
        ops.PW_BF16X3["on"] = False
        t0 = timeit(lambda: ops.pw_conv(x, 0, K * HW, w, 0, 0, K, 1, y0, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW))
        line += f"   fp32 {t0:7.1f} us ({by / t0 / 1e3:5.0f} GB/s)"
    tot += t1
    print(line, flush=True)
print(f"sum bf16x3 {tot:.1f} us")
