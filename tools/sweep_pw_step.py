"""Every 1x1-conv shape of one training step (pw_conv call list of bench.py --op-table), fp32-MFMA kernel (pw.hip) against
the bf16x3 kernel (pwx.hip), cold caches: the data behind ops.pw_bf16x3_wins (dev tool)."""
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
# (M, K, HW, residual, per_sample, dgrad-strides)
S = []
for C, HW in ((36, 60000), (72, 15000), (144, 3750)):
    h = int(C * 2.66)
    S += [(C, C, HW, 0, 0, 0), (2 * C, C, HW, 0, 0, 0), (C, C, HW, 1, 1, 0), (2 * h, C, HW, 0, 0, 0), (C, h, HW, 1, 0, 0), (C, h, HW, 0, 0, 0),
          (C, C, HW, 0, 1, 1), (2 * C, 2 * C, HW, 0, 1, 0), (C, C, HW, 0, 0, 1), (C, 2 * C, HW, 0, 0, 1), (h, C, HW, 0, 0, 1), (C, 2 * h, HW, 0, 0, 1)]
S += [(36, 36, 240000, 0, 0, 1), (72, 72, 3750, 0, 0, 0), (36, 36, 15000, 0, 0, 0)]
B = 8
torch.manual_seed(0)
x = torch.randn(B, 36, 1000, device=dev); w = torch.randn(36, 36, device=dev); y = torch.empty_like(x)
timeit(lambda: ops.pw_conv_bf16x3(x, 0, 36000, w, 0, 0, 36, 1, y, 0, 36000, B, 36, 36, 1000))     # first-use effects
for M, K, HW, res, ps, dg in S:
    if not ops._raw("cidnet_pw_conv_bf16x3_supported", M, K, HW): continue
    x = torch.randn(B, K, HW, device=dev)
    w = torch.randn((B, M, K) if ps else ((K, M) if dg else (M, K)), device=dev) / K ** 0.5
    w_bs, w_ms, w_ks = (M * K if ps else 0), (1 if dg else K), (M if dg else 1)
    if ps and dg: w_ms, w_ks = 1, M
    r = torch.randn(B, M, HW, device=dev) if res else None
    y0 = torch.empty(B, M, HW, device=dev); y1 = torch.empty_like(y0)
    old = ops.PW_BF16X3["on"]; ops.PW_BF16X3["on"] = False
    t0 = timeit(lambda: ops.pw_conv(x, 0, K * HW, w, 0, w_bs, w_ms, w_ks, y0, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW))
    ops.PW_BF16X3["on"] = old
    t1 = timeit(lambda: ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, w_bs, w_ms, w_ks, y1, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW))
    by = (M + K + (M if res else 0)) * 4.0 * HW * B
    d = (y0 - y1).abs().max().item()
    print(f"M={M:4d} K={K:4d} HW={HW:6d} res={res} ps={ps} dg={dg}: fp32 {t0:7.1f} us ({by / t0 / 1e3:5.0f} GB/s)  bf16x3 {t1:7.1f} us ({by / t1 / 1e3:5.0f} GB/s) "
          f"ratio {t0 / t1:5.2f} rule={int(ops.pw_bf16x3_wins(M, K, HW))} |d|={d:.1e}", flush=True)
