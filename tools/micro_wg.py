"""1x1 weight gradient (csrc/pw.hip, pw_wgrad_kernel): fp32 MFMA against the split-product bf16 path over the step's shapes:
time, algorithmic GB/s and error against fp64 (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
if os.environ.get("WG_DBG_FLAGS"):            # -DCIDNET_DEBUG library only (32 = the previous block order)
    import ctypes
    from hvi_cidnet_amd._lib import lib
    f = lib().raw("cidnet_debug_pw_flags"); f.argtypes = [ctypes.c_int]; f.restype = None
    f(int(os.environ["WG_DBG_FLAGS"]))


def run(dy, x, dw, B, M, N, HW, bf3, iters=20):
    ops.PW_WGRAD_BF16X3["on"] = bf3
    f = lambda: ops.pw_wgrad(dy, 0, M * HW, x, 0, N * HW, dw, 0, N, B, M, N, HW)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [(36, 36, 60000, 12), (190, 36, 60000, 4), (36, 36, 240000, 2), (382, 72, 15000, 3), (766, 144, 3750, 4), (72, 72, 15000, 8),
          (36, 95, 60000, 4), (72, 36, 60000, 4), (144, 144, 3750, 8), (144, 383, 3750, 4), (72, 191, 15000, 3), (144, 72, 15000, 3), (288, 144, 3750, 4)]
t0 = t1 = 0.0
for M, N, HW, n in SHAPES:
    B = 8
    dy = torch.randn(B, M, HW, device=dev); x = torch.randn(B, N, HW, device=dev)
    d0, d1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    a = run(dy, x, d0, B, M, N, HW, False)
    b = run(dy, x, d1, B, M, N, HW, True)
    ref = torch.einsum("bmp,bnp->mn", dy[:, :, :4096].double(), x[:, :, :4096].double()) if HW > 4096 else torch.einsum("bmp,bnp->mn", dy.double(), x.double())
    if HW > 4096:
        e0 = e1 = float("nan")          # (error is checked on the small planes; the large ones would need minutes of fp64)
    else:
        e0, e1 = (d0.double() - ref).abs().max().item(), (d1.double() - ref).abs().max().item()
    by = (M + N) * 4.0 * HW * B
    t0 += a * n; t1 += b * n
    print(f"M={M:4d} N={N:4d} HW={HW:6d} x{n:2d}: fp32 {a:6.1f} us ({by / a / 1e3:5.0f} GB/s)  bf16x3 {b:6.1f} us ({by / b / 1e3:5.0f} GB/s)  err vs fp64 {e0:.2e} / {e1:.2e}", flush=True)
print(f"per step: fp32 {t0 / 1e3:.2f} ms, bf16x3 {t1 / 1e3:.2f} ms")
