"""bf16x3 split-product 3x3 conv (csrc/conv3x.hip) vs the fp32-MFMA kernel: error of both against fp64, and time (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.ops import _p, _stream, lib

dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n

SHAPES = [(2, 36, 36, 37, 51, 0), (2, 72, 36, 20, 33, 0), (8, 36, 36, 400, 600, 0), (8, 36, 36, 400, 600, 1), (8, 36, 36, 200, 300, 0),
          (8, 72, 36, 200, 300, 0), (8, 36, 72, 200, 300, 1), (8, 144, 72, 100, 150, 0), (8, 72, 144, 100, 150, 1), (8, 72, 144, 50, 75, 0),
          (8, 144, 72, 50, 75, 1), (8, 36, 72, 100, 150, 0), (8, 72, 36, 100, 150, 1)]
for (B, M, K, H, W, flip) in SHAPES:
    x = torch.randn(B, K, H, W, device=dev)
    w = torch.randn(M, K, 3, 3, device=dev) / (3 * K ** 0.5)
    if flip:      # data-gradient form: A[m][k][tap] = Wt[k][m][8 - tap]: Wt has shape (K_out_of_fwd = K here as 'k', M ...)
        wt = torch.randn(K, M, 3, 3, device=dev) / (3 * K ** 0.5)      # forward weight (Cout = K, Cin = M)
        w_ms, w_ks = 9, 9 * M
        ref64 = F.conv_transpose2d(x.double().cpu(), wt.double().cpu(), padding=1)
        wa = wt
    else:
        w_ms, w_ks = 9 * K, 9
        ref64 = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
        wa = w
    y32 = torch.empty(B, M, H, W, device=dev); ys = torch.empty_like(y32)
    f32 = lambda: ops.conv3x3(x, wa, y32, B, M, K, H, W, w_ms, w_ks, flip=bool(flip))
    nws = ops._raw("cidnet_conv3x3_bf16x3_ws_floats", M, K)
    wsb = torch.empty(nws, device=dev)
    fs = lambda: lib().call("cidnet_conv3x3_bf16x3", _p(x), K * H * W, _p(wa), w_ms, w_ks, flip, None, 0, _p(ys), M * H * W, _p(wsb), nws, B, M, K, H, W, _stream())
    f32(); fs(); torch.cuda.synchronize()
    e32 = (y32.cpu().double() - ref64).abs().max().item(); es = (ys.cpu().double() - ref64).abs().max().item()
    t32, ts = timeit(f32), timeit(fs)
    fl = 2.0 * 9 * M * K * H * W * B
    print(f"{(B, M, K, H, W)} flip={flip}: max err vs fp64: fp32-MFMA {e32:.2e}, bf16x3 {es:.2e} (|y| max {ref64.abs().max().item():.2f}) | "
          f"fp32 {t32:8.1f} us ({fl / t32 / 1e6:6.1f} TF/s)  bf16x3 {ts:8.1f} us ({fl / ts / 1e6:6.1f} TF/s)", flush=True)
