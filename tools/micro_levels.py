"""dev tool: the bf16-mode (one-level) forms of the dense 3x3 conv, its weight gradient and the 1x1 conv against the
fp32-exact (three-level) forms on the step's shapes: microseconds per launch, GB/s of compulsory bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.ops import _p, _pe, _dt, _stream, lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
flush = torch.empty(256 << 20, device=dev, dtype=torch.float32)

def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return 1e3 * tot / n

print("== dense 3x3 (B, M, K, H, W): levels (3,3) | (1,1)")
for (B, M, K, H, W) in [(8, 36, 36, 400, 600), (8, 36, 36, 200, 300), (8, 72, 36, 200, 300), (8, 36, 72, 200, 300), (8, 144, 72, 100, 150), (8, 72, 144, 100, 150), (8, 72, 144, 50, 75)]:
    x = torch.randn(B, K, H, W, device=dev); w = torch.randn(M, K, 3, 3, device=dev) / (3 * K ** 0.5); y = torch.empty(B, M, H, W, device=dev)
    n = ops._raw("cidnet_conv3x3_bf16x3_ws_floats", M, K); ws = torch.empty(n, device=dev)
    lib().call("cidnet_conv3x3_bf16x3_prep", _p(w), 9 * K, 9, 0, _p(ws), n, M, K, _stream())
    t = {}
    for lv in (3, 1):
        t[lv] = timeit(lambda: lib().call("cidnet_conv3x3_bf16x3_pre_lv", _p(x), K * H * W, _p(ws), None, 0, _p(y), M * H * W, B, M, K, H, W, lv, lv, _stream()))
    by = (K + M) * 4.0 * H * W * B
    print(f"  {(B, M, K, H, W)}: {t[3]:7.1f} us ({by / t[3] / 1e3:6.0f} GB/s) | {t[1]:7.1f} us ({by / t[1] / 1e3:6.0f} GB/s)", flush=True)

print("== 3x3 weight gradient (B, M, N, H, W): levels 3 | 1")
for (B, M, N, H, W) in [(8, 36, 36, 400, 600), (8, 72, 36, 200, 300), (8, 144, 72, 100, 150), (8, 72, 144, 50, 75)]:
    x = torch.randn(B, N, H, W, device=dev); dy = torch.randn(B, M, H, W, device=dev); dw = torch.empty(M, N, 3, 3, device=dev)
    n = ops._raw("cidnet_conv3x3_wgrad_bf16x3_ws_floats", B, M, N, H, W); ws = torch.empty(n, device=dev)
    t = {}
    for lv in (3, 1):
        t[lv] = timeit(lambda: lib().call("cidnet_conv3x3_wgrad_bf16x3_lv", _p(dy), M * H * W, _p(x), N * H * W, _p(dw), _p(ws), n, B, M, N, H, W, lv, _stream()))
    by = (N + M) * 4.0 * H * W * B
    print(f"  {(B, M, N, H, W)}: {t[3]:7.1f} us ({by / t[3] / 1e3:6.0f} GB/s) | {t[1]:7.1f} us ({by / t[1] / 1e3:6.0f} GB/s)", flush=True)

print("== 1x1 (B, M, K, HW): fp32 x,y levels (3,3) | (1,1) | bf16 x | bf16 y | bf16 x,y     [GB/s of the bytes that form moves]")
for (B, M, K, HW) in [(8, 36, 36, 60000), (8, 190, 36, 60000), (8, 36, 95, 60000), (8, 36, 190, 60000), (8, 95, 36, 60000), (8, 72, 72, 15000), (8, 382, 72, 15000),
                      (8, 72, 191, 15000), (8, 144, 144, 3750), (8, 766, 144, 3750), (8, 144, 383, 3750)]:
    x = torch.randn(B, K, HW, device=dev); w = torch.randn(M, K, device=dev) / K ** 0.5
    n = ops._raw("cidnet_pw_conv_bf16x3_ws_floats", B, M, K, 0); ws = torch.empty(n, device=dev)
    lib().call("cidnet_pw_conv_bf16x3_prep", _p(w), 0, K, 1, _p(ws), n, 1, M, K, _stream())
    out = []
    for (lv, xb, yb) in [(3, 0, 0), (1, 0, 0), (1, 1, 0), (1, 0, 1), (1, 1, 1)]:
        xt = x.to(torch.bfloat16) if xb else x
        y = torch.empty(B, M, HW, device=dev, dtype=torch.bfloat16 if yb else torch.float32)
        tt = timeit(lambda: lib().call("cidnet_pw_conv_bf16x3_pre_t", _p(xt), xb, K * HW, _p(ws), 0, _p(y), yb, M * HW, None, 0, B, M, K, HW, lv, lv, _stream()))
        by = (K * (2 if xb else 4) + M * (2 if yb else 4)) * HW * B
        out.append(f"{tt:6.1f} us ({by / tt / 1e3:5.0f})")
    print(f"  {(B, M, K, HW)}: " + " | ".join(out), flush=True)
