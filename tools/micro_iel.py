"""IEL forward: tile-resident kernel (csrc/iel.hip) vs the unfused chain (pw_conv -> dw+gate -> pw_conv) at the
benchmark's shapes: max abs difference of out / u, and time per launch (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.ops import _p, _stream, lib, pw_conv

dev = torch.device("cuda:0")
torch.manual_seed(0)


def chain(xn, res, w_in, w_dw, w1, w2, w_out, train):
    B, C, H, W = xn.shape
    HW, h = H * W, w1.shape[0]
    pin = torch.empty((B, 2 * h, H, W), device=dev)
    pw_conv(xn, 0, C * HW, w_in, 0, 0, C, 1, pin, 0, 2 * h * HW, B, 2 * h, C, HW)
    u = torch.empty_like(pin) if train else None
    gate = torch.empty((B, h, H, W), device=dev)
    lib().call("cidnet_iel_dw_gate_fwd", _p(pin), _p(w_dw), _p(w1), _p(w2), _p(u), _p(gate), B, h, H, W, _stream())
    out = torch.empty_like(xn)
    pw_conv(gate, 0, h * HW, w_out, 0, 0, h, 1, out, 0, C * HW, B, C, h, HW, res=res, r_bs=C * HW if res is not None else 0)
    return out, u


def fused(xn, res, w_in, w_dw, w1, w2, w_out, train):
    B, C, H, W = xn.shape
    h = w1.shape[0]
    u = torch.empty((B, 2 * h, H, W), device=dev) if train else None
    out = torch.empty_like(xn)
    lib().call("cidnet_iel_fwd", _p(xn), _p(res), _p(w_in), _p(w_dw), _p(w1), _p(w2), _p(w_out), _p(u), _p(out), B, C, h, H, W, _stream())
    return out, u


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


shapes = [(8, 36, 200, 300), (8, 72, 100, 150), (8, 144, 50, 75), (2, 36, 37, 51), (1, 12, 9, 13), (3, 24, 8, 8), (32, 36, 512, 512)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, C, H, W) in shapes:
    h = int(C * 2.66)
    xn = torch.randn(B, C, H, W, device=dev)
    res = torch.randn(B, C, H, W, device=dev)
    w_in = torch.randn(2 * h, C, 1, 1, device=dev) / C ** 0.5
    w_dw = torch.randn(2 * h, 1, 3, 3, device=dev) / 3
    w1 = torch.randn(h, 1, 3, 3, device=dev) / 3
    w2 = torch.randn(h, 1, 3, 3, device=dev) / 3
    w_out = torch.randn(C, h, 1, 1, device=dev) / h ** 0.5
    args = (xn, res, w_in, w_dw, w1, w2, w_out)
    big = B * C * H * W > 2e8
    o0, u0 = chain(*args, not big)
    o1, u1 = fused(*args, not big)
    torch.cuda.synchronize()
    do = (o0 - o1).abs().max().item()
    du = (u0 - u1).abs().max().item() if not big else float("nan")
    tc = timeit(lambda: chain(*args, True and not big), 10)
    tf = timeit(lambda: fused(*args, True and not big), 10)
    tfi = timeit(lambda: fused(*args, False), 10)
    print(f"{(B, C, H, W)}: max|dout| {do:.2e} (|out| max {o0.abs().max().item():.2f})  max|du| {du:.2e}   chain {tc:8.1f} us   fused {tf:8.1f} us   fused(no u) {tfi:8.1f} us", flush=True)
