"""Micro-benchmark of cidnet_conv3x3 with ablations (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib

def run(B, M, K, H, W, flags, iters=10):
    dev = torch.device("cuda:0")
    x = torch.rand(B, K, H, W, device=dev); w = torch.rand(M, K, 3, 3, device=dev); y = torch.empty(B, M, H, W, device=dev)
    lib().raw("cidnet_debug_c3_flags")(flags)
    for _ in range(2): ops.conv3x3(x, w, y, B, M, K, H, W, 9 * K, 9)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv3x3(x, w, y, B, M, K, H, W, 9 * K, 9)
    e1.record(); torch.cuda.synchronize()
    lib().raw("cidnet_debug_c3_flags")(0)
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * 9 * M * K * H * W * B / (us * 1e-6) / 1e12

for sh in [(8, 36, 36, 400, 600), (8, 36, 36, 200, 300), (8, 72, 36, 200, 300), (8, 36, 72, 200, 300), (8, 144, 72, 100, 150), (8, 72, 144, 100, 150), (8, 72, 144, 50, 75)]:
    r = {f: run(*sh, flags=f) for f in (0, 1, 2, 3, 4, 7, 16)}
    tp = {t: run(*sh, flags=t << 8)[0] for t in (1, 2, 4, 8)}
    print(f"   tiles/block sweep {sh}: " + " ".join(f"{k}:{v:.0f}" for k, v in tp.items()))
    print(f"{sh}: full {r[0][0]:7.1f} us {r[0][1]:5.1f} TF | no-store {r[1][0]:7.1f} | no-load {r[2][0]:7.1f} | neither {r[3][0]:7.1f} | const-w {r[4][0]:7.1f} | mfma-only {r[7][0]:7.1f} ({r[7][1]:.1f} TF) | padded-tiles {r[16][0]:7.1f}")

def run_wg(B, M, N, H, W, flags, iters=10):
    dev = torch.device("cuda:0")
    x = torch.rand(B, N, H, W, device=dev); gy = torch.rand(B, M, H, W, device=dev); gw = torch.empty(M, N, 3, 3, device=dev)
    lib().raw("cidnet_debug_c3_flags")(flags)
    for _ in range(2): ops.conv3x3_wgrad(gy, x, gw, B, M, N, H, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv3x3_wgrad(gy, x, gw, B, M, N, H, W)
    e1.record(); torch.cuda.synchronize()
    lib().raw("cidnet_debug_c3_flags")(0)
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * 9 * M * N * H * W * B / (us * 1e-6) / 1e12

for sh in [(8, 36, 36, 400, 600), (8, 36, 36, 200, 300), (8, 72, 36, 200, 300), (8, 144, 72, 100, 150), (8, 36, 72, 100, 150), (8, 72, 144, 50, 75)]:
    a, p, nm, ne, pn = run_wg(*sh, flags=0), run_wg(*sh, flags=16), run_wg(*sh, flags=32), run_wg(*sh, flags=64), run_wg(*sh, flags=128)
    print(f"wgrad {sh}: {a[0]:7.1f} us {a[1]:5.1f} TF | padded tiles {p[0]:7.1f} us {p[1]:5.1f} TF | no main loop {nm[0]:7.1f} | no epilogue {ne[0]:7.1f} | N remainder as a padded tile {pn[0]:7.1f}")
