"""Micro-benchmark of cidnet_pw_conv on one shape with ablations (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib

def run(B, M, K, HW, flags, iters=20):
    dev = torch.device("cuda:0")
    x = torch.rand(B, K, HW, device=dev)
    w = torch.rand(M, K, device=dev)
    y = torch.empty(B, M, HW, device=dev)
    lib().raw("cidnet_debug_pw_flags")(flags)
    for _ in range(3):
        ops.pw_conv(x, 0, K * HW, w, 0, 0, K, 1, y, 0, M * HW, B, M, K, HW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.pw_conv(x, 0, K * HW, w, 0, 0, K, 1, y, 0, M * HW, B, M, K, HW)
    e1.record()
    torch.cuda.synchronize()
    lib().raw("cidnet_debug_pw_flags")(0)
    us = e0.elapsed_time(e1) * 1e3 / iters
    gb = 4.0 * B * HW * (M + K) / 1e9
    tf = 2.0 * B * HW * M * K / 1e12
    return us, gb / (us * 1e-6) / 1e3, tf / (us * 1e-6)

if __name__ == "__main__":
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(8, 190, 36, 60000), (8, 36, 190, 60000), (8, 36, 36, 60000), (8, 36, 95, 60000), (8, 72, 72, 15000),
              (8, 382, 72, 15000), (8, 766, 144, 3750), (8, 144, 766, 3750), (8, 36, 36, 240000)]
    for sh in shapes:
        mts = " ".join(f"MT{mt}:{run(*sh, flags=(512 << 8) | (mt << 28))[0]:.1f}" for mt in (1, 2, 3, 4))
        print(f"  forced channel tiles {sh}: {mts}")
        tb = " ".join(f"{tbk}:{run(*sh, flags=(tbk << 8))[0]:.1f}" for tbk in (256, 512, 768, 1024, 1536, 2048, 4096))
        print(f"  target-blocks sweep {sh}: {tb}")
        r = [run(*sh, flags=f | (1024 << 8)) for f in (0, 1, 2, 3, 8, 9, 4)]
        print(f"B,M,K,HW={sh}: full {r[0][0]:7.1f} us ({r[0][1]:.2f} TB/s, {r[0][2]:.1f} TF) | no-store {r[1][0]:7.1f} | no-k {r[2][0]:7.1f} | neither {r[3][0]:7.1f} | no-load {r[4][0]:7.1f} | no-load-no-store {r[5][0]:7.1f} | LDS-kernel {r[6][0]:7.1f}")
