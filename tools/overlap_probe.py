"""Do an ALU-bound kernel (dense 3x3 conv, fp32 MFMA) and an HBM-bound kernel (depthwise backward) overlap when they run on
two streams?  Times A alone, B alone and A || B (dev tool; decides whether desynchronising the two branches could pay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops

dev = torch.device("cuda:0")
B = 8
x = torch.randn(B, 36, 400, 600, device=dev); w = torch.randn(36, 36, 3, 3, device=dev) / 18; y = torch.empty_like(x)
pin = torch.randn(B, 190, 200, 300, device=dev); du = torch.randn_like(pin); dpin = torch.empty_like(pin)
wd = torch.randn(190, 1, 3, 3, device=dev); gw = torch.empty_like(wd)
xs = torch.randn(B, 36, 200, 300, device=dev); w1 = torch.randn(190, 36, 1, 1, device=dev) / 6; p1 = torch.empty(B, 190, 200, 300, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 10
def A():
    for _ in range(N): ops.conv3x3(x, w, y, B, 36, 36, 400, 600, 9 * 36, 9)
def Bk():
    for _ in range(N): ops.dw3x3_bwd(pin, du, wd, None, 190, dpin, gw, None, B, 190, 200, 300)
def Ck():
    for _ in range(N): ops.pw_conv(xs, 0, 36 * 60000, w1, 0, 0, 36, 1, p1, 0, 190 * 60000, B, 190, 36, 60000)
def run(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    if fa:
        with torch.cuda.stream(sa): fa()
    if fb:
        with torch.cuda.stream(sb): fb()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for name, fb in (("dw3x3_bwd 190ch 200x300 (HBM)", Bk), ("pw_conv 36->190 200x300 (HBM)", Ck)):
    for _ in range(2): run(A, fb)
    ta, tb, tab = min(run(A, None) for _ in range(3)), min(run(None, fb) for _ in range(3)), min(run(A, fb) for _ in range(3))
    print(f"conv3x3 36->36 400x600 x{N}: {ta:.2f} ms | {name} x{N}: {tb:.2f} ms | both on two streams: {tab:.2f} ms  (sum {ta + tb:.2f}, max {max(ta, tb):.2f})")
