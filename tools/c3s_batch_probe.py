"""Concurrency probe: the 1->36 stem conv on one stream while [3->36 stem conv, 36->36 conv] run on another, as in the
first stage of the two-stream forward.  Is the stem output reproducible with each 36->36 kernel?  (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, H, W, B = 36, 36, 400, 600, 8
i1 = torch.rand(B, 1, H, W, device=dev); i3 = torch.rand(B, 3, H, W, device=dev)
s1 = torch.randn(36, 1, 3, 3, device=dev) / 3; s3 = torch.randn(36, 3, 3, 3, device=dev) / 5
wb = torch.randn(M, K, 3, 3, device=dev) / 18
def stem(inp, ws, x=None):
    ci = inp.shape[1]
    if x is None: x = torch.empty(B, 36, H, W, device=dev)
    ops.CONV3_BF16X3["on"] = False
    ops.conv3x3(inp, ws, x, B, 36, ci, H, W, 9 * ci, 9, replicate=True)
    return x
def conv(x, w, bf):
    Kc = x.shape[1]
    y = torch.empty(B, M, H, W, device=dev)
    ops.CONV3_BF16X3["on"] = bf
    ops.conv3x3(x, w, y, B, M, Kc, H, W, 9 * Kc, 9)
    ops.CONV3_BF16X3["on"] = False
    return y
ref1 = stem(i1, s1); ref3 = stem(i3, s3)
torch.cuda.synchronize()
side, other = torch.cuda.Stream(), torch.cuda.Stream()
x24 = torch.randn(B, 24, H, W, device=dev); w24 = torch.randn(M, 24, 3, 3, device=dev) / 15
buf = torch.empty(B, 36, H, W, device=dev)
main = torch.cuda.current_stream()
for what in ("bf16x3 K=36 no stem on side",):
    for trial in range(6):
        buf.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            if what.endswith("no stem on side"): y3 = conv(ref3, wb, True)
            elif "K=24" in what: x3 = stem(i3, s3); y3 = conv(x24, w24, True)
            else: x3 = stem(i3, s3); y3 = conv(x3, wb, what.startswith("bf16x3"))
        x1 = stem(i1, s1, buf)
        torch.cuda.synchronize()
        nan = torch.isnan(x1)
        wrong = (~nan) & ((x1 - ref1).abs() > 1e-6)
        print(f"side: {what}, trial {trial}: stem(main) never-written {nan.sum().item()}, written-wrong {wrong.sum().item()}")
