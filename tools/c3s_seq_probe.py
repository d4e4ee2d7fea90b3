"""Does the bf16x3 conv disturb the packed-fp32 stem conv when both run on ONE stream, back to back?  (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K, H, W, B = 36, 36, 400, 600, 8
i1 = torch.rand(B, 1, H, W, device=dev); s1 = torch.randn(36, 1, 3, 3, device=dev) / 3
x36 = torch.randn(B, K, H, W, device=dev); wb = torch.randn(M, K, 3, 3, device=dev) / 18
def stem(out):
    ops.CONV3_BF16X3["on"] = False
    ops.conv3x3(i1, s1, out, B, 36, 1, H, W, 9, 9, replicate=True)
def conv(bf, out):
    ops.CONV3_BF16X3["on"] = bf
    ops.conv3x3(x36, wb, out, B, M, K, H, W, 9 * K, 9)
    ops.CONV3_BF16X3["on"] = False
ref = torch.empty(B, 36, H, W, device=dev); stem(ref)
cref = torch.empty(B, M, H, W, device=dev); conv(True, cref)
torch.cuda.synchronize()
buf = torch.empty_like(ref); y = torch.empty(B, M, H, W, device=dev)
for what in ("fp32 conv then stem", "bf16x3 conv then stem", "bf16x3 conv, device sync, stem", "stem then bf16x3 conv (conv checked)"):
    bad = []
    for t in range(12):
        buf.fill_(float("nan")); y.fill_(float("nan"))
        torch.cuda.synchronize()
        if what.startswith("stem"):
            stem(buf); conv(True, y)
            torch.cuda.synchronize()
            bad.append(int((y != cref).sum()))
            continue
        conv(what.startswith("bf16x3"), y)
        if "sync" in what: torch.cuda.synchronize()
        stem(buf)
        torch.cuda.synchronize()
        bad.append(int((buf != ref).sum()))
    print(f"{what}: wrong elements per trial {bad}")
