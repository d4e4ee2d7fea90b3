"""Does chunking the batch keep producer->consumer tensors in the 256 MiB Infinity Cache?  (dev tool)
IEL forward chain at level 1: pw(36->190) -> dw3x3(190) -> gate(190->95) -> pw(95->36)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib

dev = torch.device("cuda:0")
C, h, H, W = 36, 95, 200, 300
HW = H * W
w_in = torch.rand(2 * h, C, device=dev); w_dw = torch.rand(2 * h, 9, device=dev)
w1 = torch.rand(h, 9, device=dev); w2 = torch.rand(h, 9, device=dev); w_out = torch.rand(C, h, device=dev)

def chain(x, pin, u, g, out, B):
    ops.pw_conv(x, 0, C * HW, w_in, 0, 0, C, 1, pin, 0, 2 * h * HW, B, 2 * h, C, HW)
    ops.dw3x3(pin, w_dw, None, 2 * h, u, B, 2 * h, H, W)
    lib().call("cidnet_iel_gate_fwd", ops._p(u), ops._p(w1), ops._p(w2), ops._p(g), B, h, H, W, ops._stream())
    ops.pw_conv(g, 0, h * HW, w_out, 0, 0, h, 1, out, 0, C * HW, B, C, h, HW)

def bench(Btot, cb, iters=10):
    x = torch.rand(Btot, C, HW, device=dev)
    pin = torch.empty(cb, 2 * h, HW, device=dev); u = torch.empty_like(pin)
    g = torch.empty(cb, h, HW, device=dev); out = torch.empty(Btot, C, HW, device=dev)
    def run():
        for b0 in range(0, Btot, cb):
            chain(x[b0:b0 + cb], pin, u, g, out[b0:b0 + cb], cb)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for cb in (8, 4, 2, 1):
    print(f"IEL fwd chain, 8 images at level 1, chunk {cb}: {bench(8, cb):8.1f} us")
