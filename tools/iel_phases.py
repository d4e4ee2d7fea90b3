"""Phase timing of the tile-resident IEL forward kernel (build with CIDNET_EXTRA_FLAGS=-DIEL_TIMING): cycles that the
first GEMM wave and the first stencil wave of each block spend in each stage and at each barrier (dev tool)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hvi_cidnet_amd.ops import _p, _stream, lib

dev = torch.device("cuda:0")
B, C, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 36, 200, 300)))
h = int(C * 2.66)
xn = torch.randn(B, C, H, W, device=dev)
w_in = torch.randn(2 * h, C, device=dev) / C ** 0.5
w_dw = torch.randn(2 * h, 9, device=dev) / 3
w1 = torch.randn(h, 9, device=dev) / 3
w2 = torch.randn(h, 9, device=dev) / 3
w_out = torch.randn(C, h, device=dev) / h ** 0.5
u = None if os.environ.get("IEL_NOU") else torch.empty(B, 2 * h, H, W, device=dev)
out = torch.empty_like(xn)
fn = lib().raw("cidnet_debug_iel_phases")
fn.restype = ctypes.c_int
nb = 4096
buf = np.zeros(8 * nb, dtype=np.uint64)
def run():
    lib().call("cidnet_iel_fwd", _p(xn), None, _p(w_in), _p(w_dw), _p(w1), _p(w2), _p(w_out), _p(u), _p(out), B, C, h, H, W, _stream())
for _ in range(3):
    run()
fn(buf.ctypes.data_as(ctypes.c_void_p), nb)          # clears
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
fn(buf.ctypes.data_as(ctypes.c_void_p), nb)
a = buf.reshape(nb, 8).astype(np.float64)
live = a[a.sum(1) > 0]
names = ["S3", "waitA", "S0", "waitB", "S1", "waitA", "S2", "waitB"]
print(f"shape {(B, C, H, W)}: kernel {1e3 * e0.elapsed_time(e1):.1f} us, {len(live)} blocks stamped; kcycles per block (mean):")
m = live.mean(0) / 1e3
print("  GEMM wave   : " + "  ".join(f"{n} {v:7.1f}" for n, v in zip(names[:4], m[:4])) + f"   total {m[:4].sum():.1f}")
print("  stencil wave: " + "  ".join(f"{n} {v:7.1f}" for n, v in zip(names[4:], m[4:])) + f"   total {m[4:].sum():.1f}")
