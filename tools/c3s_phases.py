"""Phase timing of the bf16x3 conv kernel (build with CIDNET_EXTRA_FLAGS=-DC3S_TIMING): cycles wave 0 spends staging a
tile, at the two barriers and computing, per tile (dev tool)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hvi_cidnet_amd.ops import _p, _stream, lib
dev = torch.device("cuda:0")
B, M, K, H, W = 8, 36, 36, 400, 600
x = torch.randn(B, K, H, W, device=dev); w = torch.randn(M, K, 3, 3, device=dev) / 18; y = torch.empty(B, M, H, W, device=dev)
fn = lib().raw("cidnet_debug_c3s_phases"); fn.restype = ctypes.c_int
buf = np.zeros(8 * 1024, dtype=np.uint64)
run = lambda: lib().call("cidnet_conv3x3_bf16x3", _p(x), K * H * W, _p(w), 9 * K, 9, 0, None, 0, _p(y), M * H * W, B, M, K, H, W, _stream())
for _ in range(3): run()
fn(buf.ctypes.data_as(ctypes.c_void_p), 1024)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
fn(buf.ctypes.data_as(ctypes.c_void_p), 1024)
a = buf.reshape(1024, 8).astype(np.float64); live = a[a[:, 4] > 0]
t = live[:, 4].mean()
print(f"kernel {1e3 * e0.elapsed_time(e1):.1f} us, {len(live)} blocks, {t:.1f} tiles per block; wave 0, kcycles per tile: "
      f"staging {live[:, 0].mean() / t / 1e3:.2f}  barriers {live[:, 1].mean() / t / 1e3:.2f}  compute {live[:, 2].mean() / t / 1e3:.2f}")
