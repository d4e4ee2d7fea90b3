"""Where the split-product 3x3 weight gradient spends its time (dev tool; needs a -DCIDNET_DEBUG build:
CIDNET_EXTRA_FLAGS=-DCIDNET_DEBUG CIDNET_LIB_OUT=hvi-cidnet_amd/libcidnet_hip_dbg.so CIDNET_OBJ_DIR=/tmp/objdbg python hvi-cidnet_amd/build.py,
then CIDNET_LIB_PATH=hvi-cidnet_amd/libcidnet_hip_dbg.so python tools/c3xw_ablate.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib
dev = torch.device("cuda:0")
setf = lib().raw("cidnet_debug_c3xw_flags")
setf.argtypes = [__import__("ctypes").c_int]; setf.restype = None
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
for M, N, H, W in [(36, 36, 400, 600), (144, 72, 100, 150), (72, 36, 200, 300)]:
    dy = torch.randn(B, M, H, W, device=dev); x = torch.randn(B, N, H, W, device=dev); dw = torch.empty(M, N, 3, 3, device=dev)
    row = []
    for flags, what in [(0, "all"), (0, "all again"), (8, "no split/write"), (10, "loads only"), (1, "first tile staged only"), (2, "staging only"), (3, "neither")]:
        setf(flags)
        row.append(f"{what} {timeit(lambda: ops.conv3x3_wgrad(dy, x, dw, B, M, N, H, W)):7.1f} us")
    setf(0)
    print(f"wgrad {M}x{N} @ {H}x{W}: " + " | ".join(row))
