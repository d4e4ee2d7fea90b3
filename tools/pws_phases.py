"""Phase timing of the bf16x3 1x1-conv kernel (build with CIDNET_EXTRA_FLAGS=-DPWS_TIMING): s_memtime cycles wave 0 spends
per k-block reading fragments from LDS, splitting / issuing loads, in the MFMA burst and at the barrier (dev tool)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.ops import lib
dev = torch.device("cuda:0")
fn = lib().raw("cidnet_debug_pws_phases"); fn.restype = ctypes.c_int
buf = np.zeros(8 * 1024, dtype=np.uint64)
B = 8
for M, K, HW in ((192, 512, 60000), (144, 766, 3750), (766, 144, 3750), (72, 382, 15000), (144, 144, 15000)):
    x = torch.randn(B, K, HW, device=dev); w = torch.randn(M, K, device=dev); y = torch.empty(B, M, HW, device=dev)
    run = lambda: ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, 0, K, 1, y, 0, M * HW, B, M, K, HW)
    for _ in range(3): run()
    fn(buf.ctypes.data_as(ctypes.c_void_p), 1024)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    fn(buf.ctypes.data_as(ctypes.c_void_p), 1024)
    a = buf.reshape(1024, 8).astype(np.float64); live = a[a[:, 4] > 0]
    t = live[:, 4].mean()
    # s_memtime counts at 100 MHz on this part: report in stamps
    print(f"M={M} K={K} HW={HW}: call {1e3 * e0.elapsed_time(e1):.1f} us, {len(live)} blocks, {t:.1f} k-blocks; wave 0, stamp units per "
          f"k-block: lds-read {live[:, 0].mean() / t:.1f}  split+issue {live[:, 1].mean() / t:.1f}  mfma {live[:, 2].mean() / t:.1f}  "
          f"barrier {live[:, 3].mean() / t:.1f}")
