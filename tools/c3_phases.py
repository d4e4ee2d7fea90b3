"""Where a conv3_kernel block spends its cycles: weight staging / k loops (MFMA bursts) / epilogues (dev tool).
Needs the library built with CIDNET_EXTRA_FLAGS=-DC3_TIMING (the product build has no timing code)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hvi_cidnet_amd import ops
from hvi_cidnet_amd._lib import lib

dev = torch.device("cuda:0")
fn = lib().raw("cidnet_debug_c3_phases")
for (B, M, K, H, W) in [(8, 36, 36, 400, 600), (8, 36, 36, 200, 300), (8, 72, 36, 200, 300), (8, 144, 72, 100, 150), (8, 72, 144, 50, 75)]:
    x = torch.randn(B, K, H, W, device=dev); w = torch.randn(M, K, 3, 3, device=dev); y = torch.empty(B, M, H, W, device=dev)
    for _ in range(3): ops.conv3x3(x, w, y, B, M, K, H, W, 9 * K, 9)
    torch.cuda.synchronize()
    fn(np.zeros(4 * 8192, dtype=np.uint64).ctypes.data_as(ctypes.c_void_p), 8192)      # read + clear
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv3x3(x, w, y, B, M, K, H, W, 9 * K, 9); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nb = 8192
    buf = np.zeros(4 * nb, dtype=np.uint64)
    fn(buf.ctypes.data_as(ctypes.c_void_p), nb)
    p = buf.reshape(nb, 4).astype(np.float64)
    p = p[p[:, 3] > 0]
    tot = p[:, 3]
    # s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md): kilocycles below; 2.4 kcycles = 1 us at 2.4 GHz
    print(f"B,M,K,H,W={(B, M, K, H, W)}: {us:7.1f} us, {len(p)} blocks sampled; per block (kcycles): stage {p[:,0].mean()/1e3:6.1f}  k loops {p[:,1].mean()/1e3:7.1f}  "
          f"epilogue {p[:,2].mean()/1e3:6.1f}  total {tot.mean()/1e3:7.1f} (min {tot.min()/1e3:.1f} max {tot.max()/1e3:.1f})", flush=True)

# weight gradient: [wait for the prefetched rows at the top of each row step, main loop, epilogue, total]
for (B, M, N, H, W) in [(8, 36, 36, 400, 600), (8, 36, 36, 200, 300), (8, 144, 72, 100, 150)]:
    dy = torch.randn(B, M, H, W, device=dev); x = torch.randn(B, N, H, W, device=dev); gw = torch.empty(M, N, 3, 3, device=dev)
    lib().raw("cidnet_debug_c3_flags")(128)       # one launch (the N remainder as a padded tile): the phase buffer then holds one kind of block
    for _ in range(2): ops.conv3x3_wgrad(dy, x, gw, B, M, N, H, W)
    torch.cuda.synchronize()
    fn(np.zeros(4 * 8192, dtype=np.uint64).ctypes.data_as(ctypes.c_void_p), 8192)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv3x3_wgrad(dy, x, gw, B, M, N, H, W); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(4 * 8192, dtype=np.uint64)
    fn(buf.ctypes.data_as(ctypes.c_void_p), 8192)
    p = buf.reshape(8192, 4).astype(np.float64); p = p[p[:, 3] > 0]
    lib().raw("cidnet_debug_c3_flags")(0)
    print(f"wgrad B,M,N,H,W={(B, M, N, H, W)}: {e0.elapsed_time(e1) * 1e3:7.1f} us, {len(p)} blocks; "
          f"per block (kcycles): load wait {p[:,0].mean()/1e3:7.1f}  main loop {p[:,1].mean()/1e3:7.1f}  epilogue {p[:,2].mean()/1e3:6.1f}  total {p[:,3].mean()/1e3:7.1f}", flush=True)
