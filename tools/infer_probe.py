"""dev tool: per-iteration time of the 32x3x1024x1024 inference forward (allocator segments next to it)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd import ops
dev = torch.device("cuda:0")
x = torch.rand(32, 3, 1024, 1024, device=dev)
m = P.CIDNet().to(dev).eval()
segs = lambda: torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
with torch.no_grad(), ops.prepared_weights(os.environ.get("PREP", "1") == "1"):
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); s0 = segs()
        m(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"iter {i}: host {1e3*(t1-t0):7.1f} ms  total {1e3*(t2-t0):7.1f} ms  new segments {segs()-s0}  reserved {torch.cuda.memory_reserved(dev)/2**30:.1f} GiB", flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    print(f"3 unsynchronised forwards: {1e3*(time.perf_counter()-t0)/3:.1f} ms each")
