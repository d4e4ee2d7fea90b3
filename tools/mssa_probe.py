import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer
dev = torch.device("cuda:0")
B = 16
torch.manual_seed(0)
m = P.CIDNet_MSSA().to(dev)
tr = DataParallelTrainer(m, lr=1e-4)
x = torch.rand(B, 3, 400, 600, device=dev); gt = torch.rand(B, 3, 400, 600, device=dev)
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(x, gt)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats(dev)
    print(f"step {i}: host {1e3*(t1-t0):7.1f} ms total {1e3*(t2-t0):7.1f} ms  reserved {st['reserved_bytes.all.current']/2**30:6.1f} GiB  allocs {st.get('num_device_alloc',0)}", flush=True)
