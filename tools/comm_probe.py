"""Where does the one-rank RCCL rehearsal lose time? (dev tool)  modes: none | init | comm [n_buckets]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
mode = sys.argv[1]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
if mode == "comm": os.environ["CIDNET_DP_FORCE_ALLREDUCE"] = "1"
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
if mode in ("init", "comm"):
    if os.environ.get("LAZY") == "1": dist.init_process_group("nccl", rank=0, world_size=1)
    else: dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer
m = P.CIDNet().to(dev)
tr = DataParallelTrainer(m, lr=1e-4, n_buckets=nb)
x = torch.rand(8, 3, 400, 600, device=dev); gt = torch.rand(8, 3, 400, 600, device=dev)
for _ in range(3): tr.step(x, gt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): tr.step(x, gt)
torch.cuda.synchronize(); print(f"{mode} n_buckets={nb}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/step", flush=True)
if mode in ("init", "comm"): dist.destroy_process_group()
