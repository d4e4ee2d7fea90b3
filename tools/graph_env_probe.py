"""dev tool: whole-step hipGraph replay against eager issue under the runtime's graph knobs (set in the environment BEFORE the
process starts): DEBUG_CLR_GRAPH_PACKET_CAPTURE, DEBUG_HIP_FORCE_GRAPH_QUEUES, DEBUG_HIP_GRAPH_BATCH_SIZE.
    python tools/graph_env_probe.py [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer
P.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32")
dev = torch.device("cuda:0")
x = torch.rand(8, 3, 400, 600, device=dev); gt = torch.rand(8, 3, 400, 600, device=dev)


def timed(tr, n=12):
    for _ in range(8): tr.step(x, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): tr.step(x, gt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, 1e3 * (t1 - t0) / n

env = {k: os.environ.get(k) for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DEBUG_HIP_FORCE_GRAPH_QUEUES", "DEBUG_HIP_GRAPH_BATCH_SIZE")}
res = {}
for graph in (False, True):
    torch.manual_seed(0)
    m = P.CIDNet().to(dev)
    tr = DataParallelTrainer(m, lr=1e-4, use_graph=graph)
    res[graph] = timed(tr)
    del tr, m
    torch.cuda.empty_cache()
print(f"{env}: eager {res[False][0]:.2f} ms/step (host {res[False][1]:.2f}) | graph replay {res[True][0]:.2f} ms/step (host {res[True][1]:.2f})", flush=True)
