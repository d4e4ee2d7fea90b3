"""dev tool: where does the HOST spend its time enqueueing one training step?  cProfile over a few steps (the GPU runs behind;
the step is enqueue-bound only when this exceeds the GPU time).  python tools/host_profile.py [f32|bf16]"""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer
P.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = P.CIDNet().to(dev)
tr = DataParallelTrainer(m, lr=1e-4)
x = torch.rand(8, 3, 400, 600, device=dev); gt = torch.rand(8, 3, 400, 600, device=dev)
for _ in range(8):
    tr.step(x, gt)
torch.cuda.synchronize()
n = 6
t0 = time.perf_counter()
for _ in range(n):
    tr.step(x, gt)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue without profiler: {1e3 * (t1 - t0) / n:.2f} ms/step (GPU: {1e3 * (time.perf_counter() - t0) / n:.2f})")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    tr.step(x, gt)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
print(s.getvalue()[:6000])
