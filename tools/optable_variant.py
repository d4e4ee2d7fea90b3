"""dev tool: per-entry-point time of one training step of a variant (single stream, HIP events per C-ABI call).
    python tools/optable_variant.py tnsm|mssa [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd import ops
from hvi_cidnet_amd.dp import DataParallelTrainer
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import OpTimer
dev = torch.device("cuda:0")
variant = sys.argv[1] if len(sys.argv) > 1 else "tnsm"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
m = (P.CIDNet_TNSM if variant == "tnsm" else P.CIDNet_MSSA)().to(dev)
m.two_streams = False
lf = (lambda y, t: ops.L1LossFn.apply(y[0], t) + 0.1 * y[1].mean()) if variant == "tnsm" else None
tr = DataParallelTrainer(m, lr=1e-4, loss_fn=lf, wgrad_stream=False)
x = torch.rand(B, 3, 400, 600, device=dev); gt = torch.rand(B, 3, 400, 600, device=dev)
for _ in range(4):
    tr.step(x, gt)
torch.cuda.synchronize()
t = OpTimer().install()
for _ in range(2):
    tr.step(x, gt)
torch.cuda.synchronize()
agg = t.table(); t.remove()
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {k:44s} calls/step {v[0] // 2:5d}  ms/step {v[1] / 2:9.3f}  {100 * v[1] / tot:5.1f}%")
print(f"  sum {tot / 2:.2f} ms/step")
for k, v in sorted(t.table(by_shape=True).items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"    {k:80s} x{v[0] // 2:3d}  {v[1] / 2:8.3f} ms")
