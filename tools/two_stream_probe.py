"""Is the two-stream forward(+backward) bit-identical to the single-stream one, in fp32 and bf16-storage mode?  (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from oracle import cidnet_oracle as O
dev = torch.device("cuda:0")
m = P.CIDNet()
p = O.make_params(7)
m.load_state_dict({k: p[k] for k in m.state_dict().keys()}); m.to(dev)
x = O.synthetic_batch(91, (8, 3, 400, 600)).to(dev)
def run(two):
    m.two_streams = two
    for q in m.parameters(): q.grad = None
    y = m(x)
    y.square().mean().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), torch.cat([q.grad.flatten() for q in m.parameters() if q.grad is not None]).clone()
for mode in ("f32", "bf16"):
    P.set_storage_dtype(mode)
    y1, g1 = run(False)
    for t in range(4):
        y2, g2 = run(True)
        print(mode, "trial", t, "output max diff %.3e (elements differing: %d)  grads max rel diff %.3e" % (
            (y1 - y2).abs().max().item(), (y1 != y2).sum().item(), ((g1 - g2).abs().max() / g1.abs().max()).item()))
