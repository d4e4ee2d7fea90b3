"""Two-stream forward+backward with the bf16x3 conv enabled and the branch serialisation lifted: which tensors differ
from the single-stream run?  (dev tool for the interference described in DESIGN.md section 4 (i))"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd import ops
from oracle import cidnet_oracle as O
dev = torch.device("cuda:0")
m = P.CIDNet()
p = O.make_params(7)
m.load_state_dict({k: p[k] for k in m.state_dict().keys()}); m.to(dev)
x = O.synthetic_batch(91, (8, 3, 400, 600)).to(dev)
caps = {}
for name, mod in m.named_children():
    mod.register_forward_hook(lambda mod, i, o, name=name: caps.__setitem__(name, o.detach() if torch.is_tensor(o) else None))
ops.CONV3_BF16X3["on"] = True
ops.CONV3_BF16X3["allow_two_streams"] = True
def run(two):
    m.two_streams = two
    for q in m.parameters(): q.grad = None
    y = m(x)
    y.square().mean().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}, dict(caps)
y1, g1, c1 = run(False)
for t in range(3):
    y2, g2, c2 = run(True)
    badf = {k: int((c1[k] != c2[k]).sum()) for k in c1 if c1[k] is not None and not torch.equal(c1[k], c2[k])}
    badg = {k: int((g1[k] != g2[k]).sum()) for k in g1 if not torch.equal(g1[k], g2[k])}
    print("trial", t, "output differs in", int((y1 != y2).sum()), "elements; forward modules differing:", badf)
    print("   gradients differing (first 12):", dict(list(badg.items())[:12]), "of", len(badg))
