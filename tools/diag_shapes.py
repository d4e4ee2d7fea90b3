"""Per-parameter gradient differences against the live oracle for given input shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from oracle import cidnet_oracle as O

dev = torch.device("cuda:0")
chans = (12, 12, 24, 48)
for arg in sys.argv[1:]:
    shape = tuple(int(v) for v in arg.split(","))
    p = O.make_params(11, channels=chans)
    m = P.CIDNet(channels=list(chans))
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()}, strict=True)
    m.to(dev)
    x = O.synthetic_batch(101 + shape[2], shape)
    r = O.synthetic_batch(202 + shape[3], shape) - 0.5
    y = m(x.to(dev)); (y * r.to(dev)).sum().backward()
    for dt in (torch.float32, torch.float64):
        po = O.params_to(p, dtype=dt, requires_grad=True)
        yo = O.cidnet_forward(po, x.to(dt)); (yo * r.to(dt)).sum().backward()
        rows = []
        for n, prm in m.named_parameters():
            if prm.grad is None: continue
            g = po[n].grad.double(); d = (prm.grad.cpu().double() - g).abs().max().item()
            rows.append((d / (g.abs().max().item() + 1e-30), n, d))
        rows.sort(reverse=True)
        print(shape, dt, "fwd diff", (y.detach().cpu().double() - yo.detach().double()).abs().max().item())
        for rel, n, d in rows[:8]: print(f"   {rel:.2e} {d:.2e} {n}")
