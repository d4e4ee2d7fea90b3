import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import hvi_cidnet_amd as P
from oracle import cidnet_oracle as O
g = np.load("/root/repo/tests/golden/tnsm.npz")
dev = torch.device("cuda:0")
chans = (12, 12, 24, 48)
m = P.CIDNet_TNSM(channels=list(chans))
p = O.make_params(5, channels=chans, variant="tnsm")
m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
m.to(dev).train()
y, fz = m(torch.from_numpy(g["model_x"]).to(dev))
((y - torch.from_numpy(g["model_gt"]).to(dev)).abs().mean() + 0.1 * fz.mean()).backward()
rows = []
for n, prm in m.named_parameters():
    k = f"model_g.{n}"
    if k not in g.files: continue
    ref = torch.from_numpy(g[k]).double(); got = prm.grad.detach().cpu().double()
    rows.append(((got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30), n, ref.abs().max().item()))
rows.sort(reverse=True)
for r in rows[:15]: print(f"{r[0]:.3e}  max|g|={r[2]:.3e}  {r[1]}")
