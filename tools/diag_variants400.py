"""dev tool: per-tensor gradient error of the MSSA / TNSM variants at 1x3x400x600 against tests/golden/round4.npz
(ours vs fp64, the reference's own fp32 vs fp64, both relative to the tensor's max) -> gpurun_out/diag_variants400_<variant>.txt"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cidnet_oracle as O
import hvi_cidnet_amd as P
g = np.load(os.path.join(ROOT, "tests", "golden", "round4.npz"), allow_pickle=True)
dev = torch.device("cuda:0")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for variant in ("mssa", "tnsm"):
    tag = f"{variant}400"
    m = (P.CIDNet_MSSA if variant == "mssa" else P.CIDNet_TNSM)()
    p = O.make_params(5, variant=variant)
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()}, strict=True)
    m.to(dev).train()
    x = O.synthetic_batch(191, (1, 3, 400, 600)).to(dev).requires_grad_(True)
    gt = O.synthetic_batch(192, (1, 3, 400, 600)).to(dev)
    res = m(x)
    y = res[0] if variant == "tnsm" else res
    loss = (y - gt).abs().mean() + (0.1 * res[1].mean() if variant == "tnsm" else 0.0)
    loss.backward()
    torch.cuda.synchronize()
    rows = []
    x64 = torch.from_numpy(g[f"{tag}64_gx_strided"]).double(); xr = torch.from_numpy(g[f"{tag}_gx_strided"]).double()
    ours = x.grad[:, :, ::8, ::8].cpu().double()
    sc = x64.abs().max().item()
    e_o, e_r = (ours - x64).abs(), (xr - x64).abs()
    rows.append(("d/dx", x.grad.numel(), sc, e_o.max().item() / sc, e_r.max().item() / sc))
    bar = 2 * e_r.max().item() + 2e-4 * sc
    print(variant, "d/dx: ours max", e_o.max().item() / sc, "ref max", e_r.max().item() / sc, "n over bar", int((e_o > bar).sum()), "of", e_o.numel(),
          "ours p99.9", torch.quantile(e_o.flatten()[:1000000], 0.999).item() / sc, "ref p99.9", torch.quantile(e_r.flatten()[:1000000], 0.999).item() / sc)
    dead = set(str(n) for n in g[f"{tag}_dead"])
    for name, prm in m.named_parameters():
        if name in dead:
            continue
        s = O.grad_fingerprint(prm.grad, 512)[1].double()
        s64 = torch.from_numpy(g[f"{tag}64_gs.{name}"]).double(); sr = torch.from_numpy(g[f"{tag}_gs.{name}"]).double()
        sc = max(s64.abs().max().item(), 1e-30)
        rows.append((name, prm.grad.numel(), sc, (s - s64).abs().max().item() / sc, (sr - s64).abs().max().item() / sc))
    with open(os.path.join(ROOT, "gpurun_out", f"diag_variants400_{variant}.txt"), "w") as f:
        for r in sorted(rows, key=lambda r: -r[3] / max(r[4], 1e-12)):
            f.write(f"{r[0]:60s} n={r[1]:8d} max|g|={r[2]:.3e} ours={r[3]:.3e} ref={r[4]:.3e} ratio={r[3] / max(r[4], 1e-12):8.2f}\n")
    o = np.array([r[3] for r in rows]); rr = np.array([r[4] for r in rows])
    print(variant, "tensors", len(rows), "ours median/p90/max", np.median(o), np.percentile(o, 90), o.max(), "ref median/p90/max", np.median(rr), np.percentile(rr, 90), rr.max(),
          "n(ours > 2 ref + 2e-4)", int((o > 2 * rr + 2e-4).sum()))
