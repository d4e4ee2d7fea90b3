"""Pin the oracle to the reference and write tests/golden/*.npz.  DEVELOPMENT CONTAINER ONLY.

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

What it does
  1. imports the reference's `net` package from /root/reference (read-only, nothing copied);
  2. loads this repo's deterministic parameters (oracle.make_params) into the reference modules;
  3. asserts oracle == reference on CPU: HVIT / PHVIT forward bit-exact on random, uint8-quantised
     and adversarial pixels; whole-network forward bit-exact; gradients equal to rounding; the
     known-answer facts of SURVEY.md section 8c;
  4. writes small fixtures (inputs + the REFERENCE's outputs/gradients) under tests/golden/.
The fixtures are data only.  tests/test_oracle_golden.py re-checks the oracle against them on
any machine; the GPU parity tests check the HIP path against them too.
"""
import os
import sys

os.environ.setdefault("HF_HUB_OFFLINE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import cidnet_oracle as O

from net.CIDNet import CIDNet as RefCIDNet          # noqa: E402  (reference, imported not copied)
from net.HVI_transform import RGB_HVI as RefHVI      # noqa: E402
from net.LCA import HV_LCA as RefHVLCA, I_LCA as RefILCA  # noqa: E402
from net.transformer_utils import NormDownsample as RefDown, NormUpsample as RefUp, LayerNorm as RefLN  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)


class KinkMargin:
    """Smallest distance of any non-smooth point's argument from its kink over one forward of a reference
    model: PReLU / ReLU / LeakyReLU inputs from 0 and the runner-up gap of a channel max (SpatialAttention,
    net/CIDNet_MSSA.py:22).  A gradient fixture whose margin is below the forward tolerance tests on which
    side of a kink fp32 rounding happens to fall, not parity: one flipped PReLU pixel moves whole gradient
    tensors by percents.  Whole-model fixtures are therefore generated from the first input seed whose
    margin is >= MIN_MARGIN, and the margin is stored next to them."""
    MIN_MARGIN = 4e-6

    def __init__(self, model):
        self.margin, self.hooks = float("inf"), []
        for mod in model.modules():
            if isinstance(mod, (torch.nn.PReLU, torch.nn.ReLU, torch.nn.LeakyReLU)):
                self.hooks.append(mod.register_forward_pre_hook(self._act))
            elif type(mod).__name__ == "SpatialAttention":
                self.hooks.append(mod.register_forward_pre_hook(self._chmax))

    def _act(self, mod, inp):
        if inp[0].dim() == 4 and inp[0].shape[-1] > 1:      # feature maps, not the pooled TNSM vectors
            self.margin = min(self.margin, inp[0].detach().abs().min().item())

    def _chmax(self, mod, inp):
        top = inp[0].detach().topk(2, dim=1).values
        self.margin = min(self.margin, (top[:, 0] - top[:, 1]).min().item())

    def loss(self, y, gt):                                   # the L1 loss's own kink, |y - gt| = 0
        self.margin = min(self.margin, (y.detach() - gt).abs().min().item())

    def close(self):
        for h in self.hooks:
            h.remove()
        return self.margin


def pick_input_seed(make_model, shape, run, first=51, tries=40, min_ok=1e-6):
    """the input seed among first, first+100, ... whose forward stays farthest from every kink"""
    best = (-1.0, None)
    for t in range(tries):
        seed = first + 100 * t
        m = make_model()
        x, gt = O.synthetic_batch(seed, shape), O.synthetic_batch(seed + 1, shape)
        km = KinkMargin(m)
        with torch.no_grad():
            km.loss(run(m, x), gt)
        best = max(best, (km.close(), seed))
        if best[0] >= KinkMargin.MIN_MARGIN:
            break
    print(f"  input seed {best[1]}: kink margin {best[0]:.2e}")
    assert best[0] >= min_ok, "no input seed with a usable kink margin"
    return best[1], best[0]


def adversarial_pixels() -> torch.Tensor:
    """(1,3,1,N) image of hand-picked pixels: gray, black, white, primaries, all tie patterns."""
    px = [
        (0, 0, 0), (1, 1, 1), (.5, .5, .5), (1, 0, 0), (0, 1, 0), (0, 0, 1),
        (.8, .8, .2), (.2, .8, .8), (.8, .2, .8),          # R=G>B, G=B>R, R=B>G
        (.2, .2, .8), (.8, .2, .2), (.2, .8, .2),          # two-way min ties
        (1, 1, 0), (0, 1, 1), (1, 0, 1),
        (.3, .30000001, .3), (1e-9, 0, 0), (0, 1e-9, 0), (0, 0, 1e-9),
        (.7, .1, .10000001), (.7, .10000001, .1),          # tiny +-(g-b): exercises the %6 wrap
        (1e-4, 2e-4, 3e-4), (.999, .998, .997), (3 / 255, 3 / 255, 4 / 255),
        (254 / 255, 1, 254 / 255), (.25, .5, .75), (.75, .5, .25), (.5, .75, .25),
    ]
    t = torch.tensor(px, dtype=torch.float32).t().reshape(1, 3, 1, len(px))
    return t.contiguous()


def adversarial_hvi() -> torch.Tensor:
    """(1,3,1,N) HVI pixels for PHVIT: clamps on every side, the hi==6 black case, zeros."""
    px = [
        (1, -2e-8, .5),        # PHVIT -> (0,0,0)
        (0, 0, 0), (0, 0, 1), (0, 0, .5), (1, 0, 1), (-1, 0, 1), (0, 1, 1), (0, -1, 1),
        (2, 2, 2), (-2, -2, -1), (.3, .4, 1.5), (.3, .4, -.5), (.5, -.5, .7), (-.5, .5, .3),
        (.05, .05, .01), (.9, .9, .99), (1e-9, 1e-9, .5), (-1e-9, -3e-8, .5), (.5, -1.5e-8, .25),
        (.25, -1e-8, .8), (.1, .2, .3), (-.1, -.2, .3), (.6, -.01, .4), (-.6, .01, .4),
    ]
    return torch.tensor(px, dtype=torch.float32).t().reshape(1, 3, 1, len(px)).contiguous()


def check_equal(a, b, what, exact=True, tol=0.0):
    a, b = a.detach(), b.detach()
    if exact:
        same = torch.equal(a, b) or bool(((a == b) | (a.isnan() & b.isnan())).all())
        assert same, f"{what}: oracle != reference (max abs diff {(a - b).abs().max().item():.3e})"
    else:
        d = (a - b).abs().max().item()
        ref = b.abs().max().item()
        assert d <= tol * max(ref, 1e-30) + 1e-12, f"{what}: diff {d:.3e} vs max {ref:.3e}"
    print(f"  ok  {what}")


def gen_hvi():
    ref = RefHVI()
    imgs = {
        "rand": O.synthetic_batch(1, (2, 3, 24, 40)),
        "quant": O.synthetic_batch(2, (2, 3, 24, 40), quantised=True),
        "adv": adversarial_pixels(),
    }
    out = {}
    for kval in (0.2, 0.37):
        with torch.no_grad():
            ref.density_k.fill_(kval)
        k = ref.density_k.detach().clone()
        for name, img in imgs.items():
            x_ref = img.clone().requires_grad_(True)
            y_ref = ref.HVIT(x_ref)
            gy = O.synthetic_batch(7, tuple(y_ref.shape)) - 0.5
            ref.density_k.grad = None
            y_ref.backward(gy)
            x_o = img.clone().requires_grad_(True)
            k_o = k.clone().requires_grad_(True)
            y_o = O.hvit(x_o, k_o)
            y_o.backward(gy)
            tag = f"{name}_k{kval}"
            check_equal(y_o, y_ref, f"HVIT fwd {tag}")
            check_equal(x_o.grad, x_ref.grad, f"HVIT d/dimg {tag}", exact=False, tol=1e-6)
            check_equal(k_o.grad, ref.density_k.grad, f"HVIT d/dk {tag}", exact=False, tol=1e-5)
            assert abs(ref.this_k - kval) < 1e-6
            out[f"hvit_{tag}_in"] = img.numpy()
            out[f"hvit_{tag}_out"] = y_ref.detach().numpy()
            out[f"hvit_{tag}_gout"] = gy.numpy()
            out[f"hvit_{tag}_gin"] = x_ref.grad.numpy()
            out[f"hvit_{tag}_gk"] = ref.density_k.grad.numpy()
            out[f"hvit_{tag}_code"] = O.hvit_branch_code(img).numpy()
            # PHVIT on the HVIT output (round trip) with the same k (this_k side effect)
            z_ref_in = y_ref.detach().clone().requires_grad_(True)
            z_ref = ref.PHVIT(z_ref_in)
            gz = O.synthetic_batch(8, tuple(z_ref.shape)) - 0.5
            z_ref.backward(gz)
            z_o_in = y_ref.detach().clone().requires_grad_(True)
            z_o = O.phvit(z_o_in, ref.this_k)
            z_o.backward(gz)
            check_equal(z_o, z_ref, f"PHVIT(HVIT) fwd {tag}")
            check_equal(z_o_in.grad, z_ref_in.grad, f"PHVIT(HVIT) bwd {tag}", exact=False, tol=1e-6)
            out[f"phvit_rt_{tag}_out"] = z_ref.detach().numpy()
            out[f"phvit_rt_{tag}_gout"] = gz.numpy()
            out[f"phvit_rt_{tag}_gin"] = z_ref_in.grad.numpy()
    # PHVIT on free-standing HVI inputs, all gating modes
    hv_imgs = {
        "rand": (O.synthetic_batch(3, (2, 3, 24, 40)) * 2.6 - 1.3),
        "adv": adversarial_hvi(),
    }
    for kval in (0.0, 0.2):
        for gated, gated2 in ((False, False), (True, True)):
            ref.this_k = kval
            ref.gated, ref.gated2, ref.alpha, ref.alpha_s = gated, gated2, 0.8, 1.3
            for name, hv in hv_imgs.items():
                a = hv.clone().requires_grad_(True)
                y_ref = ref.PHVIT(a)
                gy = O.synthetic_batch(9, tuple(y_ref.shape)) - 0.5
                y_ref.backward(gy)
                b = hv.clone().requires_grad_(True)
                y_o = O.phvit(b, kval, gated, 1.3, gated2, 0.8)
                y_o.backward(gy)
                tag = f"{name}_k{kval}_g{int(gated)}"
                check_equal(y_o, y_ref, f"PHVIT fwd {tag}")
                check_equal(b.grad, a.grad, f"PHVIT bwd {tag}", exact=False, tol=1e-6)
                out[f"phvit_{tag}_in"] = hv.numpy()
                out[f"phvit_{tag}_out"] = y_ref.detach().numpy()
                out[f"phvit_{tag}_gout"] = gy.numpy()
                out[f"phvit_{tag}_gin"] = a.grad.numpy()
                out[f"phvit_{tag}_hi"] = O.phvit_sextant(hv, kval).numpy()
    ref.gated = ref.gated2 = False
    # known-answer facts
    fresh = RefHVI()
    assert fresh.this_k == 0
    ka = fresh.HVIT(torch.tensor([.8, .8, .2]).reshape(1, 3, 1, 1)).flatten().tolist()
    assert abs(ka[0] - 0.3712552) < 1e-6 and abs(ka[1] - 0.6430328) < 1e-6 and abs(ka[2] - .8) < 1e-7, ka
    blk = fresh.PHVIT(torch.tensor([1., -2e-8, .5]).reshape(1, 3, 1, 1)).flatten().tolist()
    assert blk == [0.0, 0.0, 0.0], blk
    np.savez_compressed(os.path.join(GOLD, "hvi_transform.npz"), **out)
    print("wrote hvi_transform.npz", len(out), "arrays")


def load_into(module: torch.nn.Module, p, prefix=""):
    sd = {k: p[prefix + k] for k in module.state_dict().keys()}
    module.load_state_dict(sd, strict=True)


def gen_blocks():
    """Per-block fixtures at reduced and full channel width (c/head = 18 in the full-width case)."""
    out = {}
    cases = [("w12", (12, 12, 24, 48), 2, (2, 20, 28)), ("w36", (36, 36, 72, 144), 2, (1, 16, 24))]
    for tag, chans, heads, (B, H, W) in cases:
        p = O.make_params(11, channels=chans)
        C = chans[1]
        # LayerNorm
        ln = RefLN(C)
        load_into(ln, p, "I_LCA1.norm.")
        x = (O.synthetic_batch(21, (B, C, H, W)) - 0.5) * 3
        xr = x.clone().requires_grad_(True)
        yr = ln(xr)
        gy = O.synthetic_batch(22, tuple(yr.shape)) - 0.5
        yr.backward(gy)
        xo = x.clone().requires_grad_(True)
        po = O.params_to(p, requires_grad=True)
        yo = O.layernorm_cf(xo, po["I_LCA1.norm.weight"], po["I_LCA1.norm.bias"])
        yo.backward(gy)
        check_equal(yo, yr, f"LayerNorm fwd {tag}")
        check_equal(xo.grad, xr.grad, f"LayerNorm dx {tag}", exact=False, tol=1e-6)
        out.update({f"ln_{tag}_x": x.numpy(), f"ln_{tag}_y": yr.detach().numpy(), f"ln_{tag}_gy": gy.numpy(),
                    f"ln_{tag}_gx": xr.grad.numpy(), f"ln_{tag}_gw": ln.weight.grad.numpy(),
                    f"ln_{tag}_gb": ln.bias.grad.numpy()})
        # LCA blocks
        for kind, Ref, fn in (("i_lca", RefILCA, O.i_lca), ("hv_lca", RefHVLCA, O.hv_lca)):
            pre = "I_LCA1" if kind == "i_lca" else "HV_LCA1"
            m = Ref(C, heads)
            load_into(m, p, pre + ".")
            x = O.synthetic_batch(31, (B, C, H, W)) - 0.5
            y = O.synthetic_batch(32, (B, C, H, W)) - 0.5
            xr, yr_ = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
            zr = m(xr, yr_)
            gz = O.synthetic_batch(33, tuple(zr.shape)) - 0.5
            zr.backward(gz)
            po = O.params_to(p, requires_grad=True)
            xo, yo_ = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
            zo = fn(xo, yo_, po, pre, heads)
            zo.backward(gz)
            check_equal(zo, zr, f"{kind} fwd {tag}")
            check_equal(xo.grad, xr.grad, f"{kind} dx {tag}", exact=False, tol=1e-5)
            check_equal(yo_.grad, yr_.grad, f"{kind} dy {tag}", exact=False, tol=1e-5)
            out.update({f"{kind}_{tag}_x": x.numpy(), f"{kind}_{tag}_y": y.numpy(),
                        f"{kind}_{tag}_out": zr.detach().numpy(), f"{kind}_{tag}_gout": gz.numpy(),
                        f"{kind}_{tag}_gx": xr.grad.numpy(), f"{kind}_{tag}_gy": yr_.grad.numpy()})
            for n, prm in m.named_parameters():
                check_equal(po[pre + "." + n].grad, prm.grad, f"{kind} d{n} {tag}", exact=False, tol=2e-5)
                out[f"{kind}_{tag}_g.{n}"] = prm.grad.numpy()
        # down / up blocks (odd output size exercises floor(in/2) and the align_corners ratio)
        dn = RefDown(chans[1], chans[2])
        load_into(dn, p, "IE_block2.")
        x = O.synthetic_batch(41, (B, chans[1], H + 2, W + 2)) - 0.5
        xr = x.clone().requires_grad_(True)
        yr = dn(xr)
        gy = O.synthetic_batch(42, tuple(yr.shape)) - 0.5
        yr.backward(gy)
        po = O.params_to(p, requires_grad=True)
        xo = x.clone().requires_grad_(True)
        yo = O.norm_downsample(xo, po, "IE_block2")
        yo.backward(gy)
        check_equal(yo, yr, f"down fwd {tag}")
        check_equal(xo.grad, xr.grad, f"down dx {tag}", exact=False, tol=1e-5)
        out.update({f"down_{tag}_x": x.numpy(), f"down_{tag}_out": yr.detach().numpy(), f"down_{tag}_gout": gy.numpy(),
                    f"down_{tag}_gx": xr.grad.numpy(), f"down_{tag}_g.prelu.weight": dn.prelu.weight.grad.numpy(),
                    f"down_{tag}_g.down.0.weight": dn.down[0].weight.grad.numpy()})
        up = RefUp(chans[2], chans[1])
        load_into(up, p, "ID_block2.")
        x = O.synthetic_batch(43, (B, chans[2], H // 2, W // 2)) - 0.5
        sk = O.synthetic_batch(44, (B, chans[1], H // 2 * 2, W // 2 * 2)) - 0.5
        xr, sr = x.clone().requires_grad_(True), sk.clone().requires_grad_(True)
        yr = up(xr, sr)
        gy = O.synthetic_batch(45, tuple(yr.shape)) - 0.5
        yr.backward(gy)
        po = O.params_to(p, requires_grad=True)
        xo, so = x.clone().requires_grad_(True), sk.clone().requires_grad_(True)
        yo = O.norm_upsample(xo, so, po, "ID_block2")
        yo.backward(gy)
        check_equal(yo, yr, f"up fwd {tag}")
        check_equal(xo.grad, xr.grad, f"up dx {tag}", exact=False, tol=1e-5)
        check_equal(so.grad, sr.grad, f"up dskip {tag}", exact=False, tol=1e-5)
        out.update({f"up_{tag}_x": x.numpy(), f"up_{tag}_skip": sk.numpy(), f"up_{tag}_out": yr.detach().numpy(),
                    f"up_{tag}_gout": gy.numpy(), f"up_{tag}_gx": xr.grad.numpy(), f"up_{tag}_gskip": sr.grad.numpy(),
                    f"up_{tag}_g.prelu.weight": up.prelu.weight.grad.numpy(),
                    f"up_{tag}_g.up_scale.0.weight": up.up_scale[0].weight.grad.numpy(),
                    f"up_{tag}_g.up.weight": up.up.weight.grad.numpy()})
    np.savez_compressed(os.path.join(GOLD, "blocks.npz"), **out)
    print("wrote blocks.npz", len(out), "arrays")


def gen_model():
    """Whole-network fixtures: reduced width with all gradients; full width with output, selected
    gradients and a checksum per gradient tensor; a strided sample of the 1x3x400x600 output."""
    out = {}
    cases = [("w12", (12, 12, 24, 48), (2, 3, 32, 48), False), ("w36", (36, 36, 72, 144), (1, 3, 64, 96), True)]
    for tag, chans, shape, quant in cases:
        p = O.make_params(5, channels=chans)
        m = RefCIDNet(channels=list(chans))
        load_into(m, p)
        assert len(m.state_dict()) == 191
        x = O.synthetic_batch(51, shape, quantised=quant)
        gt = O.synthetic_batch(52, shape)
        yr = m(x)
        loss_r = (yr - gt).abs().mean()
        loss_r.backward()
        po = O.params_to(p, requires_grad=True)
        yo = O.cidnet_forward(po, x)
        loss_o = (yo - gt).abs().mean()
        loss_o.backward()
        check_equal(yo, yr, f"CIDNet fwd {tag}")
        dead = [n for n, prm in m.named_parameters() if prm.grad is None]
        assert all(n.startswith("I_LCA5.") for n in dead) and len(dead) == 13, dead
        out[f"model_{tag}_x"] = x.numpy()
        out[f"model_{tag}_gt"] = gt.numpy()
        out[f"model_{tag}_out"] = yr.detach().numpy()
        out[f"model_{tag}_loss"] = np.float64(loss_r.item())
        worst = 0.0
        for n, prm in m.named_parameters():
            if prm.grad is None:
                assert po[n].grad is None, n
                continue
            g_r, g_o = prm.grad, po[n].grad
            d = (g_r - g_o).abs().max().item() / max(g_r.abs().max().item(), 1e-30)
            worst = max(worst, d)
            if tag == "w12" or n.count(".") <= 1 or "temperature" in n or n.startswith("HV_LCA3") or "norm" in n:
                out[f"model_{tag}_g.{n}"] = g_r.numpy()
            out[f"model_{tag}_gsum.{n}"] = np.array([g_r.double().sum().item(), g_r.double().abs().sum().item()])
        print(f"  ok  CIDNet grads {tag}: worst rel-to-max diff oracle vs reference {worst:.2e}")
        assert worst < 1e-4
    # config-1 plumbing case: 1x3x400x600 forward, strided sample + checksums (default-init style params)
    p = O.make_params(5, jitter=False)
    m = RefCIDNet()
    load_into(m, p)
    x = O.synthetic_batch(61, (1, 3, 400, 600))
    with torch.no_grad():
        yr = m(x)
        yo = O.cidnet_forward(p, x)
    check_equal(yo, yr, "CIDNet fwd 1x3x400x600")
    out["model_c1_out_strided"] = yr[:, :, ::16, ::16].numpy()
    out["model_c1_out_sums"] = np.array([yr.double().sum().item(), yr.double().abs().sum().item(),
                                         (yr.double() ** 2).sum().item()])
    np.savez_compressed(os.path.join(GOLD, "model.npz"), **out)
    print("wrote model.npz", len(out), "arrays")


def gen_mssa():
    """MSSA variant (net/CIDNet_MSSA.py): SpatialAttention block + whole model, reduced width, all gradients."""
    from net.CIDNet_MSSA import CIDNet as RefMSSA, SpatialAttention as RefSA
    out = {}
    sa = RefSA()
    w = O.make_params(7, channels=(12, 12, 24, 48), variant="mssa")["sa_hv3.conv1.weight"] * 4.0
    sa.load_state_dict({"conv1.weight": w})
    x = O.synthetic_batch(71, (2, 12, 20, 28)) - 0.3
    x[0, :, 3, 4] = x[0, 0, 3, 4]                        # a pixel whose channels all tie (arg-max = channel 0)
    xr = x.clone().requires_grad_(True)
    yr = sa(xr)
    gy = O.synthetic_batch(72, tuple(yr.shape)) - 0.5
    yr.backward(gy)
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo = O.spatial_attention(xo, wo)
    yo.backward(gy)
    check_equal(yo, yr, "SpatialAttention fwd")
    check_equal(xo.grad, xr.grad, "SpatialAttention dx", exact=False, tol=1e-6)
    check_equal(wo.grad, sa.conv1.weight.grad, "SpatialAttention dw", exact=False, tol=1e-5)
    out.update(sa_x=x.numpy(), sa_w=w.numpy(), sa_out=yr.detach().numpy(), sa_gout=gy.numpy(), sa_gx=xr.grad.numpy(),
               sa_gw=sa.conv1.weight.grad.numpy())
    chans = (12, 12, 24, 48)
    p = O.make_params(5, channels=chans, variant="mssa")

    def make():
        mm = RefMSSA(channels=list(chans))
        load_into(mm, p)
        return mm
    seed, margin = pick_input_seed(make, (2, 3, 32, 48), lambda mm, xx: mm(xx))
    m = make()
    assert len(m.state_dict()) == 197
    x = O.synthetic_batch(seed, (2, 3, 32, 48))
    gt = O.synthetic_batch(seed + 1, (2, 3, 32, 48))
    out["model_kink_margin"] = np.float64(margin)
    yr = m(x)
    (yr - gt).abs().mean().backward()
    po = O.params_to(p, requires_grad=True)
    yo = O.cidnet_forward(po, x, variant="mssa")
    (yo - gt).abs().mean().backward()
    check_equal(yo, yr, "CIDNet_MSSA fwd")
    out.update(model_x=x.numpy(), model_gt=gt.numpy(), model_out=yr.detach().numpy())
    worst = 0.0
    for n, prm in m.named_parameters():
        assert prm.grad is not None, n                  # I_LCA5 is live in this variant
        d = (prm.grad - po[n].grad).abs().max().item() / max(prm.grad.abs().max().item(), 1e-30)
        worst = max(worst, d)
        out[f"model_g.{n}"] = prm.grad.numpy()
    print(f"  ok  CIDNet_MSSA grads: worst rel-to-max diff oracle vs reference {worst:.2e}")
    assert worst < 1e-4
    np.savez_compressed(os.path.join(GOLD, "mssa.npz"), **out)
    print("wrote mssa.npz", len(out), "arrays")


def gen_tnsm():
    """TNSM variant (net/TNSM.py, net/CIDNet_TNSM.py): one TNSM block + whole model in training mode
    (returns (rgb, fused_noise)), reduced width, all gradients."""
    from net.CIDNet_TNSM import CIDNet_TNSM as RefTNSM
    from net.TNSM import HV_TNSM as RefHVTNSM
    out = {}
    chans = (12, 12, 24, 48)
    p = O.make_params(9, channels=chans, variant="tnsm")
    blk = RefHVTNSM(chans[1], 2)
    load_into(blk, p, "HV_TNSM1.")
    x = O.synthetic_batch(81, (2, chans[1], 20, 28)) - 0.5
    y = O.synthetic_batch(82, (2, chans[1], 20, 28)) - 0.5
    xr, yr_ = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    zr, nr = blk(xr, yr_)
    gz = O.synthetic_batch(83, tuple(zr.shape)) - 0.5
    gn = O.synthetic_batch(84, tuple(nr.shape)) - 0.5
    (zr * gz).sum().add((nr * gn).sum()).backward()
    po = O.params_to(p, requires_grad=True)
    xo, yo_ = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    zo, no = O.tnsm_block(xo, yo_, po, "HV_TNSM1", 2)
    (zo * gz).sum().add((no * gn).sum()).backward()
    check_equal(zo, zr, "TNSM block fwd")
    check_equal(no, nr, "TNSM block noise map")
    check_equal(xo.grad, xr.grad, "TNSM block dx", exact=False, tol=1e-5)
    check_equal(yo_.grad, yr_.grad, "TNSM block dy", exact=False, tol=1e-5)
    out.update(blk_x=x.numpy(), blk_y=y.numpy(), blk_out=zr.detach().numpy(), blk_noise=nr.detach().numpy(),
               blk_gout=gz.numpy(), blk_gnoise=gn.numpy(), blk_gx=xr.grad.numpy(), blk_gy=yr_.grad.numpy())
    for n, prm in blk.named_parameters():
        check_equal(po["HV_TNSM1." + n].grad, prm.grad, f"TNSM block d{n}", exact=False, tol=5e-5)
        out[f"blk_g.{n}"] = prm.grad.numpy()
    p = O.make_params(5, channels=chans, variant="tnsm")
    m = RefTNSM(channels=list(chans))
    load_into(m, p)
    assert len(m.state_dict()) == 468
    m.train()
    x = O.synthetic_batch(51, (2, 3, 32, 48))
    gt = O.synthetic_batch(52, (2, 3, 32, 48))
    yr, fr = m(x)
    ((yr - gt).abs().mean() + 0.1 * fr.mean()).backward()
    po = O.params_to(p, requires_grad=True)
    yo, fo = O.cidnet_tnsm_forward(po, x)
    ((yo - gt).abs().mean() + 0.1 * fo.mean()).backward()
    check_equal(yo, yr, "CIDNet_TNSM fwd rgb")
    check_equal(fo, fr, "CIDNet_TNSM fwd fused noise")
    m.eval()
    with torch.no_grad():
        ye, fe = m(x)
    assert fe is None
    check_equal(O.cidnet_tnsm_forward(p, x, training=False)[0], ye, "CIDNet_TNSM eval fwd")
    out.update(model_x=x.numpy(), model_gt=gt.numpy(), model_out=yr.detach().numpy(), model_noise=fr.detach().numpy())
    worst, dead = 0.0, []
    for n, prm in m.named_parameters():
        if prm.grad is None:
            dead.append(n)
            assert po[n].grad is None, n
            continue
        d = (prm.grad - po[n].grad).abs().max().item() / max(prm.grad.abs().max().item(), 1e-30)
        worst = max(worst, d)
        out[f"model_g.{n}"] = prm.grad.numpy()
    assert all(n.startswith(("I_LCA5.", "I_TNSM5.")) for n in dead), dead
    out["model_dead"] = np.array(dead)
    # fp64 ground truth of the same gradients: TNSM's un-normalised attention saturates its softmax, so the
    # reference's own fp32 gradients are ~2e-3 (relative to each tensor's max) away from the exact ones;
    # the GPU parity test measures its error against fp64 and compares it with the reference's own error
    p64 = O.params_to(p, dtype=torch.float64, requires_grad=True)
    y64, f64 = O.cidnet_tnsm_forward(p64, x.double())
    ((y64 - gt.double()).abs().mean() + 0.1 * f64.mean()).backward()
    for n, v in p64.items():
        if v.grad is not None:
            out[f"model_g64.{n}"] = v.grad.numpy().astype(np.float32)
    print(f"  ok  CIDNet_TNSM grads: worst rel-to-max diff {worst:.2e}; {len(dead)} dead tensors (I_LCA5.*, I_TNSM5.*)")
    assert worst < 1e-4
    np.savez_compressed(os.path.join(GOLD, "tnsm.npz"), **out)
    print("wrote tnsm.npz", len(out), "arrays")


def gen_losses():
    """SSIM training loss ("next" row f1).  loss.loss_utils (map_ssim, create_window) imports here; loss.losses does not
    (it imports torchvision for the VGG loss), so the SSIM class's one line of arithmetic around map_ssim,
    (1 - map_ssim(...)) * weight (loss/losses.py:189), is applied to the reference's map_ssim output."""
    from loss.loss_utils import map_ssim, create_window
    out = {}
    for tag, shape in (("a", (2, 3, 40, 52)), ("b", (1, 3, 33, 71)), ("c", (2, 1, 12, 9))):
        x = O.synthetic_batch(101, shape)
        y = (0.7 * x + 0.3 * O.synthetic_batch(102, shape)).clamp(0, 1)
        if tag == "b":
            y[..., :9, :] = 0.0                                  # a flat black band: sigma terms vanish there
        c = shape[1]
        for weight in (1.0, 0.5):
            xr = x.clone().requires_grad_(True)
            lr = (1.0 - map_ssim(xr, y, create_window(11, c), 11, c, True)) * weight
            lr.backward()
            xo = x.clone().requires_grad_(True)
            lo = O.ssim_loss(xo, y, weight)
            lo.backward()
            check_equal(lo, lr, f"SSIM loss {tag} w={weight}", exact=False, tol=1e-7)
            check_equal(xo.grad, xr.grad, f"SSIM grad {tag} w={weight}", exact=False, tol=1e-6)
            out[f"{tag}_w{weight}_loss"] = np.float64(lr.item())
            out[f"{tag}_w{weight}_grad"] = xr.grad.numpy()
        out[f"{tag}_x"], out[f"{tag}_y"] = x.numpy(), y.numpy()
    np.savez_compressed(os.path.join(GOLD, "losses.npz"), **out)
    print("wrote losses.npz", len(out), "arrays")


def gen_lr_schedule():
    """Learning-rate sequences of the reference's own scheduler classes (data/scheduler.py) as train.py builds them."""
    from data.scheduler import GradualWarmupScheduler, CosineAnnealingRestartLR
    import warnings
    out = {}
    for tag, (lr, n_ep, warm, start, use_warm) in {"default": (1e-4, 1000, 3, 0, True), "short": (2e-4, 20, 3, 0, True),
                                                   "nowarm": (1e-4, 30, 3, 0, False), "resume": (1e-4, 50, 2, 10, True)}.items():
        prm = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([prm], lr=lr)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if use_warm:
                inner = CosineAnnealingRestartLR(optimizer=opt, periods=[n_ep - warm - start], restart_weights=[1], eta_min=1e-7)
                sch = GradualWarmupScheduler(opt, multiplier=1, total_epoch=warm, after_scheduler=inner)
            else:
                sch = CosineAnnealingRestartLR(optimizer=opt, periods=[n_ep - start], restart_weights=[1], eta_min=1e-7)
            seq = [opt.param_groups[0]["lr"]]
            for _ in range(n_ep - start):
                opt.step()
                sch.step()
                seq.append(opt.param_groups[0]["lr"])
        out[tag + "_cfg"] = np.array([lr, n_ep, warm, start, float(use_warm)])
        out[tag + "_lr"] = np.array(seq, dtype=np.float64)
        print(f"  {tag}: lr[0..5] = {seq[:6]}, last = {seq[-1]:.3e}")
    np.savez_compressed(os.path.join(GOLD, "lr_schedule.npz"), **out)
    print("wrote lr_schedule.npz")


def _store_grads(out, tag, named_grads, full_names=(), small=600):
    """fingerprint (sums + strided sample) of every gradient tensor; full tensors for the small ones and
    for `full_names`"""
    for n, g in named_grads:
        sums, sample, stride = O.grad_fingerprint(g, 512)
        out[f"{tag}_gfp.{n}"] = sums.numpy()
        out[f"{tag}_gs.{n}"] = sample.numpy()
        if g.numel() <= small or n in full_names:
            out[f"{tag}_g.{n}"] = g.detach().numpy()


FULL_TENSORS = ("HVE_block1.down.0.weight", "IE_block2.down.0.weight", "HVE_block3.down.0.weight",
                "HVD_block3.up_scale.0.weight", "ID_block2.up.weight", "HVD_block1.up_scale.0.weight",
                "HV_LCA1.gdfn.project_in.weight", "I_LCA1.gdfn.project_out.weight", "HV_LCA2.ffn.q.weight",
                "I_LCA3.ffn.kv.weight", "HV_LCA6.ffn.project_out.weight", "I_LCA6.gdfn.dwconv.weight",
                "HV_LCA4.gdfn.dwconv1.weight", "I_LCA2.ffn.kv_dwconv.weight", "HVE_block0.1.weight", "ID_block0.1.weight")


def gen_fullsize():
    """BASELINE.json sizes through the imported reference:
    (a) 1x3x400x600 forward + backward, full width, jittered parameters: output, loss, d(loss)/d(input) and a
        fingerprint (sums + strided sample) of EVERY live gradient tensor, full tensors for one member of each
        kernel family -- the benchmark's own tile shapes / multi-round grids / 75- and 150-pixel rows;
    (b) 1x3x1024x1024 forward (configs[3] image size): strided output, checksums, count of black pixels;
    (c) full-width (36/36/72/144, c/head = 18) MSSA and TNSM at 1x3x64x96, all gradients."""
    out = {}
    # ---- (a) ----
    p = O.make_params(5)
    m = RefCIDNet()
    load_into(m, p)
    shape = (1, 3, 400, 600)
    x = O.synthetic_batch(161, shape).requires_grad_(True)
    gt = O.synthetic_batch(162, shape)
    km = KinkMargin(m)
    yr = m(x)
    km.loss(yr, gt)
    loss = (yr - gt).abs().mean()
    loss.backward()
    print(f"  400x600 fwd+bwd through the reference done; kink margin {km.close():.2e}")
    po = O.params_to(p, requires_grad=True)
    xo = x.detach().clone().requires_grad_(True)
    yo = O.cidnet_forward(po, xo)
    (yo - gt).abs().mean().backward()
    check_equal(yo, yr, "CIDNet fwd 1x3x400x600 (jittered params)")
    check_equal(xo.grad, x.grad, "CIDNet d/dx 400x600", exact=False, tol=1e-5)
    out["a_out_strided"] = yr.detach()[:, :, ::8, ::8].numpy()
    yd = yr.detach().double()
    out["a_out_sums"] = np.array([yd.sum().item(), yd.abs().sum().item(), (yd ** 2).sum().item()])
    out["a_loss"] = np.float64(loss.item())
    out["a_gx_strided"] = x.grad[:, :, ::8, ::8].numpy()
    out["a_gx_fp"] = O.grad_fingerprint(x.grad)[0].numpy()
    live = [(n, prm.grad) for n, prm in m.named_parameters() if prm.grad is not None]
    assert len(live) == 191 - 13
    worst = 0.0
    for n, g in live:
        worst = max(worst, (g - po[n].grad).abs().max().item() / max(g.abs().max().item(), 1e-30))
    print(f"  ok  400x600 grads: worst rel-to-max diff oracle vs reference {worst:.2e}")
    assert worst < 1e-4
    _store_grads(out, "a", live, FULL_TENSORS)
    # fp64 truth of the same gradients: at this size the reference's own fp32 gradients sit 4e-4 .. 5e-3 (of each
    # tensor's max; 0.2 for a nearly-cancelling PReLU slope) from it, so the GPU bar is "no further from fp64 than
    # twice the reference's own distance", as for TNSM
    p64 = O.params_to(p, dtype=torch.float64, requires_grad=True)
    x64 = x.detach().double().requires_grad_(True)
    (O.cidnet_forward(p64, x64) - gt.double()).abs().mean().backward()
    _store_grads(out, "a64", [(n, v.grad.float()) for n, v in p64.items() if v.grad is not None], FULL_TENSORS)
    out["a64_gx_strided"] = x64.grad[:, :, ::8, ::8].float().numpy()
    out["a64_gx_fp"] = O.grad_fingerprint(x64.grad)[0].numpy()
    # ---- (b) ----
    p1 = O.make_params(5, jitter=False)
    m = RefCIDNet()
    load_into(m, p1)
    x4 = O.synthetic_batch(171, (1, 3, 1024, 1024), quantised=True)
    with torch.no_grad():
        y4 = m(x4)
        check_equal(O.cidnet_forward(p1, x4), y4, "CIDNet fwd 1x3x1024x1024")
    out["b_out_strided"] = y4[:, :, ::32, ::32].numpy()
    yd = y4.double()
    out["b_out_sums"] = np.array([yd.sum().item(), yd.abs().sum().item(), (yd ** 2).sum().item()])
    out["b_n_black"] = np.int64(((y4 == 0).all(1)).sum().item())
    print(f"  1024x1024: {int(out['b_n_black'])} black pixels")
    # ---- (c) ----
    from net.CIDNet_MSSA import CIDNet as RefMSSA
    from net.CIDNet_TNSM import CIDNet_TNSM as RefTNSM
    shp = (1, 3, 64, 96)
    pm = O.make_params(5, variant="mssa")

    def make():
        mm = RefMSSA()
        load_into(mm, pm)
        return mm
    # ~1.5M PReLU / channel-max arguments at full width: margins of a few 1e-7 are the best any seed offers
    seed, margin = pick_input_seed(make, shp, lambda mm, xx: mm(xx), tries=12, min_ok=1e-7)
    m = make()
    x, gt = O.synthetic_batch(seed, shp), O.synthetic_batch(seed + 1, shp)
    yr = m(x)
    (yr - gt).abs().mean().backward()
    po = O.params_to(pm, requires_grad=True)
    yo = O.cidnet_forward(po, x, variant="mssa")
    (yo - gt).abs().mean().backward()
    check_equal(yo, yr, "CIDNet_MSSA full width fwd")
    out.update(mssa_x=x.numpy(), mssa_gt=gt.numpy(), mssa_out=yr.detach().numpy(), mssa_kink_margin=np.float64(margin))
    g = [(n, prm.grad) for n, prm in m.named_parameters()]
    assert all(t is not None for _, t in g) and len(g) == 197
    worst = max((t - po[n].grad).abs().max().item() / max(t.abs().max().item(), 1e-30) for n, t in g)
    print(f"  ok  full-width MSSA grads: worst rel-to-max diff oracle vs reference {worst:.2e}")
    assert worst < 1e-4
    _store_grads(out, "mssa", g)
    pt = O.make_params(5, variant="tnsm")
    m = RefTNSM()
    load_into(m, pt)
    m.train()
    x, gt = O.synthetic_batch(51, shp), O.synthetic_batch(52, shp)
    yr, fr = m(x)
    ((yr - gt).abs().mean() + 0.1 * fr.mean()).backward()
    po = O.params_to(pt, requires_grad=True)
    yo, fo = O.cidnet_tnsm_forward(po, x)
    ((yo - gt).abs().mean() + 0.1 * fo.mean()).backward()
    check_equal(yo, yr, "CIDNet_TNSM full width fwd rgb")
    check_equal(fo, fr, "CIDNet_TNSM full width fused noise")
    out.update(tnsm_x=x.numpy(), tnsm_gt=gt.numpy(), tnsm_out=yr.detach().numpy(), tnsm_noise=fr.detach().numpy())
    g = [(n, prm.grad) for n, prm in m.named_parameters() if prm.grad is not None]
    dead = [n for n, prm in m.named_parameters() if prm.grad is None]
    assert all(n.startswith(("I_LCA5.", "I_TNSM5.")) for n in dead), dead
    out["tnsm_dead"] = np.array(dead)
    _store_grads(out, "tnsm", g)
    p64 = O.params_to(pt, dtype=torch.float64, requires_grad=True)
    y64, f64 = O.cidnet_tnsm_forward(p64, x.double())
    ((y64 - gt.double()).abs().mean() + 0.1 * f64.mean()).backward()
    _store_grads(out, "tnsm64", [(n, v.grad.float()) for n, v in p64.items() if v.grad is not None])
    np.savez_compressed(os.path.join(GOLD, "fullsize.npz"), **out)
    print("wrote fullsize.npz", len(out), "arrays")


def gen_round3():
    """Round-3 fixtures, written to tests/golden/round3.npz:
    (d) CIDNet(norm=True) -- the LayerNorm option of every down / up block (net/CIDNet.py:12, net/transformer_utils.py:
        44-48, 66-70) -- reduced width, forward + every gradient through the imported reference;
    (c) BASELINE configs[4] at its image size: 1x3x400x600 forward of CIDNet_MSSA and CIDNet_TNSM (train mode: rgb and the
        fused noise map) through the imported reference, strided outputs + checksums; for TNSM also the fp64 oracle's
        output (its un-normalised attention saturates the softmax, net/TNSM.py:98-104)."""
    out = {}
    # ---- (d) ----
    chans = (12, 12, 24, 48)
    p = O.make_params(13, channels=chans, norm=True)

    def make():
        mm = RefCIDNet(channels=list(chans), norm=True)
        load_into(mm, p)
        return mm
    shp = (2, 3, 32, 48)
    seed, margin = pick_input_seed(make, shp, lambda mm, xx: mm(xx))
    m = make()
    assert len(m.state_dict()) == 191 + 24
    x, gt = O.synthetic_batch(seed, shp), O.synthetic_batch(seed + 1, shp)
    yr = m(x)
    (yr - gt).abs().mean().backward()
    po = O.params_to(p, requires_grad=True)
    yo = O.cidnet_forward(po, x)
    (yo - gt).abs().mean().backward()
    check_equal(yo, yr, "CIDNet(norm=True) fwd")
    out.update(norm_x=x.numpy(), norm_gt=gt.numpy(), norm_out=yr.detach().numpy(), norm_kink_margin=np.float64(margin))
    worst, dead = 0.0, []
    for n, prm in m.named_parameters():
        if prm.grad is None:
            dead.append(n)
            assert po[n].grad is None, n
            continue
        worst = max(worst, (prm.grad - po[n].grad).abs().max().item() / max(prm.grad.abs().max().item(), 1e-30))
        out[f"norm_g.{n}"] = prm.grad.numpy()
    assert all(n.startswith("I_LCA5.") for n in dead), dead
    print(f"  ok  CIDNet(norm=True) grads: worst rel-to-max diff oracle vs reference {worst:.2e}; kink margin {margin:.2e}")
    assert worst < 1e-4
    # ---- (c) ----
    from net.CIDNet_MSSA import CIDNet as RefMSSA
    from net.CIDNet_TNSM import CIDNet_TNSM as RefTNSM
    shape = (1, 3, 400, 600)
    x = O.synthetic_batch(181, shape)
    pm = O.make_params(5, variant="mssa")
    m = RefMSSA()
    load_into(m, pm)
    with torch.no_grad():
        ym = m(x)
        check_equal(O.cidnet_forward(pm, x, variant="mssa"), ym, "CIDNet_MSSA fwd 1x3x400x600")
    out["mssa400_out_strided"] = ym[:, :, ::8, ::8].numpy()
    yd = ym.double()
    out["mssa400_out_sums"] = np.array([yd.sum().item(), yd.abs().sum().item(), (yd ** 2).sum().item()])
    pt = O.make_params(5, variant="tnsm")
    m = RefTNSM()
    load_into(m, pt)
    m.train()
    with torch.no_grad():
        yt, ft = m(x)
        yo, fo = O.cidnet_tnsm_forward(pt, x)
        check_equal(yo, yt, "CIDNet_TNSM fwd rgb 1x3x400x600")
        check_equal(fo, ft, "CIDNet_TNSM fwd fused noise 1x3x400x600")
        y64, f64 = O.cidnet_tnsm_forward(O.params_to(pt, dtype=torch.float64), x.double())
    out["tnsm400_out_strided"] = yt[:, :, ::8, ::8].numpy()
    out["tnsm400_noise_strided"] = ft[:, :, ::8, ::8].numpy()
    out["tnsm400_out64_strided"] = y64[:, :, ::8, ::8].float().numpy()
    out["tnsm400_noise64_strided"] = f64[:, :, ::8, ::8].float().numpy()
    yd = yt.double()
    out["tnsm400_out_sums"] = np.array([yd.sum().item(), yd.abs().sum().item(), (yd ** 2).sum().item()])
    print(f"  TNSM 400x600: reference fp32 vs fp64 oracle: rgb {(yt.double() - y64).abs().max().item():.2e}, "
          f"noise {(ft.double() - f64).abs().max().item():.2e}")
    np.savez_compressed(os.path.join(GOLD, "round3.npz"), **out)
    print("wrote round3.npz", len(out), "arrays")


def gen_round4():
    """Round-4 fixtures, written to tests/golden/round4.npz: BASELINE configs[4] at its image size WITH
    the backward -- CIDNet_MSSA and CIDNet_TNSM (train mode, loss = L1 + 0.1 mean(fused noise) so that the noise branch
    trains) forward + backward on 1x3x400x600 through the imported reference: loss, d(loss)/d(input) and a fingerprint
    (sums + strided sample; small tensors whole) of every live gradient tensor, next to the same from an fp64 evaluation
    of the oracle (at this size the reference's own fp32 gradients sit 1e-4 .. 1e-2 of a tensor's max from it)."""
    from net.CIDNet_MSSA import CIDNet as RefMSSA
    from net.CIDNet_TNSM import CIDNet_TNSM as RefTNSM
    out = {}
    shape = (1, 3, 400, 600)
    x0 = O.synthetic_batch(191, shape)
    gt = O.synthetic_batch(192, shape)
    for tag, variant, Ref in (("mssa400", "mssa", RefMSSA), ("tnsm400", "tnsm", RefTNSM)):
        p = O.make_params(5, variant=variant)
        m = Ref()
        load_into(m, p)
        m.train()
        x = x0.clone().requires_grad_(True)

        def objective(res, target):
            if variant == "tnsm":
                return (res[0] - target).abs().mean() + 0.1 * res[1].mean()
            return (res - target).abs().mean()
        res = m(x)
        loss = objective(res, gt)
        loss.backward()
        po = O.params_to(p, requires_grad=True)
        xo = x0.clone().requires_grad_(True)
        reso = O.cidnet_tnsm_forward(po, xo) if variant == "tnsm" else O.cidnet_forward(po, xo, variant="mssa")
        objective(reso, gt).backward()
        yr = res[0] if variant == "tnsm" else res
        check_equal(reso[0] if variant == "tnsm" else reso, yr, f"{tag} fwd")
        live = [(n, prm.grad) for n, prm in m.named_parameters() if prm.grad is not None]
        dead = [n for n, prm in m.named_parameters() if prm.grad is None]
        assert all(n.startswith(("I_LCA5.", "I_TNSM5.")) for n in dead), dead
        worst = max((g - po[n].grad).abs().max().item() / max(g.abs().max().item(), 1e-30) for n, g in live)
        print(f"  ok  {tag} fwd+bwd through the reference: {len(live)} live gradient tensors, {len(dead)} dead; "
              f"worst rel-to-max diff oracle vs reference {worst:.2e}")
        assert worst < 1e-3
        out[f"{tag}_loss"] = np.float64(loss.item())
        out[f"{tag}_out_strided"] = yr.detach()[:, :, ::8, ::8].numpy()
        out[f"{tag}_gx_strided"] = x.grad[:, :, ::8, ::8].numpy()
        out[f"{tag}_gx_fp"] = O.grad_fingerprint(x.grad)[0].numpy()
        out[f"{tag}_dead"] = np.array(dead)
        _store_grads(out, tag, live)
        del m, res, loss, po, reso
        p64 = O.params_to(p, dtype=torch.float64, requires_grad=True)
        x64 = x0.double().requires_grad_(True)
        r64 = O.cidnet_tnsm_forward(p64, x64) if variant == "tnsm" else O.cidnet_forward(p64, x64, variant="mssa")
        objective(r64, gt.double()).backward()
        _store_grads(out, tag + "64", [(n, v.grad.float()) for n, v in p64.items() if v.grad is not None])
        out[f"{tag}64_gx_strided"] = x64.grad[:, :, ::8, ::8].float().numpy()
        out[f"{tag}64_gx_fp"] = O.grad_fingerprint(x64.grad)[0].numpy()
        out[f"{tag}64_out_strided"] = (r64[0] if variant == "tnsm" else r64).detach()[:, :, ::8, ::8].float().numpy()
        print(f"  {tag}: fp64 evaluation done")
        del p64, x64, r64
    np.savez_compressed(os.path.join(GOLD, "round4.npz"), **out)
    print("wrote round4.npz", len(out), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round4":
        gen_round4()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round3":
        gen_round3()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mssa":
        gen_mssa()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tnsm":
        gen_tnsm()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "losses":
        gen_losses()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        gen_fullsize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lr":
        gen_lr_schedule()
        sys.exit(0)
    gen_hvi()
    gen_blocks()
    gen_model()
    gen_mssa()
    gen_tnsm()
    gen_losses()
    gen_lr_schedule()
    gen_fullsize()
    gen_round3()
    gen_round4()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")
