"""CPU: the oracle (oracle/cidnet_oracle.py) reproduces the reference's outputs and gradients stored
in tests/golden/*.npz (written by oracle/gen_golden.py from the imported reference)."""
import numpy as np
import pytest
import torch

from oracle import cidnet_oracle as O


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _close(a, b, rel=1e-6, abs_=1e-7):
    a, b = _t(a).double(), _t(b).double()
    tol = rel * b.abs().max().item() + abs_
    assert (a - b).abs().max().item() <= tol, ((a - b).abs().max().item(), tol)


@pytest.mark.parametrize("name", ["rand", "quant", "adv"])
@pytest.mark.parametrize("k", [0.2, 0.37])
def test_hvit_matches_reference(golden, name, k):
    g = golden("hvi_transform")
    tag = f"{name}_k{k}"
    x = _t(g[f"hvit_{tag}_in"]).requires_grad_(True)
    kk = torch.full([1], k, requires_grad=True)
    y = O.hvit(x, kk)
    assert torch.equal(y.detach(), _t(g[f"hvit_{tag}_out"]))          # bit-exact forward on CPU
    y.backward(_t(g[f"hvit_{tag}_gout"]))
    _close(x.grad, g[f"hvit_{tag}_gin"])
    _close(kk.grad, g[f"hvit_{tag}_gk"], rel=1e-5)
    assert np.array_equal(O.hvit_branch_code(x.detach()).numpy(), g[f"hvit_{tag}_code"])
    z = O.phvit(_t(g[f"hvit_{tag}_out"]), float(np.float32(k)))     # this_k = k.item() of the fp32 parameter
    assert torch.equal(z, _t(g[f"phvit_rt_{tag}_out"]))


@pytest.mark.parametrize("name", ["rand", "adv"])
@pytest.mark.parametrize("k", [0.0, 0.2])
@pytest.mark.parametrize("gated", [0, 1])
def test_phvit_matches_reference(golden, name, k, gated):
    g = golden("hvi_transform")
    tag = f"{name}_k{k}_g{gated}"
    x = _t(g[f"phvit_{tag}_in"]).requires_grad_(True)
    y = O.phvit(x, k, bool(gated), 1.3, bool(gated), 0.8)
    assert torch.equal(y.detach(), _t(g[f"phvit_{tag}_out"]))
    y.backward(_t(g[f"phvit_{tag}_gout"]))
    _close(x.grad, g[f"phvit_{tag}_gin"])
    assert np.array_equal(O.phvit_sextant(x.detach(), k).numpy(), g[f"phvit_{tag}_hi"])


def test_known_answers():
    k = torch.full([1], 0.2)
    px = lambda r, g, b: torch.tensor([r, g, b], dtype=torch.float32).reshape(1, 3, 1, 1)
    assert O.hvit(px(0, 0, 0), k).flatten().tolist() == [0, 0, 0]
    assert O.hvit(px(.5, .5, .5), k).flatten().tolist() == [0, 0, .5]
    red = O.hvit(px(1, 0, 0), k).flatten()
    assert torch.allclose(red, torch.tensor([1., 0., 1.]), atol=1e-6)
    ka = O.hvit(px(.8, .8, .2), k).flatten()
    assert torch.allclose(ka, torch.tensor([0.3712552, 0.6430328, 0.8]), atol=1e-6)
    assert O.phvit(px(1., -2e-8, .5), 0.0).flatten().tolist() == [0, 0, 0]      # the hi == 6 black pixel
    assert len(O.param_shapes()) == 191
    assert sum(int(np.prod(s)) for s in O.param_shapes().values()) == 1975569


@pytest.mark.parametrize("tag,chans", [("w12", (12, 12, 24, 48)), ("w36", (36, 36, 72, 144))])
def test_model_matches_reference(golden, tag, chans):
    g = golden("model")
    p = O.params_to(O.make_params(5, channels=chans), requires_grad=True)
    x, gt = _t(g[f"model_{tag}_x"]), _t(g[f"model_{tag}_gt"])
    y = O.cidnet_forward(p, x)
    assert torch.equal(y.detach(), _t(g[f"model_{tag}_out"]))
    loss = (y - gt).abs().mean()
    assert abs(loss.item() - float(g[f"model_{tag}_loss"])) < 1e-7
    loss.backward()
    for n, v in p.items():
        key = f"model_{tag}_gsum.{n}"
        if key not in g.files:
            assert n.startswith("I_LCA5.") and v.grad is None
            continue
        s = g[key]
        assert abs(v.grad.double().sum().item() - s[0]) <= 1e-6 * s[1] + 1e-12, n
        if f"model_{tag}_g.{n}" in g.files:
            _close(v.grad, g[f"model_{tag}_g.{n}"], rel=1e-5)


@pytest.mark.parametrize("kind", ["i_lca", "hv_lca"])
def test_lca_blocks_match_reference(golden, kind):
    g = golden("blocks")
    for tag, chans in (("w12", (12, 12, 24, 48)), ("w36", (36, 36, 72, 144))):
        p = O.make_params(11, channels=chans)
        pre = "I_LCA1" if kind == "i_lca" else "HV_LCA1"
        fn = O.i_lca if kind == "i_lca" else O.hv_lca
        z = fn(_t(g[f"{kind}_{tag}_x"]), _t(g[f"{kind}_{tag}_y"]), p, pre, 2)
        assert torch.equal(z, _t(g[f"{kind}_{tag}_out"]))


def test_mssa_matches_reference(golden):
    """MSSA variant (net/CIDNet_MSSA.py): SpatialAttention block and the whole 197-tensor model."""
    g = golden("mssa")
    x, w = _t(g["sa_x"]).requires_grad_(True), _t(g["sa_w"]).requires_grad_(True)
    y = O.spatial_attention(x, w)
    assert torch.equal(y.detach(), _t(g["sa_out"]))
    y.backward(_t(g["sa_gout"]))
    _close(x.grad, g["sa_gx"])
    _close(w.grad, g["sa_gw"], rel=1e-5)
    p = O.params_to(O.make_params(5, channels=(12, 12, 24, 48), variant="mssa"), requires_grad=True)
    assert len(p) == 197
    out = O.cidnet_forward(p, _t(g["model_x"]), variant="mssa")
    assert torch.equal(out.detach(), _t(g["model_out"]))
    (out - _t(g["model_gt"])).abs().mean().backward()
    for n, v in p.items():
        _close(v.grad, g[f"model_g.{n}"], rel=1e-5)


def test_tnsm_matches_reference(golden):
    """TNSM variant (net/TNSM.py, net/CIDNet_TNSM.py): 468-tensor model, (rgb, fused_noise) in training mode"""
    g = golden("tnsm")
    p = O.params_to(O.make_params(5, channels=(12, 12, 24, 48), variant="tnsm"), requires_grad=True)
    assert len(p) == 468
    rgb, fz = O.cidnet_tnsm_forward(p, _t(g["model_x"]))
    assert torch.equal(rgb.detach(), _t(g["model_out"])) and torch.equal(fz.detach(), _t(g["model_noise"]))
    ((rgb - _t(g["model_gt"])).abs().mean() + 0.1 * fz.mean()).backward()
    dead = set(g["model_dead"].tolist())
    for n, v in p.items():
        if n in dead:
            assert v.grad is None
        else:
            _close(v.grad, g[f"model_g.{n}"], rel=1e-5)


def test_ssim_oracle_matches_reference_fixture(golden):
    """oracle.ssim_loss reproduces the reference's map_ssim-based SSIM loss and its gradient (tests/golden/losses.npz)"""
    import torch
    g = golden("losses")
    for tag in ("a", "b", "c"):
        for weight in (1.0, 0.5):
            x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
            y = torch.from_numpy(g[f"{tag}_y"])
            loss = O.ssim_loss(x, y, weight)
            loss.backward()
            assert abs(loss.item() - float(g[f"{tag}_w{weight}_loss"])) <= 1e-7
            ref = torch.from_numpy(g[f"{tag}_w{weight}_grad"])
            assert (x.grad - ref).abs().max().item() <= 1e-6 * ref.abs().max().item() + 1e-12


@pytest.mark.parametrize("tag", ["default", "short", "nowarm", "resume"])
def test_lr_schedule_matches_reference_sequence(golden, tag):
    """schedule.WarmupCosineLR ("next" row f2) reproduces the learning rates the reference's own scheduler classes set,
    step by step (tests/golden/lr_schedule.npz was written by stepping data/scheduler.py as train.py:165-181 builds it)"""
    from hvi_cidnet_amd.schedule import WarmupCosineLR
    g = golden("lr_schedule")
    lr, n_ep, warm, start, use_warm = g[tag + "_cfg"]
    sch = WarmupCosineLR(lr, int(n_ep), int(warm), int(start), bool(use_warm))
    ref = g[tag + "_lr"]
    got = np.array([sch.lr_after(n) for n in range(len(ref))])
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-18), np.abs(got - ref).max()


def _fp_close(g, tag, name, grad, rel=2e-5):
    sums, sample, _ = O.grad_fingerprint(grad, 512)
    ref = _t(g[f"{tag}_gs.{name}"]).double()
    assert (sample.double() - ref).abs().max().item() <= rel * ref.abs().max().item() + 1e-9, (tag, name)
    fp = g[f"{tag}_gfp.{name}"]
    assert abs(sums[0].item() - fp[0]) <= rel * fp[1] + 1e-9 and abs(sums[2].item() - fp[2]) <= rel * fp[1] + 1e-9, (tag, name)


def test_fullsize_oracle_matches_reference_fixture(golden):
    """the oracle at BASELINE.json's image size (1x3x400x600 forward + backward) and the full-width MSSA / TNSM
    variants reproduce the reference's outputs and gradient fingerprints of tests/golden/fullsize.npz"""
    g = golden("fullsize")
    torch.set_num_threads(8)
    p = O.params_to(O.make_params(5), requires_grad=True)
    x = O.synthetic_batch(161, (1, 3, 400, 600)).requires_grad_(True)
    y = O.cidnet_forward(p, x)
    assert torch.equal(y.detach()[:, :, ::8, ::8], _t(g["a_out_strided"]))
    (y - O.synthetic_batch(162, (1, 3, 400, 600))).abs().mean().backward()
    _close(x.grad[:, :, ::8, ::8], g["a_gx_strided"], rel=1e-5)
    n = 0
    for name, v in p.items():
        if v.grad is not None:
            _fp_close(g, "a", name, v.grad)
            n += 1
    assert n == 191 - 13
    p = O.params_to(O.make_params(5, variant="mssa"), requires_grad=True)
    y = O.cidnet_forward(p, _t(g["mssa_x"]), variant="mssa")
    assert torch.equal(y.detach(), _t(g["mssa_out"]))
    (y - _t(g["mssa_gt"])).abs().mean().backward()
    for name, v in p.items():
        _fp_close(g, "mssa", name, v.grad)
    p = O.params_to(O.make_params(5, variant="tnsm"), requires_grad=True)
    y, fz = O.cidnet_tnsm_forward(p, _t(g["tnsm_x"]))
    assert torch.equal(y.detach(), _t(g["tnsm_out"])) and torch.equal(fz.detach(), _t(g["tnsm_noise"]))
    ((y - _t(g["tnsm_gt"])).abs().mean() + 0.1 * fz.mean()).backward()
    dead = set(g["tnsm_dead"].tolist())
    for name, v in p.items():
        if name not in dead:
            _fp_close(g, "tnsm", name, v.grad)


def test_oracle_norm_option_matches_reference(golden):
    """CIDNet(norm=True) (net/CIDNet.py:12; the LayerNorm of every down / up block, net/transformer_utils.py:44-48,66-70):
    the oracle against the reference's output and gradients in round3.npz"""
    g = golden("round3")
    chans = (12, 12, 24, 48)
    p = O.params_to(O.make_params(13, channels=chans, norm=True), requires_grad=True)
    assert len(p) == 191 + 24
    y = O.cidnet_forward(p, _t(g["norm_x"]))
    assert torch.equal(y.detach(), _t(g["norm_out"]))
    (y - _t(g["norm_gt"])).abs().mean().backward()
    n = 0
    for k, v in p.items():
        if k.startswith("I_LCA5."):
            assert v.grad is None
            continue
        _close(v.grad, g[f"norm_g.{k}"], rel=1e-5)
        n += 1
    assert n == 191 + 24 - 13


def test_oracle_mssa_400x600_forward_matches_reference(golden):
    """BASELINE configs[4] image size: the oracle's CIDNet_MSSA forward at 1x3x400x600 against the reference's"""
    g = golden("round3")
    with torch.no_grad():
        y = O.cidnet_forward(O.make_params(5, variant="mssa"), O.synthetic_batch(181, (1, 3, 400, 600)), variant="mssa")
    assert torch.equal(y[:, :, ::8, ::8], _t(g["mssa400_out_strided"]))
