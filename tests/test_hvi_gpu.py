"""GPU parity of K1/K2 (HVIT / PHVIT HIP kernels) through the C ABI against the golden fixtures
(reference outputs) and against the CPU oracle on seeded inputs.  Tolerances: values 2e-6 abs
(north-star bar is 1e-4), mask / arg-max / sextant decisions bit-exact."""
import numpy as np
import pytest
import torch

from oracle import cidnet_oracle as O

pytestmark = pytest.mark.gpu
VAL_TOL = 2e-6


def _t(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def _maxdiff(a, b):
    return (a.detach().cpu().double() - _t(b).double()).abs().max().item()


def _grad_ok(a, b, rel=1e-4):
    b = _t(b).double()
    d = (a.detach().cpu().double() - b).abs().max().item()
    assert d <= rel * b.abs().max().item() + 1e-7, (d, b.abs().max().item())


@pytest.mark.parametrize("name", ["rand", "quant", "adv"])
@pytest.mark.parametrize("k", [0.2, 0.37])
def test_hvit_golden(golden, dev, name, k):
    from hvi_cidnet_amd import ops
    g = golden("hvi_transform")
    tag = f"{name}_k{k}"
    x = _t(g[f"hvit_{tag}_in"], dev).requires_grad_(True)
    kk = torch.full([1], k, device=dev, requires_grad=True)
    y = ops.HVITFn.apply(x, kk)
    assert _maxdiff(y, g[f"hvit_{tag}_out"]) <= VAL_TOL
    code = ops.hvit_branch_code(x.detach(), kk.detach()).cpu().numpy()
    assert np.array_equal(code, g[f"hvit_{tag}_code"])            # bit-exact index/mask logic
    y.backward(_t(g[f"hvit_{tag}_gout"], dev))
    _grad_ok(x.grad, g[f"hvit_{tag}_gin"])
    _grad_ok(kk.grad, g[f"hvit_{tag}_gk"])
    # round trip through PHVIT with the k snapshot kept on the device
    z_in = _t(g[f"hvit_{tag}_out"], dev).requires_grad_(True)
    z = ops.PHVITFn.apply(z_in, None, None, kk.detach(), False, 1.3, False, 1.0)
    assert _maxdiff(z, g[f"phvit_rt_{tag}_out"]) <= 5e-6
    z.backward(_t(g[f"phvit_rt_{tag}_gout"], dev))
    # HSV->RGB is continuous but not differentiable across sextant boundaries, and quantised images
    # sit exactly on them (hue = n/6): compare gradients where the sextant decision agrees
    kf = float(np.float32(k))
    hi = ops.phvit_sextant(z_in.detach(), kf).cpu()
    same = (hi == O.phvit_sextant(_t(g[f"hvit_{tag}_out"]), kf)).unsqueeze(1).expand(-1, 3, -1, -1)
    assert same.float().mean().item() > (0.85 if name == "adv" else 0.97)   # adv: 28 hand-picked boundary pixels
    ga, gb = z_in.grad.detach().cpu(), _t(g[f"phvit_rt_{tag}_gin"])
    assert (ga - gb).abs()[same].max().item() <= 1e-4 * gb.abs().max().item() + 1e-7


@pytest.mark.parametrize("name", ["rand", "adv"])
@pytest.mark.parametrize("k", [0.0, 0.2])
@pytest.mark.parametrize("gated", [0, 1])
def test_phvit_golden(golden, dev, name, k, gated):
    from hvi_cidnet_amd import ops
    g = golden("hvi_transform")
    tag = f"{name}_k{k}_g{gated}"
    x = _t(g[f"phvit_{tag}_in"], dev).requires_grad_(True)
    y = ops.PHVITFn.apply(x, None, None, k, bool(gated), 1.3, bool(gated), 0.8)
    hi = ops.phvit_sextant(x.detach(), k).cpu().numpy()
    ref_hi = g[f"phvit_{tag}_hi"]
    # a sextant may legitimately flip where atan2 differs by an ulp at a boundary; HSV->RGB is
    # continuous there, EXCEPT the hi==6 black pixel, which must be reproduced exactly
    assert np.array_equal(hi == 6, ref_hi == 6)
    assert (hi != ref_hi).mean() < 1e-3
    assert _maxdiff(y, g[f"phvit_{tag}_out"]) <= 5e-6
    y.backward(_t(g[f"phvit_{tag}_gout"], dev))
    same = torch.from_numpy(hi == ref_hi)
    ga, gb = x.grad.detach().cpu(), _t(g[f"phvit_{tag}_gin"])
    m = same.unsqueeze(1).expand_as(ga)
    d = (ga - gb).abs()[m].max().item()
    assert d <= 1e-4 * gb.abs().max().item() + 1e-7


@pytest.mark.parametrize("shape", [(2, 3, 50, 75), (1, 3, 37, 41), (8, 3, 400, 600)])
def test_hvit_vs_oracle_seeded(dev, shape):
    """odd sizes exercise the scalar tail and the unaligned 16-byte path; 8x3x400x600 is config 2"""
    from hvi_cidnet_amd import ops
    for quant in (False, True):
        x = O.synthetic_batch(123, shape, quantised=quant)
        k = torch.full([1], 0.23)
        y = ops.HVITFn.apply(x.to(dev), k.to(dev)).cpu()
        ref = O.hvit(x, k)
        assert (y - ref).abs().max().item() <= VAL_TOL
        assert torch.equal(ops.hvit_branch_code(x.to(dev), k.to(dev)).cpu(), O.hvit_branch_code(x))
        z = ops.PHVITFn.apply(y.to(dev), None, None, 0.23, False, 1.3, False, 1.0).cpu()
        assert (z - x).abs().max().item() <= 2e-4      # PHVIT o HVIT ~ identity (s = sqrt(eps) floor)
        zr = O.phvit(ref, 0.23)
        assert (z - zr).abs().max().item() <= 1e-5


def test_phvit_residual_fusion(dev):
    from hvi_cidnet_amd import ops
    hvi = (O.synthetic_batch(5, (2, 3, 24, 36)) * 2 - 1)
    hv = (O.synthetic_batch(6, (2, 2, 24, 36)) - .5)
    iv = (O.synthetic_batch(7, (2, 1, 24, 36)) - .5)
    a, b, c = (t.to(dev).requires_grad_(True) for t in (hvi, hv, iv))
    y = ops.PHVITFn.apply(a, b, c, 0.2, False, 1.3, False, 1.0)
    gy = (O.synthetic_batch(8, (2, 3, 24, 36)) - .5)
    y.backward(gy.to(dev))
    ao, bo, co = (t.clone().requires_grad_(True) for t in (hvi, hv, iv))
    yo = O.phvit(torch.cat([bo, co], 1) + ao, 0.2)
    yo.backward(gy)
    assert (y.cpu() - yo).abs().max().item() <= 5e-6
    for u, v in ((a, ao), (b, bo), (c, co)):
        _grad_ok(u.grad, v.grad.numpy())


def test_this_k_semantics(dev):
    from hvi_cidnet_amd.hvi_transform import RGB_HVI
    m = RGB_HVI().to(dev)
    assert m.this_k == 0                                   # fresh module (reference :14)
    with torch.no_grad():
        m.density_k.fill_(0.31)
    m.HVIT(torch.rand(1, 3, 8, 8, device=dev))
    assert abs(m.this_k - 0.31) < 1e-6                     # value of k at the most recent HVIT (:38)
    assert set(m.state_dict().keys()) == {"density_k"}
