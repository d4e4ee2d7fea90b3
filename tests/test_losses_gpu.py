"""GPU parity of the device-side training losses ("next" row f1): SSIM (and L1) against the reference's own values and
gradients in tests/golden/losses.npz (written by oracle/gen_golden.py from loss/loss_utils.map_ssim), against the oracle
on seeded full-size inputs, and the combined RGB + HVI objective against the oracle's composition."""
import numpy as np
import pytest
import torch

from oracle import cidnet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("weight", [1.0, 0.5])
def test_ssim_golden(golden, dev, tag, weight):
    import hvi_cidnet_amd as P
    g = golden("losses")
    x = torch.from_numpy(g[f"{tag}_x"]).to(dev).requires_grad_(True)
    y = torch.from_numpy(g[f"{tag}_y"]).to(dev)
    loss = P.SSIM(weight=weight)(x, y)
    loss.backward()
    ref_l, ref_g = float(g[f"{tag}_w{weight}_loss"]), torch.from_numpy(g[f"{tag}_w{weight}_grad"])
    assert abs(loss.item() - ref_l) <= 2e-6, (loss.item(), ref_l)
    d = (x.grad.cpu() - ref_g).abs().max().item()
    assert d <= 1e-4 * ref_g.abs().max().item() + 1e-9, f"SSIM grad: {d:.3e} vs max {ref_g.abs().max().item():.3e}"


def test_ssim_full_size_vs_oracle(dev):
    """8x3x400x600 (the benchmark's image size): value against the fp64 oracle, gradient on one sample, upstream scale"""
    import hvi_cidnet_amd as P
    x = O.synthetic_batch(111, (8, 3, 400, 600))
    y = (0.8 * x + 0.2 * O.synthetic_batch(112, (8, 3, 400, 600))).clamp(0, 1)
    xd = x.to(dev).requires_grad_(True)
    loss = P.SSIM(weight=0.5)(xd, y.to(dev))
    (3.0 * loss).backward()
    ref = O.ssim_loss(x.double(), y.double(), 0.5)
    assert abs(loss.item() - ref.item()) <= 2e-6
    xs = x[:1].double().requires_grad_(True)
    (3.0 * O.ssim_loss(xs, y[:1].double(), 0.5) / 8.0).backward()      # a sample's share of the batch mean
    d = (xd.grad[:1].cpu().double() - xs.grad).abs().max().item()
    assert d <= 1e-4 * xs.grad.abs().max().item() + 1e-10, d


@pytest.mark.parametrize("shape", [(2, 3, 40, 52), (1, 3, 33, 71), (2, 1, 7, 5), (1, 3, 2, 9)])
def test_edge_loss_vs_oracle(dev, shape):
    """EdgeLoss value and gradient against the fp64 oracle (even / odd sizes: the x4 up-sampling grid and the
    replicate-padding adjoint at all four borders; planes smaller than the 5x5 window)"""
    import hvi_cidnet_amd as P
    x = O.synthetic_batch(131, shape)
    y = (0.6 * x + 0.4 * O.synthetic_batch(132, shape)).clamp(0, 1)
    xd = x.to(dev).requires_grad_(True)
    loss = P.EdgeLoss(loss_weight=50.0)(xd, y.to(dev))
    (0.5 * loss).backward()
    x64 = x.double().requires_grad_(True)
    ref = O.edge_loss(x64, y.double(), 50.0)
    (0.5 * ref).backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-9, (loss.item(), ref.item())
    d = (xd.grad.cpu().double() - x64.grad).abs().max().item()
    assert d <= 1e-4 * x64.grad.abs().max().item() + 1e-12, d


def test_edge_loss_full_size(dev):
    import hvi_cidnet_amd as P
    x = O.synthetic_batch(133, (8, 3, 400, 600))
    y = O.synthetic_batch(134, (8, 3, 400, 600))
    loss = P.EdgeLoss(loss_weight=50.0)(x.to(dev), y.to(dev))
    ref = O.edge_loss(x[:2].double(), y[:2].double(), 50.0)            # the loss is a mean: compare a 2-sample sub-batch
    sub = P.EdgeLoss(loss_weight=50.0)(x[:2].to(dev), y[:2].to(dev))
    assert abs(sub.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert loss.item() > 0


def test_cidnet_loss_composition(dev):
    """loss_rgb + HVI_weight * loss_hvi with L1 + SSIM + Edge terms (train.py:61-65 minus the perceptual one) == oracle composition,
    and its gradient reaches the image through both the RGB terms and the HVIT of the output"""
    import hvi_cidnet_amd as P
    m = P.CIDNet(channels=[12, 12, 24, 48]).to(dev)
    crit = P.CIDNetLoss(m, L1_weight=1.0, D_weight=0.5, E_weight=50.0, HVI_weight=1.0)
    out = O.synthetic_batch(121, (2, 3, 32, 48))
    gt = O.synthetic_batch(122, (2, 3, 32, 48))
    od = out.to(dev).requires_grad_(True)
    loss = crit(od, gt.to(dev))
    loss.backward()
    k = m.trans.density_k.detach().cpu().double()
    o64 = out.double().requires_grad_(True)
    oh, gh = O.hvit(o64, k), O.hvit(gt.double(), k)
    ref = ((o64 - gt.double()).abs().mean() + O.ssim_loss(o64, gt.double(), 0.5) + O.edge_loss(o64, gt.double(), 50.0)) + \
        1.0 * ((oh - gh).abs().mean() + O.ssim_loss(oh, gh, 0.5) + O.edge_loss(oh, gh, 50.0))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 5e-6, (loss.item(), ref.item())
    d = (od.grad.cpu().double() - o64.grad).abs()
    # |.| kinks of the L1 terms and the hue branch cuts of HVIT: compare where the two gradients can agree
    ok = d <= 2e-4 * o64.grad.abs().max().item() + 1e-9
    assert ok.double().mean().item() > 0.995, f"only {ok.double().mean().item():.4f} of the gradient entries agree"


def test_cidnet_loss_density_k_gradient_through_target(dev):
    """train.py:60-65: gt_hvi = model.HVIT(gt_rgb) is not detached, so density_k gets gradient through the output AND
    the target of every HVI-space term (they largely cancel as the output approaches gt).  The trans.density_k
    gradient of CIDNetLoss must equal the oracle's with k a differentiable leaf on both sides, and each loss class
    must hand its target the right gradient."""
    import hvi_cidnet_amd as P
    m = P.CIDNet(channels=[12, 12, 24, 48]).to(dev)
    crit = P.CIDNetLoss(m, L1_weight=1.0, D_weight=0.5, E_weight=50.0, HVI_weight=1.0)
    out = O.synthetic_batch(141, (2, 3, 32, 48))
    gt = (0.7 * out + 0.3 * O.synthetic_batch(142, (2, 3, 32, 48))).clamp(0, 1)
    od = out.to(dev).requires_grad_(True)
    m.trans.density_k.grad = None
    loss = crit(od, gt.to(dev))
    loss.backward()
    k64 = m.trans.density_k.detach().cpu().double().requires_grad_(True)
    o64 = out.double().requires_grad_(True)
    oh, gh = O.hvit(o64, k64), O.hvit(gt.double(), k64)
    ref = ((o64 - gt.double()).abs().mean() + O.ssim_loss(o64, gt.double(), 0.5) + O.edge_loss(o64, gt.double(), 50.0)) + \
        1.0 * ((oh - gh).abs().mean() + O.ssim_loss(oh, gh, 0.5) + O.edge_loss(oh, gh, 50.0))
    ref.backward()
    gk, rk = m.trans.density_k.grad.item(), k64.grad.item()
    # the output-only half (what a detached target would give) is far from the full value: the test can tell them apart
    k2 = m.trans.density_k.detach().cpu().double().requires_grad_(True)
    oh2, gh2 = O.hvit(out.double(), k2), O.hvit(gt.double(), k2).detach()
    ((oh2 - gh2).abs().mean() + O.ssim_loss(oh2, gh2, 0.5) + O.edge_loss(oh2, gh2, 50.0)).backward()
    assert abs(k2.grad.item() - rk) > 20 * (2e-3 * abs(rk) + 1e-6), "fixture cannot distinguish a detached target"
    assert abs(gk - rk) <= 2e-3 * abs(rk) + 1e-6, (gk, rk, k2.grad.item())
    # per-class target gradients against autograd on the oracle
    a, b = O.synthetic_batch(143, (2, 3, 24, 40)), O.synthetic_batch(144, (2, 3, 24, 40))
    for name, mod, fn in (("l1", P.L1Loss(), lambda x, y: (x - y).abs().mean()),
                          ("ssim", P.SSIM(weight=0.5), lambda x, y: O.ssim_loss(x, y, 0.5)),
                          ("edge", P.EdgeLoss(loss_weight=50.0), lambda x, y: O.edge_loss(x, y, 50.0))):
        xd, yd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        (2.0 * mod(xd, yd)).backward()
        x64, y64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
        (2.0 * fn(x64, y64)).backward()
        for got, want, which in ((xd.grad, x64.grad, "input"), (yd.grad, y64.grad, "target")):
            d = (got.cpu().double() - want).abs().max().item()
            assert d <= 1e-4 * want.abs().max().item() + 1e-10, f"{name} {which} gradient: {d:.3e}"


@pytest.mark.parametrize("shape", [(2, 3, 32, 48), (1, 3, 37, 51), (1, 3, 18, 22)])
def test_perceptual_loss_vs_oracle(dev, shape):
    """VGG19 perceptual loss ("next" row f4) as train.py:192 builds it (conv1_2, conv2_2, conv3_4, conv4_4 before the ReLU,
    'mse', range_norm) against the fp64 oracle restated from loss/vgg_arch.py + loss/losses.py (parity unpinned: torchvision
    is absent): value, d/dx, the frozen weights get no gradient; odd sizes exercise MaxPool2d's floor and its uncovered
    last row / column in the backward"""
    import hvi_cidnet_amd as P
    crit = P.PerceptualLoss({"conv1_2": 1, "conv2_2": 1, "conv3_4": 0.5, "conv4_4": 2}, perceptual_weight=1.5, criterion="mse").to(dev)
    assert all(not p.requires_grad for p in crit.parameters())
    keys = list(crit.state_dict().keys())
    assert "vgg.vgg_net.conv1_1.weight" in keys and "vgg.vgg_net.conv4_4.bias" in keys and "vgg.mean" in keys
    x = O.synthetic_batch(151, shape) * 2 - 1
    gt = (0.6 * x + 0.4 * (O.synthetic_batch(152, shape) * 2 - 1))
    xd = x.to(dev).requires_grad_(True)
    loss, style = crit(xd, gt.to(dev))
    assert style is None
    (0.7 * loss).backward()
    convs = {n: (getattr(crit.vgg.vgg_net, n).weight.detach().cpu().double(), getattr(crit.vgg.vgg_net, n).bias.detach().cpu().double())
             for n in O.VGG19_NAMES if n != "pool" and hasattr(crit.vgg.vgg_net, n)}
    x64 = x.double().requires_grad_(True)
    ref = O.perceptual_loss(x64, gt.double(), convs, {"conv1_2": 1, "conv2_2": 1, "conv3_4": 0.5, "conv4_4": 2}, 1.5, True)
    (0.7 * ref).backward()
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-9, (loss.item(), ref.item())
    d = (xd.grad.cpu().double() - x64.grad).abs().max().item()
    assert d <= 2e-4 * x64.grad.abs().max().item() + 1e-12, (d, x64.grad.abs().max().item())


def test_cidnet_loss_with_perceptual_term(dev):
    """CIDNetLoss(P_weight > 0) = the reference's full objective (train.py:61-65): the VGG term is added on the RGB pair
    and on the HVIT pair, and the model still trains"""
    import hvi_cidnet_amd as P
    m = P.CIDNet(channels=[12, 12, 24, 48]).to(dev)
    full = P.CIDNetLoss(m, P_weight=1e-2).to(dev)
    base = P.CIDNetLoss(m, P_weight=0.0)
    out = O.synthetic_batch(161, (1, 3, 32, 48)).to(dev).requires_grad_(True)
    gt = O.synthetic_batch(162, (1, 3, 32, 48)).to(dev)
    lf, lb = full(out, gt), base(out, gt)
    p_rgb = full.perceptual(out, gt)[0]
    p_hvi = full.perceptual(m.HVIT(out), m.HVIT(gt))[0]
    assert abs(lf.item() - (lb.item() + 1e-2 * (p_rgb.item() + p_hvi.item()))) <= 1e-5 * abs(lf.item())
    lf.backward()
    assert torch.isfinite(out.grad).all() and out.grad.abs().max().item() > 0


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 3, 9, 7), (2, 3, 8, 2), (1, 3, 2, 12), (1, 3, 64, 96)])
def test_tnsm_noise_objective_vs_oracle(dev, shape):
    """train_tnsm.py:68-72 on device (VERDICT r2 missing #1): weight * (noise_consistency_loss + noise_smoothing_loss) and
    its gradients wrt the fused noise map and wrt output_rgb against the fp64 oracle restatement (parity unpinned: the
    formula is inline in a script that cannot be imported), even / odd / two-row / two-column sizes (with a single row
    or column the reference formula is the mean of an empty tensor, NaN; the network never produces such maps)."""
    import hvi_cidnet_amd as P
    B, C, H, W = shape
    nm = O.synthetic_batch(301, shape)
    out = O.synthetic_batch(302, shape)
    im = (0.6 * out + 0.4 * O.synthetic_batch(303, shape)).clamp(0, 1)
    w = 0.7
    nm64, out64 = nm.double().requires_grad_(True), out.double().requires_grad_(True)
    c64, s64 = O.tnsm_noise_losses(nm64, out64, im.double())
    (w * (c64 + s64)).backward()
    nmd, outd = nm.to(dev).requires_grad_(True), out.to(dev).requires_grad_(True)
    loss = P.tnsm_noise_loss(nmd, outd, im.to(dev), w)
    (2.0 * loss).backward()                                     # upstream scalar != 1 exercises the device-side scaling
    ref = (w * (c64 + s64)).item()
    assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref)), (loss.item(), ref)
    for got, want, what in ((nmd.grad, nm64.grad, "d/d noise_map"), (outd.grad, out64.grad, "d/d output_rgb")):
        want = 2.0 * want
        err = (got.cpu().double() - want).abs().max().item()
        assert err <= 1e-5 * want.abs().max().item() + 1e-9, (what, err)


def test_cidnet_loss_with_tnsm_terms(dev):
    """CIDNetLoss(tnsm_weight=1) = the base objective + the TNSM noise terms, through CIDNet_TNSM's own train-mode outputs"""
    import hvi_cidnet_amd as P
    chans = (12, 12, 24, 48)
    m = P.CIDNet_TNSM(channels=list(chans))
    p = O.make_params(9, channels=chans, variant="tnsm")
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
    m.to(dev).train()
    x = O.synthetic_batch(311, (1, 3, 32, 48)).to(dev)
    gt = O.synthetic_batch(312, (1, 3, 32, 48)).to(dev)
    rgb, noise = m(x)
    base = P.CIDNetLoss(m)(rgb, gt)
    full = P.CIDNetLoss(m, tnsm_weight=1.0)(rgb, gt, noise, x)
    c, s = O.tnsm_noise_losses(noise.detach().cpu().double(), rgb.detach().cpu().double(), x.cpu().double())
    assert abs((full - base).item() - (c + s).item()) <= 5e-6
    full.backward()
    assert m.noise_fusion[0].weight.grad is not None and torch.isfinite(m.noise_fusion[0].weight.grad).all()
