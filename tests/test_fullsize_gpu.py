"""GPU parity at BASELINE.json's own sizes against the reference's outputs (tests/golden/fullsize.npz, written by
oracle/gen_golden.py::gen_fullsize through the imported reference):

  * 1x3x400x600 forward + backward, full width -- and the SAME image eight times as the benchmark's 8x3x400x600
    batch (mean-L1 over eight identical samples has the single-sample gradient), so the tile shapes, multi-round
    grids and split-K slab counts the benchmark launches are the ones compared with the reference;
  * 1x3x1024x1024 forward (configs[3] image size) and one 32x3x1024x1024 pass through batch-independence;
  * full-width (36/36/72/144, c/head = 18) CIDNet_MSSA and CIDNet_TNSM at 1x3x64x96, every gradient tensor.

Large gradient tensors are stored as a fingerprint (sum, sum|.|, dot with a fixed weight vector) plus a strided
sample (oracle.grad_fingerprint); one tensor per kernel family is stored whole.
Bars: outputs 1e-4 abs (north star).  Gradients at 400x600: the reference's own fp32 gradients sit 4e-4 .. 5e-3 of each
tensor's max away from the fp64 truth (0.2 for a nearly-cancelling PReLU slope; measured, oracle/gen_golden.py), so the
fixture also holds the fp64 gradients and the bar is: our distance from fp64 <= 2x the reference's own + 2e-4 of the max
(4x for one-element gradients, see check_grad)."""
import numpy as np
import pytest
import torch

from oracle import cidnet_oracle as O

pytestmark = pytest.mark.gpu


def _t(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def load(module, params):
    module.load_state_dict({k: params[k] for k in module.state_dict().keys()}, strict=True)


def check_grad(g, tag, name, grad, rel=2e-4, sum_rel=2e-4, tag64=None):
    """gradient tensor `grad` of parameter `name` against the fixture group `tag` (the reference's fp32 values).
    With `tag64` (fp64 truth of the same gradient) the bar is relative to the reference's OWN fp32 rounding distance
    from fp64: ours may be at most twice as far (+ rel of the tensor's max)."""
    sums, sample, _ = O.grad_fingerprint(grad, 512)
    ref_s, ref_fp = _t(g[f"{tag}_gs.{name}"]).double(), g[f"{tag}_gfp.{name}"]
    key = f"{tag}_g.{name}"
    full = _t(g[key]).double() if key in g.files else None
    ours_full = grad.detach().cpu().double()
    if tag64 is None:
        scale = ref_s.abs().max().item()
        d = (sample.double() - ref_s).abs().max().item()
        assert d <= rel * scale + 1e-7, f"{tag} d{name}: strided sample differs by {d:.3e} (max |g| {scale:.3e})"
        tol = sum_rel * ref_fp[1] + 1e-7
        assert abs(sums[0].item() - ref_fp[0]) <= tol, f"{tag} d{name}: sum {sums[0].item():.6e} vs {ref_fp[0]:.6e}"
        assert abs(sums[2].item() - ref_fp[2]) <= tol, f"{tag} d{name}: weighted sum {sums[2].item():.6e} vs {ref_fp[2]:.6e}"
        if full is not None:
            d = (ours_full - full).abs().max().item()
            assert d <= rel * full.abs().max().item() + 1e-7, f"{tag} d{name}: full tensor differs by {d:.3e}"
        return int(full is not None)
    s64, fp64 = _t(g[f"{tag64}_gs.{name}"]).double(), g[f"{tag64}_gfp.{name}"]
    scale = s64.abs().max().item()
    ref_err = (ref_s - s64).abs().max().item()
    our_err = (sample.double() - s64).abs().max().item()
    # One-element gradients (PReLU slopes, temperatures, density_k) are sums of ~1e7 signed terms that cancel to 1e-3 of
    # their magnitude; where fp32 rounding lands is a coin toss for ANY summation order (the reference's own value is
    # 5e-4 relative off fp64 here), so they get four instead of two times the reference's distance.
    fac = 4.0 if grad.numel() == 1 else 2.0
    assert our_err <= fac * ref_err + rel * scale + 1e-10, \
        f"{tag} d{name}: sample error vs fp64 {our_err:.3e}, reference's own {ref_err:.3e}, max |g| {scale:.3e}"
    for i in (0, 2):
        ref_e, our_e = abs(ref_fp[i] - fp64[i]), abs(sums[i].item() - fp64[i])
        # every element may be off by the per-element bar (rel of the tensor's max, checked above on the sample): n such
        # errors, independent, move a sum by ~ sqrt(n) of that (x3 for safety).  A dropped border tap or tile shows as
        # a bias of >= 1e-3 of sum|g|, far above this.
        assert our_e <= fac * ref_e + 3.0 * grad.numel() ** 0.5 * rel * scale + 1e-10, \
            f"{tag} d{name}: sum[{i}] error vs fp64 {our_e:.3e}, reference's own {ref_e:.3e}, sum|g| {fp64[1]:.3e}"
    if full is not None:
        f64 = _t(g[f"{tag64}_g.{name}"]).double()
        ref_err = (full - f64).abs().max().item()
        our_err = (ours_full - f64).abs().max().item()
        assert our_err <= fac * ref_err + rel * f64.abs().max().item() + 1e-10, \
            f"{tag} d{name}: full-tensor error vs fp64 {our_err:.3e}, reference's own {ref_err:.3e}"
    return int(full is not None)


@pytest.mark.parametrize("batch", [1, 8])
def test_cidnet_400x600_fwd_bwd_golden(golden, dev, batch):
    import hvi_cidnet_amd as P
    g = golden("fullsize")
    m = P.CIDNet()
    load(m, O.make_params(5))
    m.to(dev)
    x1 = O.synthetic_batch(161, (1, 3, 400, 600))
    gt1 = O.synthetic_batch(162, (1, 3, 400, 600))
    x = x1.repeat(batch, 1, 1, 1).to(dev).requires_grad_(True)
    gt = gt1.repeat(batch, 1, 1, 1).to(dev)
    y = m(x)
    ref = _t(g["a_out_strided"])
    for b in range(batch):
        d = (y[b:b + 1, :, ::8, ::8].detach().cpu() - ref).abs().max().item()
        assert d <= 1e-4, f"sample {b}: strided output differs by {d:.3e}"
    s = g["a_out_sums"]
    yd = y[batch - 1].detach().double()
    assert abs(yd.sum().item() - s[0]) <= 1e-5 * s[1]
    assert abs((yd ** 2).sum().item() - s[2]) <= 1e-5 * s[2]
    loss = (y - gt).abs().mean()
    assert abs(loss.item() - float(g["a_loss"])) < 1e-5
    loss.backward()
    # d(loss)/d(input): each of the identical samples carries 1/batch of the single-sample gradient
    gx = x.grad[batch - 1:batch].detach().cpu() * batch
    refx, x64 = _t(g["a_gx_strided"]).double(), _t(g["a64_gx_strided"]).double()
    ref_err = (refx - x64).abs().max().item()
    our_err = (gx[:, :, ::8, ::8].double() - x64).abs().max().item()
    assert our_err <= 2.0 * ref_err + 2e-4 * x64.abs().max().item(), f"d/dx error vs fp64 {our_err:.3e}, reference's own {ref_err:.3e}"
    fp = O.grad_fingerprint(gx)[0]
    ref_e = abs(g["a_gx_fp"][2] - g["a64_gx_fp"][2])
    assert abs(fp[2].item() - g["a64_gx_fp"][2]) <= 2.0 * ref_e + 1e-5 * g["a64_gx_fp"][1]
    n_full = n = 0
    for name, prm in m.named_parameters():
        if name.startswith("I_LCA5."):
            assert prm.grad is None
            continue
        n_full += check_grad(g, "a", name, prm.grad, tag64="a64")
        n += 1
    assert n == 191 - 13 and n_full >= 16


def test_cidnet_1024_forward_golden(golden, dev):
    """configs[3]: 1024x1024 forward against the reference, then the 32-image batch through batch independence"""
    import hvi_cidnet_amd as P
    g = golden("fullsize")
    m = P.CIDNet()
    load(m, O.make_params(5, jitter=False))
    m.to(dev).eval()
    x1 = O.synthetic_batch(171, (1, 3, 1024, 1024), quantised=True).to(dev)
    with torch.no_grad():
        y1 = m(x1)
    d = (y1[:, :, ::32, ::32].cpu() - _t(g["b_out_strided"])).abs().max().item()
    assert d <= 1e-4, f"1024x1024 strided output differs by {d:.3e}"
    s = g["b_out_sums"]
    yd = y1.double()
    assert abs(yd.sum().item() - s[0]) <= 1e-5 * s[1]
    assert abs((yd ** 2).sum().item() - s[2]) <= 1e-5 * s[2]
    # black pixels: PHVIT's hi == 6 case and v == 0 give exact zeros in the reference; a pixel whose hue sits within
    # rounding of the 6/6 boundary may fall on either side, hence a small allowance on the count
    n_black = int((y1 == 0).all(1).sum().item())
    assert abs(n_black - int(g["b_n_black"])) <= 4, (n_black, int(g["b_n_black"]))
    xs = torch.cat([x1, O.synthetic_batch(172, (30, 3, 1024, 1024)).to(dev), x1], 0)
    with torch.no_grad():
        y32 = m(xs)
    assert y32.shape == (32, 3, 1024, 1024) and torch.isfinite(y32).all()
    for i in (0, 31):
        d = (y32[i:i + 1] - y1).abs().max().item()
        assert d <= 2e-6, f"sample {i} of the 32-image batch differs from its single-image output by {d:.3e}"


def test_cidnet_mssa_fullwidth_golden(golden, dev):
    import hvi_cidnet_amd as P
    g = golden("fullsize")
    m = P.CIDNet_MSSA()
    load(m, O.make_params(5, variant="mssa"))
    m.to(dev)
    y = m(_t(g["mssa_x"], dev))
    d = (y.detach().cpu() - _t(g["mssa_out"])).abs().max().item()
    assert d <= 1e-4, f"full-width MSSA forward differs by {d:.3e}"
    (y - _t(g["mssa_gt"], dev)).abs().mean().backward()
    # kink margin of this fixture is 1.9e-7 (stored): an fp32 forward that differs by a few 1e-7 may put one PReLU /
    # channel-max argument on the other side, which moves a gradient by one pixel's share (~1/1500 at 32x48)
    n = 0
    for name, prm in m.named_parameters():
        assert prm.grad is not None, name
        check_grad(g, "mssa", name, prm.grad, rel=1e-3, sum_rel=1e-3)
        n += 1
    assert n == 197


def test_cidnet_tnsm_fullwidth_golden(golden, dev):
    import hvi_cidnet_amd as P
    g = golden("fullsize")
    m = P.CIDNet_TNSM()
    load(m, O.make_params(5, variant="tnsm"))
    m.to(dev).train()
    y, fz = m(_t(g["tnsm_x"], dev))
    d = (y.detach().cpu() - _t(g["tnsm_out"])).abs().max().item()
    assert d <= 1e-4, f"full-width TNSM rgb differs by {d:.3e}"
    d = (fz.detach().cpu() - _t(g["tnsm_noise"])).abs().max().item()
    assert d <= 1e-5, f"full-width TNSM fused noise differs by {d:.3e}"
    ((y - _t(g["tnsm_gt"], dev)).abs().mean() + 0.1 * fz.mean()).backward()
    dead = set(g["tnsm_dead"].tolist())
    # as in test_model_gpu.test_cidnet_tnsm_golden: the un-normalised attention saturates the softmax and the
    # reference's own fp32 gradients sit ~1e-3 from the fp64 truth, so our error against fp64 may not exceed twice
    # the reference's own (+2e-4 of the tensor's max), on the strided sample of every tensor
    n = 0
    for name, prm in m.named_parameters():
        if name in dead:
            assert prm.grad is None, name
            continue
        ours = O.grad_fingerprint(prm.grad, 512)[1].double()
        g64 = _t(g[f"tnsm64_gs.{name}"]).double()
        ref = _t(g[f"tnsm_gs.{name}"]).double()
        ref_err = (ref - g64).abs().max().item()
        our_err = (ours - g64).abs().max().item()
        assert our_err <= 2.0 * ref_err + 2e-4 * g64.abs().max().item() + 1e-9, (name, our_err, ref_err)
        n += 1
    assert n == 468 - len(dead) or n > 400


def _sums_ok(y, s, what):
    yd = y.double()
    assert abs(yd.sum().item() - s[0]) <= 1e-5 * s[1], what
    assert abs((yd ** 2).sum().item() - s[2]) <= 1e-5 * s[2], what


def test_cidnet_mssa_400x600_forward_and_bs16(golden, dev):
    """BASELINE configs[4] (CIDNet_MSSA, bs=16, 400x600 planes): the 1x3x400x600 forward against the reference
    (tests/golden/round3.npz), then the same image inside a 16-image batch (batch independence: the 7x7 gate, the
    channel mean / max pass and every other kernel of sa.hip run on the benchmark's plane sizes and grid shapes)."""
    import hvi_cidnet_amd as P
    g = golden("round3")
    m = P.CIDNet_MSSA()
    load(m, O.make_params(5, variant="mssa"))
    m.to(dev).eval()
    x1 = O.synthetic_batch(181, (1, 3, 400, 600)).to(dev)
    with torch.no_grad():
        y1 = m(x1)
    d = (y1[:, :, ::8, ::8].cpu() - _t(g["mssa400_out_strided"])).abs().max().item()
    assert d <= 1e-4, f"MSSA 400x600 strided output differs from the reference by {d:.3e}"
    _sums_ok(y1, g["mssa400_out_sums"], "MSSA 400x600 checksums")
    xs = torch.cat([O.synthetic_batch(182, (7, 3, 400, 600)).to(dev), x1, O.synthetic_batch(183, (8, 3, 400, 600)).to(dev)], 0)
    with torch.no_grad():
        y16 = m(xs)
    assert y16.shape == (16, 3, 400, 600) and torch.isfinite(y16).all()
    d = (y16[7:8] - y1).abs().max().item()
    assert d <= 2e-6, f"sample 7 of the 16-image batch differs from its single-image output by {d:.3e}"


def test_cidnet_tnsm_400x600_forward_and_bs16(golden, dev):
    """BASELINE configs[4] (CIDNet_TNSM, bs=16, 400x600): train-mode forward (rgb + fused noise map) at 1x3x400x600.
    TNSM's attention is NOT normalised (net/TNSM.py:98-104): its logits are raw dot products over 60 000 pixels, the
    softmax saturates, and the reference's own fp32 rgb sits 1.3e-3 from an fp64 evaluation at this size (measured,
    oracle/gen_golden.py::gen_round3).  Bar: our distance from fp64 <= twice the reference's own + 1e-4; the fused noise
    map (no attention upstream of it) 1e-5 against the reference.  Then batch independence inside 16 images."""
    import hvi_cidnet_amd as P
    g = golden("round3")
    m = P.CIDNet_TNSM()
    load(m, O.make_params(5, variant="tnsm"))
    m.to(dev).train()
    x1 = O.synthetic_batch(181, (1, 3, 400, 600)).to(dev)
    with torch.no_grad():
        y1, f1 = m(x1)
    y64, ref = _t(g["tnsm400_out64_strided"]).double(), _t(g["tnsm400_out_strided"]).double()
    ref_err = (ref - y64).abs().max().item()
    our_err = (y1[:, :, ::8, ::8].cpu().double() - y64).abs().max().item()
    assert our_err <= 2.0 * ref_err + 1e-4, f"TNSM 400x600 rgb: error vs fp64 {our_err:.3e}, the reference's own {ref_err:.3e}"
    d = (f1[:, :, ::8, ::8].cpu() - _t(g["tnsm400_noise_strided"])).abs().max().item()
    assert d <= 1e-5, f"TNSM 400x600 fused noise map differs from the reference by {d:.3e}"
    xs = torch.cat([O.synthetic_batch(182, (7, 3, 400, 600)).to(dev), x1, O.synthetic_batch(183, (8, 3, 400, 600)).to(dev)], 0)
    with torch.no_grad():
        y16, f16 = m(xs)
    assert y16.shape == (16, 3, 400, 600) and f16.shape == (16, 3, 400, 600) and torch.isfinite(y16).all()
    assert (f16[7:8] - f1).abs().max().item() <= 2e-6
    d = (y16[7:8] - y1).abs().max().item()
    assert d <= 2e-6, f"sample 7 of the 16-image batch differs from its single-image output by {d:.3e}"


def test_cidnet_400x600_bf16_mode_vs_reference(golden, dev):
    """BASELINE configs[2] names bf16: P.set_precision("bf16") -- convolution operands rounded to bf16 on the matrix cores
    (one product per term, fp32 accumulation: the arithmetic of a bf16 autocast conv) and the LCA-internal tensors STORED
    as bfloat16 -- on the benchmark's 8x3x400x600 batch, against the REFERENCE's fp32 output and the fp64 gradients of
    tests/golden/fullsize.npz.  Its own tolerance tier, stated here: output 5e-3 absolute (bf16 operand rounding is 2^-9
    relative per product; measured 3e-4); every gradient tensor with more than 16 elements: cosine similarity of its
    strided sample with the fp64 truth >= 0.995, sample error <= 10 % of the tensor's max, whole-tensor sum within 10 % of
    sum|g| (measured: worst 5.5 % / cos 0.999); the one-element gradients -- PReLU slopes, temperatures, density_k: sums of
    ~1e7 cancelling terms -- are excluded, bf16 rounding moves them by more than their own magnitude."""
    import hvi_cidnet_amd as P
    g = golden("fullsize")
    m = P.CIDNet()
    load(m, O.make_params(5))
    m.to(dev)
    x1 = O.synthetic_batch(161, (1, 3, 400, 600))
    gt1 = O.synthetic_batch(162, (1, 3, 400, 600))
    x, gt = x1.repeat(8, 1, 1, 1).to(dev), gt1.repeat(8, 1, 1, 1).to(dev)
    P.set_precision("bf16")
    try:
        y = m(x)
        (y - gt).abs().mean().backward()
        torch.cuda.synchronize()
    finally:
        P.set_precision("f32")
    d = (y[:1, :, ::8, ::8].detach().cpu() - _t(g["a_out_strided"])).abs().max().item()
    print(f"bf16 mode, 8x3x400x600: output max |diff| vs the reference {d:.3e}")
    assert d <= 5e-3
    worst, wname, wcos, cname, n = 0.0, "", 1.0, "", 0
    for name, prm in m.named_parameters():
        if name.startswith("I_LCA5."):
            assert prm.grad is None
            continue
        sums, sample, _ = O.grad_fingerprint(prm.grad, 512)
        s64, fp64 = _t(g[f"a64_gs.{name}"]).double(), g[f"a64_gfp.{name}"]
        scale = max(s64.abs().max().item(), 1e-30)
        e = (sample.double() - s64).abs().max().item() / scale
        if prm.grad.numel() > 1:
            if e > worst:
                worst, wname = e, name
            assert e <= 1e-1, (name, e)
            assert abs(sums[0].item() - fp64[0]) <= 1e-1 * fp64[1] + 1e-9, name
        if prm.grad.numel() > 16:
            cos = (sample.double() * s64).sum().item() / max(sample.double().norm().item() * s64.norm().item(), 1e-300)
            if cos < wcos:
                wcos, cname = cos, name
            assert cos >= 0.995, (name, cos)
        n += 1
    print(f"bf16 mode: worst gradient sample error vs fp64 {worst:.3e} of the tensor's max ({wname}); lowest cosine {wcos:.5f} ({cname})")
    assert n == 178


@pytest.mark.parametrize("variant,batch", [("mssa", 1), ("mssa", 16), ("tnsm", 1), ("tnsm", 16)])
def test_variants_400x600_backward_golden(golden, dev, variant, batch):
    """BASELINE configs[4] with the backward, at its image size and batch: CIDNet_MSSA / CIDNet_TNSM (train mode, loss = L1 +
    0.1 mean(fused noise) for TNSM) forward + backward on 3x400x600 against tests/golden/round4.npz -- every live gradient
    tensor of the imported reference (net/CIDNet_MSSA.py:100-159, net/CIDNet_TNSM.py:101-294) as fingerprints, next to an fp64
    evaluation of the same step.  batch = 16 uses identical samples: every sample's output equals the single-image fixture
    and the parameter gradients of the mean loss are unchanged.

    The bar, per tensor (strided sample of 512 elements, error against fp64):
        ours <= 2 x the reference's own error + rel_noise x max|g| + abs_noise
    rel_noise = 3 x the 90th percentile, over all tensors, of the reference's OWN error relative to the tensor's max (MSSA:
    ~6e-4; TNSM: ~5e-2 -- its un-normalised attention saturates the softmax, net/TNSM.py:98-104, and the reference's fp32
    gradients are that far from fp64); abs_noise = 1e-2 x the median over tensors of max|g| (gradients two orders below the
    model's typical gradient are rounding noise in the reference too: up to 800 % off there).  One-element gradients are
    sums of ~1e7 cancelling terms, so where a single tensor's fp32 rounding lands is a coin toss for ANY summation order; the
    population terms say "inside the reference's own noise envelope", the 2x term keeps well-conditioned tensors tight.
    d(loss)/d(input): at most 0.1 % of the sampled pixels above the per-pixel bar (a PReLU / channel-max argument within
    fp32 rounding of its kink flips one pixel's share; measured: 1 of 11 250)."""
    import hvi_cidnet_amd as P
    g = golden("round4")
    tag = f"{variant}400"
    m = (P.CIDNet_MSSA if variant == "mssa" else P.CIDNet_TNSM)()
    load(m, O.make_params(5, variant=variant))
    m.to(dev).train()
    x1 = O.synthetic_batch(191, (1, 3, 400, 600))
    gt1 = O.synthetic_batch(192, (1, 3, 400, 600))
    x = x1.repeat(batch, 1, 1, 1).to(dev).requires_grad_(True)
    gt = gt1.repeat(batch, 1, 1, 1).to(dev)
    res = m(x)
    y = res[0] if variant == "tnsm" else res
    loss = (y - gt).abs().mean() + (0.1 * res[1].mean() if variant == "tnsm" else 0.0)
    ref, r64 = _t(g[f"{tag}_out_strided"]).double(), _t(g[f"{tag}64_out_strided"]).double()
    ref_err = (ref - r64).abs().max().item()
    for b in (0, batch - 1):
        our_err = (y[b:b + 1, :, ::8, ::8].detach().cpu().double() - r64).abs().max().item()
        assert our_err <= 2.0 * ref_err + 1e-4, f"{tag} sample {b}: output error vs fp64 {our_err:.3e}, the reference's own {ref_err:.3e}"
    assert abs(loss.item() - float(g[f"{tag}_loss"])) <= 2.0 * ref_err + 1e-5
    loss.backward()
    torch.cuda.synchronize()
    dead = set(str(n) for n in g[f"{tag}_dead"])
    live = [(n, prm) for n, prm in m.named_parameters() if n not in dead]
    for n, prm in m.named_parameters():
        assert (prm.grad is None) == (n in dead), n
    assert len(live) == (197 if variant == "mssa" else 450)
    scale = {n: max(_t(g[f"{tag}64_gs.{n}"]).abs().max().item(), 1e-30) for n, _ in live}
    ref_e = {n: (_t(g[f"{tag}_gs.{n}"]).double() - _t(g[f"{tag}64_gs.{n}"]).double()).abs().max().item() for n, _ in live}
    rel_noise = 3.0 * float(np.percentile([ref_e[n] / scale[n] for n, _ in live], 90))
    abs_noise = 1e-2 * float(np.median([scale[n] for n, _ in live]))
    worst = (0.0, "")
    for n, prm in live:
        ours = O.grad_fingerprint(prm.grad, 512)[1].double()
        e = (ours - _t(g[f"{tag}64_gs.{n}"]).double()).abs().max().item()
        bar = 2.0 * ref_e[n] + rel_noise * scale[n] + abs_noise
        worst = max(worst, (e / bar, n))
        assert e <= bar, f"{tag} d{n}: error vs fp64 {e:.3e} > {bar:.3e} (reference's own {ref_e[n]:.3e}, max |g| {scale[n]:.3e})"
    print(f"{tag} bs={batch}: rel_noise {rel_noise:.2e}, abs_noise {abs_noise:.2e}, worst error / bar {worst[0]:.2f} ({worst[1]})")
    gx = x.grad[batch - 1:batch].detach().cpu() * batch
    refx, x64 = _t(g[f"{tag}_gx_strided"]).double(), _t(g[f"{tag}64_gx_strided"]).double()
    e_ref = (refx - x64).abs()
    e_our = (gx[:, :, ::8, ::8].double() - x64).abs()
    bar = 2.0 * e_ref.max().item() + 2e-4 * x64.abs().max().item()
    n_over = int((e_our > bar).sum().item())
    assert n_over <= 1e-3 * e_our.numel(), f"{tag} d/dx: {n_over} of {e_our.numel()} sampled pixels above {bar:.3e}"
    q_our, q_ref = torch.quantile(e_our.flatten(), 0.999).item(), torch.quantile(e_ref.flatten(), 0.999).item()
    assert q_our <= 2.0 * q_ref + 2e-4 * x64.abs().max().item(), f"{tag} d/dx: 99.9th percentile error {q_our:.3e}, the reference's {q_ref:.3e}"
