"""GPU parity of the reference-named blocks and of the whole CIDNet against the golden fixtures
(the reference's own outputs and gradients, tests/golden/*.npz) through the public module API.
Bars: outputs 1e-4 abs (north star; observed ~1e-6), gradients 1e-4 of the tensor's max."""
import os

import numpy as np
import pytest
import torch

from oracle import cidnet_oracle as O

pytestmark = pytest.mark.gpu


def _t(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def out_ok(a, b, tol=1e-4, what=""):
    d = (a.detach().cpu().double() - _t(b).double()).abs().max().item()
    assert d <= tol, f"{what}: max abs diff {d:.3e}"
    return d


def grad_ok(a, b, rel=1e-4, what=""):
    b = _t(b).double()
    d = (a.detach().cpu().double() - b).abs().max().item()
    tol = rel * b.abs().max().item() + 1e-7
    assert d <= tol, f"{what}: max diff {d:.3e} > {tol:.3e}"


def load(module, params, prefix=""):
    sd = {k: params[prefix + k] for k in module.state_dict().keys()}
    module.load_state_dict(sd, strict=True)


CASES = [("w12", (12, 12, 24, 48)), ("w36", (36, 36, 72, 144))]


@pytest.mark.parametrize("tag,chans", CASES)
def test_layernorm_golden(golden, dev, tag, chans):
    import hvi_cidnet_amd as P
    g = golden("blocks")
    p = O.make_params(11, channels=chans)
    m = P.LayerNorm(chans[1])
    load(m, p, "I_LCA1.norm.")
    m.to(dev)
    x = _t(g[f"ln_{tag}_x"], dev).requires_grad_(True)
    y = m(x)
    out_ok(y, g[f"ln_{tag}_y"], 1e-5, "ln fwd")
    y.backward(_t(g[f"ln_{tag}_gy"], dev))
    grad_ok(x.grad, g[f"ln_{tag}_gx"], what="ln gx")
    grad_ok(m.weight.grad, g[f"ln_{tag}_gw"], what="ln gw")
    grad_ok(m.bias.grad, g[f"ln_{tag}_gb"], what="ln gb")


@pytest.mark.parametrize("tag,chans", CASES)
@pytest.mark.parametrize("kind", ["i_lca", "hv_lca"])
def test_lca_golden(golden, dev, tag, chans, kind):
    import hvi_cidnet_amd as P
    g = golden("blocks")
    p = O.make_params(11, channels=chans)
    pre = "I_LCA1" if kind == "i_lca" else "HV_LCA1"
    m = (P.I_LCA if kind == "i_lca" else P.HV_LCA)(chans[1], 2)
    load(m, p, pre + ".")
    m.to(dev)
    x = _t(g[f"{kind}_{tag}_x"], dev).requires_grad_(True)
    y = _t(g[f"{kind}_{tag}_y"], dev).requires_grad_(True)
    z = m(x, y)
    out_ok(z, g[f"{kind}_{tag}_out"], 1e-5, f"{kind} fwd")
    z.backward(_t(g[f"{kind}_{tag}_gout"], dev))
    grad_ok(x.grad, g[f"{kind}_{tag}_gx"], what="gx")
    grad_ok(y.grad, g[f"{kind}_{tag}_gy"], what="gy")
    for n, prm in m.named_parameters():
        grad_ok(prm.grad, g[f"{kind}_{tag}_g.{n}"], what=f"{kind} d{n}")


@pytest.mark.parametrize("tag,chans", CASES)
def test_down_up_golden(golden, dev, tag, chans):
    import hvi_cidnet_amd as P
    g = golden("blocks")
    p = O.make_params(11, channels=chans)
    dn = P.NormDownsample(chans[1], chans[2])
    load(dn, p, "IE_block2.")
    dn.to(dev)
    x = _t(g[f"down_{tag}_x"], dev).requires_grad_(True)
    y = dn(x)
    out_ok(y, g[f"down_{tag}_out"], 1e-5, "down fwd")
    y.backward(_t(g[f"down_{tag}_gout"], dev))
    grad_ok(x.grad, g[f"down_{tag}_gx"], what="down gx")
    grad_ok(dn.prelu.weight.grad, g[f"down_{tag}_g.prelu.weight"], what="down dslope")
    grad_ok(dn.down[0].weight.grad, g[f"down_{tag}_g.down.0.weight"], what="down dW")
    up = P.NormUpsample(chans[2], chans[1])
    load(up, p, "ID_block2.")
    up.to(dev)
    x = _t(g[f"up_{tag}_x"], dev).requires_grad_(True)
    s = _t(g[f"up_{tag}_skip"], dev).requires_grad_(True)
    y = up(x, s)
    out_ok(y, g[f"up_{tag}_out"], 1e-5, "up fwd")
    y.backward(_t(g[f"up_{tag}_gout"], dev))
    grad_ok(x.grad, g[f"up_{tag}_gx"], what="up gx")
    grad_ok(s.grad, g[f"up_{tag}_gskip"], what="up gskip")
    grad_ok(up.prelu.weight.grad, g[f"up_{tag}_g.prelu.weight"], what="up dslope")
    grad_ok(up.up_scale[0].weight.grad, g[f"up_{tag}_g.up_scale.0.weight"], what="up dW3")
    grad_ok(up.up.weight.grad, g[f"up_{tag}_g.up.weight"], what="up dW1")


@pytest.mark.parametrize("tag,chans", CASES)
def test_cidnet_golden(golden, dev, tag, chans):
    import hvi_cidnet_amd as P
    g = golden("model")
    p = O.make_params(5, channels=chans)
    m = P.CIDNet(channels=list(chans))
    load(m, p)
    m.to(dev)
    x, gt = _t(g[f"model_{tag}_x"], dev), _t(g[f"model_{tag}_gt"], dev)
    y = m(x)
    d = out_ok(y, g[f"model_{tag}_out"], 1e-4, "CIDNet fwd")
    print(f"CIDNet {tag} fwd max abs diff vs reference: {d:.3e}")
    loss = (y - gt).abs().mean()
    assert abs(loss.item() - float(g[f"model_{tag}_loss"])) < 1e-5
    loss.backward()
    n_checked = 0
    for n, prm in m.named_parameters():
        if n.startswith("I_LCA5."):
            assert prm.grad is None          # dead block in the reference too
            continue
        key = f"model_{tag}_g.{n}"
        if key in g.files:
            grad_ok(prm.grad, g[key], rel=2e-4, what=f"d{n}")
            n_checked += 1
        s = g[f"model_{tag}_gsum.{n}"]
        assert abs(prm.grad.double().sum().item() - s[0]) <= 2e-4 * s[1] + 1e-9, n
    assert n_checked > 30


def test_cidnet_norm_option_golden(golden, dev):
    """CIDNet(norm=True): the LayerNorm option of the down / up blocks (net/CIDNet.py:12, net/transformer_utils.py:44-48,
    66-70; VERDICT r2 missing #4) against the reference's output and every gradient (tests/golden/round3.npz)"""
    import hvi_cidnet_amd as P
    g = golden("round3")
    chans = (12, 12, 24, 48)
    p = O.make_params(13, channels=chans, norm=True)
    m = P.CIDNet(channels=list(chans), norm=True)
    assert len(m.state_dict()) == 191 + 24
    load(m, p)
    m.to(dev)
    y = m(_t(g["norm_x"], dev))
    out_ok(y, g["norm_out"], 1e-4, "CIDNet(norm=True) fwd")
    (y - _t(g["norm_gt"], dev)).abs().mean().backward()
    n = 0
    for name, prm in m.named_parameters():
        if name.startswith("I_LCA5."):
            assert prm.grad is None
            continue
        grad_ok(prm.grad, g[f"norm_g.{name}"], rel=2e-4, what=f"norm=True d{name}")
        n += 1
    assert n == 191 + 24 - 13


def test_cidnet_config1_400x600(golden, dev):
    """config 1 of BASELINE.json: one 1x3x400x600 forward, default-style parameters"""
    import hvi_cidnet_amd as P
    g = golden("model")
    m = P.CIDNet()
    load(m, O.make_params(5, jitter=False))
    m.to(dev).eval()
    x = O.synthetic_batch(61, (1, 3, 400, 600)).to(dev)
    with torch.no_grad():
        y = m(x)
    out_ok(y[:, :, ::16, ::16], g["model_c1_out_strided"], 1e-4, "c1 strided")
    s = g["model_c1_out_sums"]
    yd = y.double()
    assert abs(yd.sum().item() - s[0]) <= 1e-5 * s[1]
    assert abs((yd ** 2).sum().item() - s[2]) <= 1e-5 * s[2]


def test_cidnet_full_size_batch_properties(dev):
    """BASELINE.json configs[1] size (400x600, full width): size-independent properties that hold for the reference
    because nothing in CIDNet mixes samples (per-pixel LayerNorm, per-sample attention, shared PReLU slope):
    (1) a sample's output does not depend on its batch; (2) the gradient of a summed loss over a batch is the sum of
    the per-sample gradients.  Exercises every kernel's full-size tiling (partial tiles, W % 4 != 0 levels, split-K,
    multi-round grids) where the golden fixtures only cover reduced sizes."""
    import hvi_cidnet_amd as P
    m = P.CIDNet()
    load(m, O.make_params(7))
    m.to(dev)
    m.two_streams = True
    x = O.synthetic_batch(91, (8, 3, 400, 600)).to(dev)
    with torch.no_grad():
        y8 = m(x)
        for i in (0, 5):
            yi = m(x[i:i + 1].contiguous())
            d = (y8[i:i + 1] - yi).abs().max().item()
            assert d <= 2e-6, f"sample {i}: batch-of-8 output differs from single-sample output by {d:.3e}"
    gt = O.synthetic_batch(92, (2, 3, 400, 600)).to(dev)

    def grads(xs, gs):
        for p in m.parameters():
            p.grad = None
        (m(xs) - gs).abs().sum().backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    g_ab = grads(x[:2].contiguous(), gt)
    g_a = grads(x[0:1].contiguous(), gt[0:1].contiguous())
    g_b = grads(x[1:2].contiguous(), gt[1:2].contiguous())
    assert len(g_ab) == 191 - 13                                   # I_LCA5.* stays dead at every size
    for n, g in g_ab.items():
        ref = g_a[n].double() + g_b[n].double()
        d = (g.double() - ref).abs().max().item()
        tol = 2e-4 * ref.abs().max().item() + 1e-6
        assert d <= tol, f"d{n}: batch gradient differs from the sum of per-sample gradients by {d:.3e} > {tol:.3e}"


def test_module_surface(dev):
    import hvi_cidnet_amd as P
    m = P.CIDNet().to(dev)
    assert list(m.state_dict().keys()) == list(O.param_shapes().keys())
    assert sum(p.numel() for p in m.parameters()) == 1975569
    with pytest.raises(RuntimeError, match="multiples of 8"):
        m(torch.rand(1, 3, 20, 24, device=dev))
    m.trans.gated, m.trans.gated2, m.trans.alpha, m.trans.alpha_s = True, True, 0.8, 1.3     # eval.py:46-55
    with torch.no_grad():
        y = m(torch.rand(1, 3, 16, 24, device=dev))
    assert y.shape == (1, 3, 16, 24) and torch.isfinite(y).all()
    assert abs(m.trans.this_k - 0.2) < 1e-6


def test_spatial_attention_golden(golden, dev):
    import hvi_cidnet_amd as P
    g = golden("mssa")
    m = P.SpatialAttention()
    m.load_state_dict({"conv1.weight": _t(g["sa_w"])})
    m.to(dev)
    x = _t(g["sa_x"], dev).requires_grad_(True)
    y = m(x)
    out_ok(y, g["sa_out"], 1e-5, "SA fwd")
    y.backward(_t(g["sa_gout"], dev))
    grad_ok(x.grad, g["sa_gx"], what="SA gx")
    grad_ok(m.conv1.weight.grad, g["sa_gw"], what="SA gw")


def test_cidnet_mssa_golden(golden, dev):
    """config 5 (MSSA): 197 tensors, I_LCA5 live, SpatialAttention after every up block"""
    import hvi_cidnet_amd as P
    g = golden("mssa")
    chans = (12, 12, 24, 48)
    m = P.CIDNet_MSSA(channels=list(chans))
    load(m, O.make_params(5, channels=chans, variant="mssa"))
    m.to(dev)
    y = m(_t(g["model_x"], dev))
    d = out_ok(y, g["model_out"], 1e-4, "CIDNet_MSSA fwd")
    print(f"CIDNet_MSSA fwd max abs diff vs reference: {d:.3e}")
    (y - _t(g["model_gt"], dev)).abs().mean().backward()
    for n, prm in m.named_parameters():
        assert prm.grad is not None, n
        grad_ok(prm.grad, g[f"model_g.{n}"], rel=2e-4, what=f"d{n}")
    assert len(m.state_dict()) == 197


def test_tnsm_block_golden(golden, dev):
    """one TrainableNoiseSuppression block (net/TNSM.py) incl. its noise-map output, all gradients"""
    import hvi_cidnet_amd as P
    g = golden("tnsm")
    chans = (12, 12, 24, 48)
    m = P.HV_TNSM(chans[1], 2)
    load(m, O.make_params(9, channels=chans, variant="tnsm"), "HV_TNSM1.")
    m.to(dev)
    x = _t(g["blk_x"], dev).requires_grad_(True)
    y = _t(g["blk_y"], dev).requires_grad_(True)
    z, nm = m(x, y)
    out_ok(z, g["blk_out"], 1e-5, "TNSM fwd")
    out_ok(nm, g["blk_noise"], 1e-5, "TNSM noise map")
    ((z * _t(g["blk_gout"], dev)).sum() + (nm * _t(g["blk_gnoise"], dev)).sum()).backward()
    grad_ok(x.grad, g["blk_gx"], what="TNSM gx")
    grad_ok(y.grad, g["blk_gy"], what="TNSM gy")
    for n, prm in m.named_parameters():
        grad_ok(prm.grad, g[f"blk_g.{n}"], rel=2e-4, what=f"TNSM d{n}")


def test_cidnet_tnsm_golden(golden, dev):
    """config 5 (TNSM): 468 tensors; training mode returns (rgb, fused_noise), eval mode (rgb, None)"""
    import hvi_cidnet_amd as P
    g = golden("tnsm")
    chans = (12, 12, 24, 48)
    m = P.CIDNet_TNSM(channels=list(chans))
    load(m, O.make_params(5, channels=chans, variant="tnsm"))
    m.to(dev).train()
    y, fz = m(_t(g["model_x"], dev))
    d = out_ok(y, g["model_out"], 1e-4, "CIDNet_TNSM rgb")
    out_ok(fz, g["model_noise"], 1e-5, "CIDNet_TNSM fused noise")
    print(f"CIDNet_TNSM fwd max abs diff vs reference: {d:.3e}")
    ((y - _t(g["model_gt"], dev)).abs().mean() + 0.1 * fz.mean()).backward()
    dead = set(g["model_dead"].tolist())
    # TNSM's attention is NOT L2-normalised (net/TNSM.py:98-104): its logits are raw dot products over H*W and
    # the softmax sits close to saturation, which amplifies fp32 summation-order differences in every upstream
    # gradient.  The reference's own fp32 gradients are ~2e-3 (of each tensor's max) away from the fp64 truth
    # (20% for the nearly-cancelling PReLU slopes), so the bar is: our error against fp64 must not exceed
    # twice the reference's own error (+1e-4 of the tensor's max).  The forward bars stay at 1e-4 / 1e-5.
    for n, prm in m.named_parameters():
        if n in dead:
            assert prm.grad is None, n
            continue
        g64 = _t(g[f"model_g64.{n}"]).double()
        ref_err = (_t(g[f"model_g.{n}"]).double() - g64).abs().max().item()
        our_err = (prm.grad.detach().cpu().double() - g64).abs().max().item()
        assert our_err <= 2.0 * ref_err + 1e-4 * g64.abs().max().item() + 1e-9, (n, our_err, ref_err)
    m.eval()
    with torch.no_grad():
        ye, fe = m(_t(g["model_x"], dev))
    assert fe is None and ye.shape == y.shape


@pytest.mark.parametrize("shape,quantised", [((1, 3, 8, 8), False), ((2, 3, 16, 24), True), ((1, 3, 40, 56), False),
                                             ((3, 3, 24, 104), False), ((1, 3, 104, 40), True), ((1, 3, 8, 200), False)])
def test_cidnet_shape_sweep_vs_oracle(dev, shape, quantised):
    """ragged and degenerate sizes against the oracle run live on the host: planes narrower than one 4-pixel lane
    group at the deeper levels (8x8 input => 4x4, 2x2, 1x1 planes: the per-element loaders, bilinear resampling
    to and from a single pixel), widths that are not multiples of 4 after downsampling, batch 3, uint8-like inputs
    with many channel ties.  Forward at the north-star bar (1e-4 abs) against the fp32 oracle; gradients of a smooth
    objective against the oracle evaluated in fp64 (the fp32 host evaluation is itself up to 1e-2 off on these
    small planes, tools/diag_shapes.py), bar 5e-4 of each tensor's max + 5e-6 (scalar gradients of tiny magnitude
    -- PReLU slopes, temperatures -- carry fp32 summation noise of that absolute size)."""
    import hvi_cidnet_amd as P
    chans = (12, 12, 24, 48)
    p = O.make_params(11, channels=chans)
    m = P.CIDNet(channels=list(chans))
    load(m, p)
    m.to(dev)
    x = O.synthetic_batch(101 + shape[2], shape, quantised=quantised)
    r = O.synthetic_batch(202 + shape[3], shape) - 0.5
    y = m(x.to(dev))
    (y * r.to(dev)).sum().backward()
    with torch.no_grad():
        out_ok(y, O.cidnet_forward(p, x).numpy(), 1e-4, f"fwd {shape}")
    po = O.params_to(p, dtype=torch.float64, requires_grad=True)
    yo = O.cidnet_forward(po, x.double())
    (yo * r.double()).sum().backward()
    out_ok(y, yo.detach().numpy(), 1e-4, f"fwd vs fp64 {shape}")
    n_checked = 0
    for n, prm in m.named_parameters():
        if n.startswith("I_LCA5."):
            assert prm.grad is None
            continue
        g = po[n].grad
        d = (prm.grad.detach().cpu().double() - g).abs().max().item()
        tol = 5e-4 * g.abs().max().item() + 5e-6
        assert d <= tol, f"d{n} {shape}: max diff {d:.3e} > {tol:.3e}"
        n_checked += 1
    assert n_checked == 191 - 13


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_two_streams_bit_identical_to_single_stream(dev, storage):
    """The two-branch schedule (cidnet.CIDNet._par / _lca_pair: I branch and HV branch on two streams) only reorders
    independent kernels, so forward output and every parameter gradient must equal the single-stream run bit for bit, at
    the full 8x400x600 size where the branches really overlap on the device -- in both storage modes."""
    import hvi_cidnet_amd as P
    m = P.CIDNet()
    load(m, O.make_params(7))
    m.to(dev)
    x = O.synthetic_batch(91, (8, 3, 400, 600)).to(dev)

    def run(two):
        m.two_streams = two
        for q in m.parameters():
            q.grad = None
        y = m(x)
        y.square().mean().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), torch.cat([q.grad.flatten() for q in m.parameters() if q.grad is not None]).clone()

    P.set_storage_dtype(storage)
    try:
        y1, g1 = run(False)
        for _ in range(2):
            y2, g2 = run(True)
            assert torch.equal(y1, y2)
            assert torch.equal(g1, g2)
    finally:
        P.set_storage_dtype("f32")


def _dump_rerun_mismatch(tag, ys, y32):
    """Failing-run evidence: positions (b, c, y, x) and values of every element that differs
    between reruns, written under gpurun_out/ (merged back from the GPU box) -- copy it to profiles/ when it appears."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rec = {}
    lines = []
    for j in (1, 2):
        idx = (ys[0] != ys[j]).nonzero().cpu()
        rec[f"idx_0_vs_{j}"] = idx.numpy()
        for t, nm in ((ys[0], "run0"), (ys[j], f"run{j}"), (y32, "fp32")):
            rec[f"{nm}_at_0_vs_{j}"] = t[tuple(idx.T)].cpu().numpy() if len(idx) else np.zeros(0, np.float32)
        lines.append(f"run0 vs run{j}: {len(idx)} elements differ")
        for r in range(min(len(idx), 200)):
            b, c, y, x = idx[r].tolist()
            lines.append(f"  (b={b}, c={c}, y={y}, x={x})  run0={ys[0][b, c, y, x].item():.9e} run{j}={ys[j][b, c, y, x].item():.9e} "
                         f"fp32={y32[b, c, y, x].item():.9e}")
    np.savez(os.path.join(out, f"fail_{tag}_rerun_mismatch.npz"), **rec)
    with open(os.path.join(out, f"fail_{tag}_rerun_mismatch.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def test_cidnet_with_bf16x3_conv(dev):
    """The split-product conv (ops.CONV3_BF16X3, csrc/conv3x.hip: the default) inside the whole model at 8x3x400x600 with
    both branch streams on: output within fp32 rounding of the fp32-MFMA conv path (2e-6 absolute on outputs in [0, 1])
    and three runs bit-identical.  A mismatch writes the differing elements' positions and values under gpurun_out/
    (round 2's kernel, csrc/conv3s.hip, failed this rerun check once in seven runs; it was replaced)."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    m = P.CIDNet()
    load(m, O.make_params(7))
    m.to(dev)
    m.two_streams = True
    x = O.synthetic_batch(91, (8, 3, 400, 600)).to(dev)
    old = dict(ops.CONV3_BF16X3)
    try:
        with torch.no_grad():
            ops.CONV3_BF16X3["on"] = False
            y32 = m(x)
            ops.CONV3_BF16X3["on"] = True
            ys = [m(x) for _ in range(3)]
        torch.cuda.synchronize()
    finally:
        ops.CONV3_BF16X3.update(old)
    d01, d02 = (ys[0] - ys[1]).abs(), (ys[0] - ys[2]).abs()
    if not (torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])):
        _dump_rerun_mismatch("bf16x3_conv", ys, y32)
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), (
        f"reruns differ: {int((d01 > 0).sum())} / {int((d02 > 0).sum())} elements, max {d01.max().item():.3e} / {d02.max().item():.3e}; "
        f"vs fp32 path: {[(y - y32).abs().max().item() for y in ys]}")
    assert (ys[0] - y32).abs().max().item() <= 2e-6


def test_cidnet_with_bf16x3_pw_conv(dev):
    """Opt-in split-product 1x1 conv (ops.PW_BF16X3, csrc/pws.hip) inside the whole model, forward and backward: output and
    gradients within fp32 rounding of the default path, and reproducible."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    m = P.CIDNet()
    load(m, O.make_params(7))
    m.to(dev)
    x = O.synthetic_batch(91, (2, 3, 400, 600)).to(dev)

    def run():
        for q in m.parameters():
            q.grad = None
        y = m(x)
        y.square().mean().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), {n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}

    old = dict(ops.PW_BF16X3)
    try:
        ops.PW_BF16X3["on"] = False
        y32, g32 = run()
        ops.PW_BF16X3["on"] = True
        ys, gs = run()
        ys2, gs2 = run()
    finally:
        ops.PW_BF16X3.update(old)
    assert torch.equal(ys, ys2) and all(torch.equal(gs[n], gs2[n]) for n in gs)
    assert (ys - y32).abs().max().item() <= 5e-6
    for n in g32:
        scale = g32[n].abs().max().item() + 1e-12
        assert (gs[n] - g32[n]).abs().max().item() <= 2e-3 * scale, n
