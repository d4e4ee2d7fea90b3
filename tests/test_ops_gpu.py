"""GPU parity of the individual C-ABI kernels (K3-K11) against plain fp32 PyTorch CPU references of the
same op on seeded inputs, including ragged sizes (pixel tails, channel tails, unaligned rows)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cidnet_oracle as O

pytestmark = pytest.mark.gpu


def rnd(seed, shape, scale=1.0):
    return (O.synthetic_batch(seed, shape) - 0.5) * (2 * scale)


def close(a, b, rel=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    d = (a - b).abs().max().item()
    tol = rel * b.abs().max().item() + 1e-6
    assert d <= tol, f"{what}: max diff {d:.3e} > tol {tol:.3e}"


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 36, 36, 8, 12), (1, 95, 36, 13, 17), (2, 36, 190, 20, 30), (1, 144, 766, 5, 15),
                                         (2, 383, 144, 10, 15), (1, 3, 7, 3, 3), (2, 72, 72, 100, 150),
                                         (8, 36, 190, 200, 300), (4, 36, 36, 400, 600), (8, 72, 36, 200, 300),  # multi-tile blocks
                                         (8, 766, 144, 50, 75), (8, 144, 766, 50, 75), (2, 288, 288, 50, 75), (3, 383, 144, 50, 75),
                                         (2, 100, 72, 9, 13)])  # small planes: split-K kernel, K > 384 in two passes
def test_pw_conv_fwd_dgrad_wgrad(dev, B, Ci, Co, H, W):
    from hvi_cidnet_amd import ops
    x, w, r = rnd(1, (B, Ci, H, W)), rnd(2, (Co, Ci, 1, 1), 0.3), rnd(3, (B, Co, H, W))
    gy = rnd(4, (B, Co, H, W))
    HW = H * W
    xd, wd, rd, gyd = (t.to(dev) for t in (x, w, r, gy))
    y = torch.empty_like(rd)
    ops.pw_conv(xd, 0, Ci * HW, wd, 0, 0, Ci, 1, y, 0, Co * HW, B, Co, Ci, HW, res=rd, r_bs=Co * HW)
    close(y, F.conv2d(x, w) + r, what="fwd+res")
    dx = torch.empty_like(xd)
    ops.pw_conv(gyd, 0, Co * HW, wd, 0, 0, 1, Ci, dx, 0, Ci * HW, B, Ci, Co, HW)
    close(dx, F.conv_transpose2d(gy, w), what="dgrad")
    gw = torch.empty_like(wd)
    ops.pw_wgrad(gyd, 0, Co * HW, xd, 0, Ci * HW, gw, 0, Ci, B, Co, Ci, HW)
    ref = torch.einsum("bmhw,bnhw->mn", gy.double(), x.double())
    close(gw.reshape(Co, Ci), ref, what="wgrad")
    gwb = torch.empty((B, Co, Ci), device=dev)
    ops.pw_wgrad(gyd, 0, Co * HW, xd, 0, Ci * HW, gwb, 0, Ci, B, Co, Ci, HW, per_sample=True)
    close(gwb, torch.einsum("bmhw,bnhw->bmn", gy.double(), x.double()), what="wgrad per-sample")


def test_pw_conv_per_sample_and_slices(dev):
    """per-sample weights + reading / writing channel slices of wider tensors (the qkv layout)"""
    from hvi_cidnet_amd import ops
    B, C, H, W = 3, 24, 9, 11
    HW = H * W
    qkv = rnd(5, (B, 3 * C, H, W))
    M = rnd(6, (B, C, C), 0.4)
    out = torch.zeros((B, 2 * C, H, W), device=dev)
    ops.pw_conv(qkv.to(dev), 2 * C * HW, 3 * C * HW, M.to(dev), 0, C * C, C, 1, out, C * HW, 2 * C * HW, B, C, C, HW)
    ref = torch.einsum("bmk,bkhw->bmhw", M, qkv[:, 2 * C:])
    close(out[:, C:], ref, what="per-sample slice")
    assert out[:, :C].abs().max().item() == 0.0


@pytest.mark.parametrize("B,C,H,W", [(2, 36, 8, 12), (1, 144, 50, 75), (2, 12, 7, 9), (1, 72, 100, 150)])
def test_layernorm(dev, B, C, H, W):
    from hvi_cidnet_amd import ops
    x = rnd(7, (B, C, H, W), 2.0)
    w, b = 1 + rnd(8, (C,), 0.2), rnd(9, (C,), 0.1)
    gy = rnd(10, (B, C, H, W))
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = O.layernorm_cf(xr, wr, br)
    yr.backward(gy)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.LayerNormCFFn.apply(xd, wd, bd, 1e-6)
    y.backward(gy.to(dev))
    close(y, yr, what="fwd")
    close(xd.grad, xr.grad, what="gx")
    close(wd.grad, wr.grad, what="gw")
    close(bd.grad, br.grad, what="gb")


@pytest.mark.parametrize("B,C,H,W", [(2, 36, 8, 12), (1, 72, 5, 6), (2, 144, 5, 7), (1, 144, 6, 6), (3, 36, 10, 10), (1, 36, 7, 9)])
@pytest.mark.parametrize("use", ["both", "a", "b", "res"])
def test_layernorm_dual(dev, B, C, H, W, use):
    """ops.LayerNormDualFn: two LayerNorm modules on one tensor in one forward and one backward pass (the x-norm of an LCA
    block and the y-norm of its partner).  Outputs bit-equal to the single-module kernel; all gradients (input incl. the
    residual hand-over, both modules' weight and bias) against the oracle's autograd, also when one output is unused; the last
    shape has no dual kernel (H*W odd for C = 36) and takes the two-pass route through LayerNorm.forward_dual."""
    from hvi_cidnet_amd import ops
    import hvi_cidnet_amd as P
    x = rnd(51, (B, C, H, W), 2.0)
    wa, ba, wb, bb = 1 + rnd(52, (C,), 0.2), rnd(53, (C,), 0.1), 1 + rnd(54, (C,), 0.3), rnd(55, (C,), 0.2)
    ga, gb_, gr = rnd(56, (B, C, H, W)), rnd(57, (B, C, H, W)), rnd(58, (B, C, H, W))
    xr, war, bar, wbr, bbr = (t.clone().requires_grad_(True) for t in (x, wa, ba, wb, bb))
    ya_r, yb_r = O.layernorm_cf(xr, war, bar), O.layernorm_cf(xr, wbr, bbr)
    na, nb = P.LayerNorm(C).to(dev), P.LayerNorm(C).to(dev)
    with torch.no_grad():
        na.weight.copy_(wa); na.bias.copy_(ba); nb.weight.copy_(wb); nb.bias.copy_(bb)
    xd = x.to(dev).requires_grad_(True)
    assert bool(ops.ln_dual_supported(xd)) == (not (C == 36 and (H * W) % 4) and not (C == 72 and (H * W) % 2))
    ya, yb, xres = na.forward_dual(xd, nb)
    assert torch.equal(ya, ops.LayerNormCFFn.apply(xd.detach(), na.weight.detach(), na.bias.detach(), 1e-6))
    assert torch.equal(yb, ops.LayerNormCFFn.apply(xd.detach(), nb.weight.detach(), nb.bias.detach(), 1e-6))
    assert xres.data_ptr() == xd.data_ptr()
    outs, refs, grads = [], [], []
    if use in ("both", "a"):
        outs.append(ya); refs.append(ya_r); grads.append(ga)
    if use in ("both", "b"):
        outs.append(yb); refs.append(yb_r); grads.append(gb_)
    if use in ("both", "res"):
        outs.append(xres); refs.append(xr * 1.0); grads.append(gr)
    torch.autograd.backward(refs, grads)
    torch.autograd.backward(outs, [g.to(dev) for g in grads])
    close(xd.grad, xr.grad, what="gx")
    for mod, wr_, br_, used in ((na, war, bar, use in ("both", "a")), (nb, wbr, bbr, use in ("both", "b"))):
        if used:
            close(mod.weight.grad, wr_.grad, what="gw")
            close(mod.bias.grad, br_.grad, what="gb")
        else:
            assert mod.weight.grad is None or float(mod.weight.grad.abs().max()) == 0.0


@pytest.mark.parametrize("B,M,N,HW", [(2, 190, 36, 600), (1, 36, 95, 1000), (3, 36, 36, 96), (2, 72, 36, 404), (1, 190, 36, 32),
                                      (5, 181, 33, 260), (1, 36, 95, 12)])
def test_pw_bwd_fused(dev, B, M, N, HW):
    """csrc/pwb.hip: data gradient and weight gradient of a 1x1 conv in one kernel, against fp64 and no less accurate than
    the two separate kernels; ragged last chunk (HW not a multiple of 32), channel counts inside a tile, planes smaller
    than a chunk; NaN-prefilled outputs, reruns bit-identical"""
    from hvi_cidnet_amd import ops
    assert ops._raw("cidnet_pw_bwd_fused_supported", M, N, HW)
    gy, x, w = rnd(61, (B, M, HW)), rnd(62, (B, N, HW)), rnd(63, (M, N), 0.3)
    gx_ref = torch.einsum("mn,bmp->bnp", w.double(), gy.double())
    dw_ref = torch.einsum("bmp,bnp->mn", gy.double(), x.double())
    gyd, xd, wd = gy.to(dev), x.to(dev), w.to(dev)
    outs = []
    for _ in range(2):
        gx = torch.full((B, N, HW), float("nan"), device=dev)
        dw = torch.full((M, N), float("nan"), device=dev)
        ops.pw_bwd_fused(gyd, xd, wd, gx, dw, B, M, N, HW)
        outs.append((gx, dw))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    close(outs[0][0], gx_ref.float(), what="gx")
    close(outs[0][1], dw_ref.float(), what="dw")
    gx2, dw2 = torch.empty_like(outs[0][0]), torch.empty_like(outs[0][1])
    ops.pw_conv(gyd, 0, M * HW, wd, 0, 0, 1, N, gx2, 0, N * HW, B, N, M, HW)        # data gradient: transposed strides
    ops.pw_wgrad(gyd, 0, M * HW, xd, 0, N * HW, dw2, 0, N, B, M, N, HW)
    for a, b, ref in ((outs[0][0], gx2, gx_ref), (outs[0][1], dw2, dw_ref)):
        ea, eb = (a.double().cpu() - ref).abs().max().item(), (b.double().cpu() - ref).abs().max().item()
        assert ea <= 2.0 * eb + 1e-6 * ref.abs().max().item(), (ea, eb)


@pytest.mark.parametrize("B,C,H,W", [(2, 12, 9, 13), (1, 190, 20, 30), (1, 6, 50, 75), (2, 3, 17, 66)])
def test_dw3x3(dev, B, C, H, W):
    from hvi_cidnet_amd import ops
    x, w, add = rnd(11, (B, C, H, W)), rnd(12, (C, 1, 3, 3), 0.5), rnd(13, (B, C, H, W))
    cs = C // 3
    xd, wd, ad = x.to(dev), w.to(dev), add.to(dev)
    w1, w2 = wd[:cs].contiguous(), wd[cs:].contiguous()
    y = torch.empty_like(xd)
    ops.dw3x3(xd, w1, w2, cs, y, B, C, H, W)
    close(y, F.conv2d(x, w, padding=1, groups=C), what="fwd")
    ops.dw3x3(xd, w1, w2, cs, y, B, C, H, W, flip=True, addend=ad)
    close(y, F.conv_transpose2d(x, w, padding=1, groups=C) + add, what="dgrad+add")
    g = rnd(14, (B, C, H, W))
    g1, g2 = torch.empty_like(w1), torch.empty_like(w2)
    ops.dw3x3_wgrad(xd, g.to(dev), g1, g2, cs, B, C, H, W)
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, padding=1, groups=C).backward(g)
    close(torch.cat([g1, g2]), wr.grad, what="wgrad")


@pytest.mark.parametrize("B,Ci,Co,H,W,rep", [(2, 36, 36, 16, 24, False), (1, 72, 144, 10, 15, False), (1, 12, 24, 9, 70, False),
                                             (2, 3, 36, 16, 24, True), (2, 36, 2, 13, 19, True), (1, 1, 36, 8, 8, True),
                                             (1, 36, 1, 8, 72, True), (1, 144, 72, 50, 75, False),
                                             (2, 4, 36, 21, 37, False), (1, 36, 3, 40, 50, False), (2, 3, 36, 40, 50, True),
                                             (1, 36, 4, 35, 21, True), (1, 2, 20, 33, 600, True)])
def test_conv3x3(dev, B, Ci, Co, H, W, rep):
    from hvi_cidnet_amd import ops
    x, w = rnd(21, (B, Ci, H, W)), rnd(22, (Co, Ci, 3, 3), 0.3)
    gy = rnd(23, (B, Co, H, W))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = O.rep_conv3x3(xr, wr) if rep else F.conv2d(xr, wr, padding=1)
    yr.backward(gy)
    if rep:
        xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = ops.RepConv3x3Fn.apply(xd, wd)
        y.backward(gy.to(dev))
        close(y, yr, what="fwd")
        close(xd.grad, xr.grad, what="dgrad (replicate)")
        close(wd.grad, wr.grad, what="wgrad (replicate)")
    else:
        xd, wd, gyd = x.to(dev), w.to(dev), gy.to(dev)
        y = torch.empty_like(gyd)
        ops.conv3x3(xd, wd, y, B, Co, Ci, H, W, 9 * Ci, 9)
        close(y, yr, what="fwd")
        dx = torch.empty_like(xd)
        ops.conv3x3(gyd, wd, dx, B, Ci, Co, H, W, 9, 9 * Ci, flip=True)
        close(dx, xr.grad, what="dgrad")
        gw = torch.empty_like(wd)
        ops.conv3x3_wgrad(gyd, xd, gw, B, Co, Ci, H, W)
        close(gw, wr.grad, what="wgrad")


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 36, 36, 16, 24), (1, 72, 144, 10, 15), (1, 144, 72, 50, 75), (3, 36, 72, 13, 37),
                                         (1, 36, 108, 7, 5), (2, 72, 36, 9, 33), (1, 36, 36, 1, 4), (5, 36, 72, 4, 32),
                                         (2, 36, 36, 70, 130), (1, 36, 36, 3, 31), (9, 36, 36, 5, 64)])
def test_conv3x3_wgrad_bf16x3(dev, B, Ci, Co, H, W):
    """csrc/conv3xw.hip against fp64 (ragged tiles, several 36-channel chunks on either side, planes smaller than one tile,
    blocks with one and with many tiles), twice into NaN-filled outputs: every element written, run-to-run identical"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    assert ops.CONV3_WGRAD_BF16X3["on"] and lib().raw("cidnet_conv3x3_wgrad_bf16x3_supported")(Co, Ci, H, W)
    x, gy = rnd(41, (B, Ci, H, W)), rnd(42, (B, Co, H, W))
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), gy.double(), padding=1)
    xd, gyd = x.to(dev), gy.to(dev)
    outs = []
    for _ in range(2):
        gw = torch.full((Co, Ci, 3, 3), float("nan"), device=dev)
        ops.conv3x3_wgrad(gyd, xd, gw, B, Co, Ci, H, W)
        outs.append(gw)
    close(outs[0], ref.float(), what="wgrad bf16x3")
    assert torch.equal(outs[0], outs[1])
    # no less accurate than the fp32-MFMA kernel
    ops.CONV3_WGRAD_BF16X3["on"] = False
    try:
        g32 = torch.empty_like(outs[0])
        ops.conv3x3_wgrad(gyd, xd, g32, B, Co, Ci, H, W)
    finally:
        ops.CONV3_WGRAD_BF16X3["on"] = True
    e3, e32 = (outs[0].double().cpu() - ref).abs().max().item(), (g32.double().cpu() - ref).abs().max().item()
    assert e3 <= 2.0 * e32 + 1e-6 * ref.abs().max().item(), (e3, e32)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 36, 36, 16, 24), (1, 72, 36, 10, 15), (1, 12, 24, 9, 70), (1, 24, 12, 4, 4), (1, 144, 72, 25, 37)])
def test_conv3x3_addend_and_down_skip(dev, B, Ci, Co, H, W):
    """cidnet_conv3x3_add == conv + addend bitwise (same kernel body, one more load in the epilogue), and the
    NormDownsample-with-skip function gives the gradients of the plain block plus the skip's gradient"""
    from hvi_cidnet_amd import ops
    x, w, r = rnd(31, (B, Ci, H, W)).to(dev), rnd(32, (Co, Ci, 3, 3), 0.3).to(dev), rnd(33, (B, Co, H, W)).to(dev)
    y0, y1 = torch.empty_like(r), torch.empty_like(r)
    ops.conv3x3(x, w, y0, B, Co, Ci, H, W, 9 * Ci, 9)
    ops.conv3x3(x, w, y1, B, Co, Ci, H, W, 9 * Ci, 9, addend=r)
    assert torch.equal(y1, y0 + r)
    if H % 2 or W % 2:
        return
    slope = torch.tensor([0.2], device=dev)
    go, gs = rnd(34, (B, Co, H // 2, W // 2)).to(dev), rnd(35, (B, Ci, H, W)).to(dev)
    xa, wa, sa = (t.clone().requires_grad_(True) for t in (x, w, slope))
    ya = ops.DownFn.apply(xa, wa, sa)
    ya.backward(go)
    xb, wb, sb = (t.clone().requires_grad_(True) for t in (x, w, slope))
    yb, xs = ops.DownResFn.apply(xb, wb, sb)
    assert xs.data_ptr() == xb.data_ptr()
    torch.autograd.backward([yb, xs], [go, gs])
    assert torch.equal(ya, yb)
    close(xb.grad, xa.grad + gs, what="dgrad + skip")
    assert torch.equal(wb.grad, wa.grad) and torch.equal(sb.grad, sa.grad)


@pytest.mark.parametrize("B,C,Hi,Wi,Ho,Wo", [(2, 5, 16, 24, 8, 12), (1, 3, 17, 23, 8, 11), (2, 4, 8, 12, 16, 24), (1, 2, 50, 75, 100, 150)])
def test_bilinear_bwd_and_down(dev, B, C, Hi, Wi, Ho, Wo):
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    x = rnd(31, (B, C, Hi, Wi))
    g = rnd(32, (B, C, Ho, Wo))
    xr = x.clone().requires_grad_(True)
    yr = O.bilinear_ac(xr, (Ho, Wo))
    yr.backward(g)
    din = torch.empty((B, C, Hi, Wi), device=dev)
    gd, xd = g.to(dev), x.to(dev)          # keep the device tensors alive while raw pointers are in flight
    ops.bilinear_bwd(gd, din, B, C, Hi, Wi, Ho, Wo)
    close(din, xr.grad, what="bilinear adjoint")
    if Ho == Hi // 2 and Wo == Wi // 2:
        slope = torch.tensor([0.17])
        sd = slope.to(dev)
        pre = torch.empty((B, C, Ho, Wo), device=dev)
        out = torch.empty_like(pre)
        lib().call("cidnet_down_prelu_fwd", ops._p(xd), ops._p(sd), ops._p(pre), ops._p(out), B, C, Hi, Wi, ops._stream())
        close(pre, yr, what="down")
        close(out, F.prelu(yr, slope), what="down+prelu")


@pytest.mark.parametrize("B,h,H,W", [(2, 5, 9, 13), (1, 31, 16, 24), (2, 127, 4, 6), (1, 7, 3, 3), (1, 63, 8, 12), (1, 95, 40, 300)])
def test_iel_gate_dw_bwd_fused_equals_unfused(dev, B, h, H, W):
    """fused gate + dwconv1/2 backward == gate_bwd followed by the depthwise backward"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    u, dg = rnd(51, (B, 2 * h, H, W), 1.5).to(dev), rnd(52, (B, h, H, W)).to(dev)
    w1, w2 = rnd(53, (h, 1, 3, 3), 0.5).to(dev), rnd(54, (h, 1, 3, 3), 0.5).to(dev)
    da, ds = torch.empty_like(u), torch.empty_like(u)
    lib().call("cidnet_iel_gate_bwd", ops._p(u), ops._p(w1), ops._p(w2), ops._p(dg), ops._p(da), ops._p(ds), B, h, H, W, ops._stream())
    du_ref, g1_ref, g2_ref = torch.empty_like(u), torch.empty_like(w1), torch.empty_like(w2)
    ops.dw3x3_bwd(u, da, w1, w2, h, du_ref, g1_ref, g2_ref, B, 2 * h, H, W, addend=ds)
    du, g1, g2 = torch.empty_like(u), torch.empty_like(w1), torch.empty_like(w2)
    n = lib().raw("cidnet_iel_gate_dw_bwd_ws_floats")(B, h, H, W)
    ws = torch.empty(n, device=dev)
    lib().call("cidnet_iel_gate_dw_bwd", ops._p(u), ops._p(w1), ops._p(w2), ops._p(dg), ops._p(du), ops._p(g1), ops._p(g2), ops._p(ws),
               n, B, h, H, W, ops._stream())
    close(du, du_ref, what="du")
    close(g1, g1_ref, what="gw1")
    close(g2, g2_ref, what="gw2")


@pytest.mark.parametrize("B,h,H,W", [(2, 5, 9, 13), (1, 31, 16, 24), (2, 127, 4, 6), (1, 7, 3, 3), (1, 63, 8, 12), (1, 95, 40, 300), (1, 3, 50, 75)])
def test_iel_dw_gate_fwd_fused_equals_unfused(dev, B, h, H, W):
    """fused dwconv + gate forward == depthwise conv followed by the gate kernel (both outputs)"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    pin = rnd(61, (B, 2 * h, H, W), 1.5).to(dev)
    w, w1, w2 = rnd(62, (2 * h, 1, 3, 3), 0.4).to(dev), rnd(63, (h, 1, 3, 3), 0.5).to(dev), rnd(64, (h, 1, 3, 3), 0.5).to(dev)
    u_ref = torch.empty_like(pin)
    ops.dw3x3(pin, w, None, 2 * h, u_ref, B, 2 * h, H, W)
    g_ref = torch.empty((B, h, H, W), device=dev)
    lib().call("cidnet_iel_gate_fwd", ops._p(u_ref), ops._p(w1), ops._p(w2), ops._p(g_ref), B, h, H, W, ops._stream())
    u, g = torch.empty_like(pin), torch.empty_like(g_ref)
    lib().call("cidnet_iel_dw_gate_fwd", ops._p(pin), ops._p(w), ops._p(w1), ops._p(w2), ops._p(u), ops._p(g), B, h, H, W, ops._stream())
    close(u, u_ref, what="u")
    close(g, g_ref, what="gate")


@pytest.mark.parametrize("shape", [(2, 36, 37, 51), (1, 12, 9, 13), (3, 24, 8, 8), (2, 72, 20, 30), (1, 48, 16, 24), (2, 36, 64, 100)])
@pytest.mark.parametrize("with_res", [False, True])
def test_iel_fused_forward_equals_chain(dev, shape, with_res):
    """tile-resident IEL forward (csrc/iel.hip: p, u, gate only in LDS) against the unfused chain the model runs
    (pw_conv -> dw+gate -> pw_conv), which the golden tests pin to the reference: out within 1e-5, u (the tensor the
    backward reads) within 2e-6; ragged sizes put tiles on every image border and on the first / last row of the tensor"""
    from hvi_cidnet_amd import ops
    B, C, H, W = shape
    h = int(C * 2.66)
    g = torch.Generator(device=dev).manual_seed(C + H)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    xn, res = rnd(B, C, H, W), (rnd(B, C, H, W) if with_res else None)
    w_in, w_dw = rnd(2 * h, C, 1, 1) / C ** 0.5, rnd(2 * h, 1, 3, 3) / 3
    w1, w2, w_out = rnd(h, 1, 3, 3) / 3, rnd(h, 1, 3, 3) / 3, rnd(C, h, 1, 1) / h ** 0.5
    assert ops._raw("cidnet_iel_fwd_supported", C, h) == 1
    old = dict(ops.IEL_FUSED)
    try:
        ops.IEL_FUSED.update(fwd=False)
        ref = ops.IELFn.apply(xn, res, w_in, w_dw, w1, w2, w_out, False)
        u_ref = torch.empty(B, 2 * h, H, W, device=dev)
        pin = torch.empty_like(u_ref)
        ops.pw_conv(xn, 0, C * H * W, w_in, 0, 0, C, 1, pin, 0, 2 * h * H * W, B, 2 * h, C, H * W)
        ops.dw3x3(pin, w_dw, None, 2 * h, u_ref, B, 2 * h, H, W)
        u = torch.empty_like(u_ref)
        out = torch.empty_like(xn)
        ops.lib().call("cidnet_iel_fwd", ops._p(xn), ops._p(res), ops._p(w_in), ops._p(w_dw), ops._p(w1), ops._p(w2), ops._p(w_out),
                       ops._p(u), ops._p(out), B, C, h, H, W, ops._stream())
        ops.IEL_FUSED.update(fwd=True)
        out2 = ops.IELFn.apply(xn, res, w_in, w_dw, w1, w2, w_out, False)       # the op-layer switch, inference form (no u)
    finally:
        ops.IEL_FUSED.update(old)
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out, out2)
    assert (u - u_ref).abs().max().item() <= 2e-6 * max(1.0, u_ref.abs().max().item())


def test_bf16_storage_mode_iel_chain(dev):
    """row J1: with set_storage_dtype('bf16') the IEL chain's hidden tensors are stored as bfloat16 (arithmetic fp32).
    Tolerance tier of this mode, against the fp32 path on the same inputs: output within 2e-2 of the output scale,
    every gradient within 3e-2 of its tensor's max (bf16 keeps 8 significant bits: 4e-3 per stored value, and a hidden
    value passes through 3-6 stored tensors); and the saved tensors really are bf16."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    B, C, H, W = 2, 36, 40, 56
    torch.manual_seed(3)
    m = P.IEL(C).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    res = torch.randn(B, C, H, W, device=dev)
    gout = torch.randn(B, C, H, W, device=dev)
    outs = {}
    try:
        for mode in ("f32", "bf16"):
            P.set_storage_dtype(mode)
            for p in m.parameters():
                p.grad = None
            xr = x.clone().requires_grad_(True)
            y = m(xr, residual=res)
            saved = [t for t in y.grad_fn.saved_tensors if t.dim() == 4 and t.shape[0] == B and tuple(t.shape[2:]) == (H, W) and t.shape[1] != C]
            assert saved and all(t.dtype == (torch.bfloat16 if mode == "bf16" else torch.float32) for t in saved)
            y.backward(gout)
            outs[mode] = (y.detach().clone(), xr.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        P.set_storage_dtype("f32")
    (y0, gx0, g0), (y1, gx1, g1) = outs["f32"], outs["bf16"]
    assert (y1 - y0).abs().max().item() <= 2e-2 * y0.abs().max().item()
    assert (gx1 - gx0).abs().max().item() <= 3e-2 * gx0.abs().max().item()
    for n in g0:
        assert (g1[n] - g0[n]).abs().max().item() <= 3e-2 * g0[n].abs().max().item(), n
    assert (y1 - y0).abs().max().item() > 0          # the mode really changes the stored values


def test_bf16_storage_mode_whole_model(dev):
    """bf16 storage mode on the whole CIDNet (reduced width): forward within 2e-2 of the fp32 mode, gradients within 5e-2
    of each tensor's max -- the looser tier of this mode; fp32 stays the parity mode"""
    import hvi_cidnet_amd as P
    chans = (12, 12, 24, 48)
    p = O.make_params(5, channels=chans)
    x = O.synthetic_batch(51, (2, 3, 32, 48)).to(dev)
    gt = O.synthetic_batch(52, (2, 3, 32, 48)).to(dev)
    res = {}
    try:
        for mode in ("f32", "bf16"):
            P.set_storage_dtype(mode)
            m = P.CIDNet(channels=list(chans))
            m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
            m.to(dev)
            y = m(x)
            (y - gt).abs().mean().backward()
            res[mode] = (y.detach(), {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None})
    finally:
        P.set_storage_dtype("f32")
    d = (res["bf16"][0] - res["f32"][0]).abs().max().item()
    assert 0 < d <= 2e-2, d
    # gradients: relative L2 error per tensor (max-norm comparisons of tiny, nearly cancelling gradients are noisy in any mode)
    worst, wname = 0.0, ""
    for n, g in res["f32"][1].items():
        if g.numel() < 64:
            continue
        e = ((res["bf16"][1][n] - g).double().norm() / g.double().norm().clamp_min(1e-30)).item()
        if e > worst:
            worst, wname = e, n
    print(f"bf16 storage mode: output max |diff| {d:.3e}; worst gradient relative L2 error {worst:.3e} ({wname})")
    assert worst <= 0.15, (worst, wname)          # measured 0.079 (a depthwise weight gradient summed over bf16-rounded du)


@pytest.mark.parametrize("B,M,K,H,W,flip,add", [(2, 36, 36, 37, 51, 0, 0), (2, 72, 36, 20, 33, 0, 1), (1, 36, 36, 64, 96, 1, 1),
                                                (2, 100, 36, 12, 40, 0, 0), (1, 36, 72, 21, 150, 1, 0), (1, 144, 72, 13, 75, 0, 1),
                                                (1, 72, 144, 10, 75, 1, 0), (3, 36, 36, 8, 32, 0, 0), (1, 5, 36, 1, 1, 0, 0)])
def test_conv3x3_bf16x3_split_products(dev, B, M, K, H, W, flip, add):
    """csrc/conv3x.hip: the zero-pad 3x3 conv on the BF16 matrix cores with every fp32 operand split exactly into three bf16
    values and the six significant cross products accumulated in fp32.  Against the fp64 convolution its error must not
    exceed the fp32-MFMA kernel's (x1.5 + 1e-6 of the output scale): forward and data-gradient (flipped, transposed
    weights) forms, addend epilogue, ragged sizes (rows of 75 / 150 pixels end inside a pixel quad), several 48-channel
    output chunks and 36-channel input chunks; every output element written (NaN prefill) and reruns bit-identical."""
    from hvi_cidnet_amd import ops
    import torch.nn.functional as F
    g = torch.Generator(device=dev).manual_seed(B * 1000 + M + H)
    x = torch.randn(B, K, H, W, device=dev, generator=g)
    r = torch.randn(B, M, H, W, device=dev, generator=g) if add else None
    if flip:
        wt = torch.randn(K, M, 3, 3, device=dev, generator=g) / (3 * K ** 0.5)      # forward weight (Cout = K, Cin = M)
        w_ms, w_ks = 9, 9 * M
        ref = F.conv_transpose2d(x.double().cpu(), wt.double().cpu(), padding=1)
    else:
        wt = torch.randn(M, K, 3, 3, device=dev, generator=g) / (3 * K ** 0.5)
        w_ms, w_ks = 9 * K, 9
        ref = F.conv2d(x.double().cpu(), wt.double().cpu(), padding=1)
    if add:
        ref = ref + r.double().cpu()
    assert ops._raw("cidnet_conv3x3_bf16x3_supported", M, K) == 1
    y32 = torch.empty(B, M, H, W, device=dev)
    ys, ys2 = torch.full((B, M, H, W), float("nan"), device=dev), torch.full((B, M, H, W), float("nan"), device=dev)
    old = dict(ops.CONV3_BF16X3)
    try:
        ops.CONV3_BF16X3["on"] = False
        ops.conv3x3(x, wt, y32, B, M, K, H, W, w_ms, w_ks, flip=bool(flip), addend=r)
        ops.CONV3_BF16X3["on"] = True
        ops.conv3x3(x, wt, ys, B, M, K, H, W, w_ms, w_ks, flip=bool(flip), addend=r)
        ops.conv3x3(x, wt, ys2, B, M, K, H, W, w_ms, w_ks, flip=bool(flip), addend=r)
    finally:
        ops.CONV3_BF16X3.update(old)
    assert not torch.isnan(ys).any()
    assert torch.equal(ys, ys2)
    e32 = (y32.cpu().double() - ref).abs().max().item()
    es = (ys.cpu().double() - ref).abs().max().item()
    assert es <= 1.5 * e32 + 1e-6 * ref.abs().max().item(), (es, e32)


@pytest.mark.parametrize("B,M,K,HW,res,form", [
    (2, 144, 766, 3750, False, "fwd"), (2, 766, 144, 3750, True, "fwd"), (3, 72, 382, 1501, True, "fwd"),
    (2, 36, 36, 999, False, "fwd"), (1, 20, 32, 64, False, "fwd"), (2, 191, 72, 777, False, "dgrad"),
    (2, 144, 144, 1000, True, "per_sample"), (1, 17, 7, 67, True, "fwd"), (2, 95, 36, 64, True, "fwd"), (1, 288, 288, 3750, False, "dgrad"),
    (2, 72, 72, 15000, True, "fwd"), (2, 60, 95, 130, True, "fwd"), (1, 383, 144, 3750, True, "fwd"), (2, 36, 95, 1502, True, "dgrad")])
def test_pw_conv_bf16x3_split_products(dev, B, M, K, HW, res, form):
    """csrc/pwx.hip: the 1x1 conv on the BF16 matrix cores with exact three-way split operands against the fp64 product;
    bar = 3x the fp32-MFMA kernel's own error + 2e-6 of the output scale (the matrix core's fp32 accumulation of six
    products per term is a little looser than the fp32 MFMA at K = 766: 5e-6 against 1.3e-6).  Shapes: every block layout
    (1, 2, 4 waves of a block on one pixel group; 2-5 channel tiles per wave; several channel chunks), ragged pixel /
    channel / K tails (planes whose size is not a multiple of 4 or of 64 -- the last group is pulled back and stores only
    the pixels it owns --, a 64-pixel plane), the data-gradient weight strides and per-sample weights (attention fold);
    every output element written (NaN prefill) and a rerun bit-identical."""
    from hvi_cidnet_amd import ops
    g = torch.Generator(device=dev).manual_seed(M * 7 + K)
    x = torch.randn(B, K, HW, device=dev, generator=g)
    r = torch.randn(B, M, HW, device=dev, generator=g) if res else None
    if form == "dgrad":                                     # weight stored (K, M): A[m][k] = w[k][m]
        w = torch.randn(K, M, device=dev, generator=g) / K ** 0.5
        w_bs, w_ms, w_ks = 0, 1, M
        a = w.t().double().cpu().expand(B, M, K)
    elif form == "per_sample":
        w = torch.randn(B, M, K, device=dev, generator=g) / K ** 0.5
        w_bs, w_ms, w_ks = M * K, K, 1
        a = w.double().cpu()
    else:
        w = torch.randn(M, K, device=dev, generator=g) / K ** 0.5
        w_bs, w_ms, w_ks = 0, K, 1
        a = w.double().cpu().expand(B, M, K)
    ref = torch.bmm(a, x.double().cpu())
    if res:
        ref = ref + r.double().cpu()
    y32, ys = torch.empty(B, M, HW, device=dev), torch.full((B, M, HW), float("nan"), device=dev)
    old = dict(ops.PW_BF16X3)
    try:
        ops.PW_BF16X3["on"] = False
        ops.pw_conv(x, 0, K * HW, w, 0, w_bs, w_ms, w_ks, y32, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW)
    finally:
        ops.PW_BF16X3.update(old)
    ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, w_bs, w_ms, w_ks, ys, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW)
    ys2 = torch.full((B, M, HW), float("nan"), device=dev)
    ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, w_bs, w_ms, w_ks, ys2, 0, M * HW, B, M, K, HW, res=r, r_off=0, r_bs=M * HW)
    assert not torch.isnan(ys).any() and torch.equal(ys, ys2)
    e32 = (y32.cpu().double() - ref).abs().max().item()
    es = (ys.cpu().double() - ref).abs().max().item()
    assert es <= 3 * e32 + 2e-6 * ref.abs().max().item(), (es, e32)


def test_prepared_weight_operands_equal_per_call_preparation(dev):
    """ops.enable_prepared_weights: the split / fragment-ordered weight operands of the bf16x3 1x1 and 3x3 convs kept across
    calls.  Results are bit-identical to the per-call preparation (cidnet_*_bf16x3 == _prep + _pre); the batched refresh
    (one launch for all cached operands of a family) equals the individual preparation; a weight changed through torch
    (version counter) is re-prepared on use, one changed behind torch's back is picked up by refresh_prepared_weights."""
    from hvi_cidnet_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    B, HW, H, W = 2, 40 * 60, 40, 60
    pw_shapes = [(36, 95, "fwd"), (72, 36, "dgrad"), (190, 72, "fwd"), (40, 144, "dgrad")]
    c3_shapes = [(36, 36, 0), (72, 36, 1), (36, 72, 0)]
    pws = []
    for M, K, form in pw_shapes:
        w = torch.randn((M, K) if form == "fwd" else (K, M), device=dev, generator=g) / K ** 0.5
        pws.append((M, K, w, (K, 1) if form == "fwd" else (1, M), torch.randn(B, K, HW, device=dev, generator=g)))
    c3s = []
    for M, K, flip in c3_shapes:
        w = torch.randn((K, M, 3, 3) if flip else (M, K, 3, 3), device=dev, generator=g) / (3 * K ** 0.5)
        c3s.append((M, K, flip, w, (9, 9 * M) if flip else (9 * K, 9), torch.randn(B, K, H, W, device=dev, generator=g)))

    def run_all():
        outs = []
        for M, K, w, (ms, ks), x in pws:
            y = torch.full((B, M, HW), float("nan"), device=dev)
            ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, 0, ms, ks, y, 0, M * HW, B, M, K, HW)
            outs.append(y)
        for M, K, flip, w, (ms, ks), x in c3s:
            y = torch.full((B, M, H, W), float("nan"), device=dev)
            ops.conv3x3(x, w, y, B, M, K, H, W, ms, ks, flip=bool(flip))
            outs.append(y)
        return outs

    def same(a, b):
        return all(torch.equal(p, q) for p, q in zip(a, b))
    old = dict(ops.CONV3_BF16X3)
    ops.CONV3_BF16X3["on"] = True
    try:
        ops.enable_prepared_weights(False)
        ref = run_all()
        ops.enable_prepared_weights(True)
        first = run_all()                       # every operand prepared on first use
        assert ops._PREP["stats"][1] >= len(pws) + len(c3s)
        hits0 = ops._PREP["stats"][0]
        again = run_all()                       # all hits
        assert ops._PREP["stats"][0] - hits0 == len(pws) + len(c3s)
        assert same(ref, first) and same(ref, again)
        # in-place update through torch: seen by the version counter
        pws[0][2].mul_(1.5)
        c3s[1][3].add_(0.01)
        ops.enable_prepared_weights(False)
        ref2 = run_all()
        ops.enable_prepared_weights(True)
        run_all()
        keep = pws[0][2].clone(), c3s[1][3].clone()
        pws[0][2].mul_(0.3); c3s[1][3].sub_(0.02)                # changed while cached ...
        assert not same(ref2, run_all())
        pws[0][2].copy_(keep[0]); c3s[1][3].copy_(keep[1])       # ... and back (in-place: the version counter moves)
        assert same(ref2, run_all())
        assert not same(ref, ref2)
        # a write behind torch's back (raw kernel on the buffer, as the fused Adam does): stale until the refresh
        for t in (pws[1][2], pws[2][2], c3s[0][3], c3s[2][3]):
            ops.lib().call("cidnet_scale", ops._p(t), None, ops._f(0.5), ops._p(t), t.numel(), ops._stream())
        stale = run_all()
        assert same(stale, ref2)
        ops.refresh_prepared_weights(dev if isinstance(dev, torch.device) else torch.device(dev))
        fresh = run_all()
        ops.enable_prepared_weights(False)
        ref3 = run_all()
        assert same(fresh, ref3) and not same(ref3, ref2)
    finally:
        ops.enable_prepared_weights(False)
        ops.CONV3_BF16X3.update(old)


def test_bilinear_tables_once_per_shape(dev):
    """cidnet_bilinear_bwd == _tabs + _pre; ops.bilinear_bwd keeps the tap tables per shape"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    for (B, C, Hi, Wi, Ho, Wo) in [(2, 5, 16, 24, 8, 12), (1, 7, 10, 15, 20, 30), (2, 3, 50, 75, 100, 150)]:
        g = rnd(7, (B, C, Ho, Wo)).to(dev)
        a, b2 = torch.full((B, C, Hi, Wi), float("nan"), device=dev), torch.full((B, C, Hi, Wi), float("nan"), device=dev)
        n = ops._raw("cidnet_bilinear_bwd_ws_floats", Hi, Wi)
        ws = torch.empty(n, device=dev)
        lib().call("cidnet_bilinear_bwd", ops._p(g), ops._p(a), ops._p(ws), ws.numel(), B, C, Hi, Wi, Ho, Wo, ops._stream())
        ops.bilinear_bwd(g, b2, B, C, Hi, Wi, Ho, Wo)
        ops.bilinear_bwd(g, b2, B, C, Hi, Wi, Ho, Wo)
        assert torch.equal(a, b2)


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("B,M,K,HW,res", [(2, 36, 95, 60 * 40, True), (1, 190, 36, 75 * 50, False), (2, 72, 191, 1000, True),
                                          (1, 144, 766, 75 * 50 + 2, False), (2, 288, 144, 129, False)])
def test_pw_conv_one_level_is_the_bf16_product(dev, B, M, K, HW, res):
    """ops.set_math_levels(1) (the bf16 mode): the 1x1 conv multiplies operands rounded to nearest bf16, ONE product per term,
    fp32 accumulation.  Reference = the fp64 product of the rounded operands; what is left is fp32 summation error."""
    from hvi_cidnet_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + K)
    x = torch.randn(B, K, HW, device=dev, generator=g)
    w = torch.randn(M, K, device=dev, generator=g) / K ** 0.5
    r = torch.randn(B, M, HW, device=dev, generator=g) if res else None
    ref = torch.matmul(_bf16_round(w).cpu(), _bf16_round(x).cpu())
    if res:
        ref = ref + r.double().cpu()
    y = torch.full((B, M, HW), float("nan"), device=dev)
    ops.set_math_levels(1)
    try:
        ops.pw_conv_bf16x3(x, 0, K * HW, w, 0, 0, K, 1, y, 0, M * HW, B, M, K, HW, res=r, r_bs=M * HW)
    finally:
        ops.set_math_levels(3)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item() + 1e-6, err
    # and it is NOT the fp32 product: bf16 rounding of the operands is visible (guards against a silently ignored mode)
    exact = torch.matmul(w.double().cpu(), x.double().cpu()) + (r.double().cpu() if res else 0)
    assert (y.double().cpu() - exact).abs().max().item() > 1e-4


@pytest.mark.parametrize("B,M,K,H,W,flip,add", [(1, 36, 36, 24, 40, 0, False), (2, 72, 36, 16, 75, 1, True), (1, 36, 72, 20, 34, 0, True)])
@pytest.mark.parametrize("wl,xl", [(1, 1), (3, 1)])
def test_conv3x3_level_modes(dev, B, M, K, H, W, flip, add, wl, xl):
    """cidnet_conv3x3_bf16x3_pre_lv: (1, 1) both operands rounded to bf16, one product; (3, 1) exact weights, rounded
    activations -- against the fp64 convolution of the correspondingly rounded operands"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    g = torch.Generator(device=dev).manual_seed(M + K + H)
    x = torch.randn(B, K, H, W, device=dev, generator=g)
    r = torch.randn(B, M, H, W, device=dev, generator=g) if add else None
    if flip:
        wt = torch.randn(K, M, 3, 3, device=dev, generator=g) / (3 * K ** 0.5)
        w_ms, w_ks = 9, 9 * M
    else:
        wt = torch.randn(M, K, 3, 3, device=dev, generator=g) / (3 * K ** 0.5)
        w_ms, w_ks = 9 * K, 9
    wr = (_bf16_round(wt) if wl == 1 else wt.double()).cpu()
    xr = _bf16_round(x).cpu()
    ref = F.conv_transpose2d(xr, wr, padding=1) if flip else F.conv2d(xr, wr, padding=1)
    if add:
        ref = ref + r.double().cpu()
    n = ops._raw("cidnet_conv3x3_bf16x3_ws_floats", M, K)
    ws = torch.empty(n, device=dev)
    y = torch.full((B, M, H, W), float("nan"), device=dev)
    lib().call("cidnet_conv3x3_bf16x3_prep", ops._p(wt), w_ms, w_ks, flip, ops._p(ws), ws.numel(), M, K, ops._stream())
    lib().call("cidnet_conv3x3_bf16x3_pre_lv", ops._p(x), K * H * W, ops._p(ws), ops._p(r), M * H * W, ops._p(y), M * H * W, B, M, K, H, W,
               wl, xl, ops._stream())
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("B,M,N,H,W", [(2, 36, 36, 16, 40), (1, 72, 36, 12, 75), (1, 36, 72, 9, 34)])
def test_conv3x3_wgrad_one_level(dev, B, M, N, H, W):
    from hvi_cidnet_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N + W)
    x = torch.randn(B, N, H, W, device=dev, generator=g)
    dy = torch.randn(B, M, H, W, device=dev, generator=g)
    xr = _bf16_round(x).cpu().requires_grad_(False)
    wz = torch.zeros(M, N, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wz, padding=1).backward(_bf16_round(dy).cpu())
    dw = torch.full((M, N, 3, 3), float("nan"), device=dev)
    ops.set_math_levels(1)
    try:
        ops.conv3x3_wgrad(dy, x, dw, B, M, N, H, W)
    finally:
        ops.set_math_levels(3)
    err = (dw.double().cpu() - wz.grad).abs().max().item()
    assert err <= 1e-5 * wz.grad.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("B,M,N,HW,dts", [(2, 36, 95, 2400, (0, 0)), (1, 190, 36, 3750, (1, 0)), (2, 72, 72, 1000, (0, 1)), (1, 95, 36, 600, (1, 1))])
def test_pw_wgrad_one_level(dev, B, M, N, HW, dts):
    """CIDNET_WGRAD_BF16_1LEVEL with fp32 and bf16-stored operands"""
    from hvi_cidnet_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + N)
    dy = torch.randn(B, M, HW, device=dev, generator=g)
    x = torch.randn(B, N, HW, device=dev, generator=g)
    ref = torch.einsum("bmp,bnp->mn", _bf16_round(dy).cpu(), _bf16_round(x).cpu())
    dyt = dy.to(torch.bfloat16) if dts[0] else dy
    xt = x.to(torch.bfloat16) if dts[1] else x
    dw = torch.full((M, N), float("nan"), device=dev)
    ops.set_math_levels(1)
    try:
        ops.pw_wgrad(dyt, 0, M * HW, xt, 0, N * HW, dw, 0, N, B, M, N, HW)
    finally:
        ops.set_math_levels(3)
    err = (dw.double().cpu() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("B,M,K,HW,res", [(2, 36, 95, 60 * 40, True), (1, 190, 36, 75 * 50, False), (2, 72, 191, 1001, True), (1, 144, 383, 130, False)])
@pytest.mark.parametrize("xb,yb", [(True, False), (False, True), (True, True)])
def test_pw_conv_bf16_typed_tensors(dev, B, M, K, HW, res, xb, yb):
    """cidnet_pw_conv_bf16x3_pre_t: activations and / or output STORED as bf16 (the bf16 mode's IEL / CAB tensors).  A bf16
    input is read without conversion (v_perm packs the channel pairs); a bf16 output is the fp32 result rounded to nearest.
    Reference: fp64 product of the bf16-rounded operands, then the same rounding of the output."""
    from hvi_cidnet_amd import ops
    g = torch.Generator(device=dev).manual_seed(M + K + HW)
    x = torch.randn(B, K, HW, device=dev, generator=g)
    w = torch.randn(M, K, device=dev, generator=g) / K ** 0.5
    r = torch.randn(B, M, HW, device=dev, generator=g) if res else None
    ref = torch.matmul(_bf16_round(w).cpu(), _bf16_round(x).cpu())
    if res:
        ref = ref + r.double().cpu()
    xt = x.to(torch.bfloat16) if xb else x
    y = torch.full((B, M, HW), float("nan"), device=dev, dtype=torch.bfloat16 if yb else torch.float32)
    ops.set_math_levels(1)
    try:
        ops.pw_conv(xt, 0, K * HW, w, 0, 0, K, 1, y, 0, M * HW, B, M, K, HW, res=r, r_bs=M * HW)
    finally:
        ops.set_math_levels(3)
    assert not torch.isnan(y.float()).any()
    err = (y.double().cpu() - ref).abs().max().item()
    bar = (2.0 ** -8 if yb else 3e-6) * ref.abs().max().item() + 1e-6          # bf16 store: half an ulp of the largest value
    assert err <= bar, (err, bar)


@pytest.mark.parametrize("B,C,H,W", [(2, 36, 8, 12), (1, 72, 10, 15), (2, 144, 5, 7), (1, 36, 25, 40)])
def test_layernorm_bf16_typed_output_and_gradient(dev, B, C, H, W):
    """cidnet_ln_cf_fwd_t / _bwd_res_t: in the bf16 mode the LayerNorm of an LCA block writes its OUTPUT as bf16 (the fp32
    result rounded to nearest) and reads the incoming gradient as bf16; x, statistics, gx and the parameter gradients stay
    fp32.  Against the fp32 kernels on the same (bf16-representable) gradient: bit-equal arithmetic, so equal results."""
    from hvi_cidnet_amd import ops
    if not ops._raw("cidnet_ln_cf_typed_supported", B, C, H * W):
        pytest.skip("shape without typed LayerNorm kernels")
    x = rnd(61, (B, C, H, W), 2.0).to(dev)
    w, b = (1.0 + 0.3 * rnd(62, (C,))).to(dev), (0.2 * rnd(63, (C,))).to(dev)
    gy = rnd(64, (B, C, H, W)).to(dev).to(torch.bfloat16)            # a bf16-representable gradient
    res = rnd(65, (B, C, H, W)).to(dev)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y, xp = ops.LayerNormResFn.apply(xr, wr, br, 1e-6, None, dt)
        assert y.dtype == dt
        torch.autograd.backward([y, xp], [gy.to(dt), res])
        outs[dt] = (y, xr.grad, wr.grad, br.grad)
    y32, yh = outs[torch.float32][0], outs[torch.bfloat16][0]
    assert torch.equal(yh, y32.to(torch.bfloat16))                    # the same value, rounded on store
    for a, b_ in zip(outs[torch.float32][1:], outs[torch.bfloat16][1:]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("B,C,H,W", [(2, 12, 9, 13), (1, 108, 16, 24), (1, 7, 3, 3), (2, 36, 40, 75)])
def test_dw3x3_bf16_typed(dev, B, C, H, W):
    """cidnet_dw3x3_t with bf16-stored input / output: the fp32 stencil on the bf16 values, rounded on store"""
    from hvi_cidnet_amd import ops
    x = rnd(71, (B, C, H, W)).to(dev).to(torch.bfloat16)
    w = rnd(72, (C, 1, 3, 3), 0.5).to(dev)
    y32 = torch.empty((B, C, H, W), device=dev)
    yh = torch.full((B, C, H, W), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.dw3x3(x.float(), w, None, C, y32, B, C, H, W)
    ops.dw3x3(x, w, None, C, yh, B, C, H, W)
    assert torch.equal(yh, y32.to(torch.bfloat16))


@pytest.mark.parametrize("B,C,heads,HW", [(2, 36, 2, 60 * 40), (1, 72, 4, 1001), (2, 144, 8, 130)])
def test_attn_fwd_bf16_typed_qkv(dev, B, C, heads, HW):
    """cidnet_attn_fwd_t: q | k | v stored as bf16 are read as the same values the fp32 entry point reads from their fp32 copies"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd._lib import lib
    ch = C // heads
    qkv = rnd(81, (B, 3 * C, HW)).to(dev).to(torch.bfloat16)
    T = (1.0 + 0.1 * rnd(82, (heads, 1, 1))).to(dev)
    wp = rnd(83, (C, C), 0.3).to(dev)
    res = []
    for t in (qkv.float().contiguous(), qkv):
        attn = torch.empty((B, heads, ch, ch), device=dev); shat = torch.empty_like(attn)
        nq = torch.empty((B, C), device=dev); nk = torch.empty_like(nq); M = torch.empty((B, C, C), device=dev)
        n = ops._raw("cidnet_attn_gram_ws_floats", B, C, heads, HW)
        ws = torch.empty(n, device=dev)
        lib().call("cidnet_attn_fwd_t", ops._p(t), ops._dt(t), ops._p(T), ops._p(wp), ops._p(attn), ops._p(shat), ops._p(nq), ops._p(nk),
                   ops._p(M), ops._p(ws), ws.numel(), B, C, heads, HW, 1, ops._stream())
        res.append((attn, shat, nq, nk, M))
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)
