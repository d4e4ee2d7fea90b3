"""Inference wrapper and checkpoint formats ("next" row f3): padding arithmetic and file round trips on the CPU, the
full wrapper against the oracle's composition on the GPU."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import cidnet_oracle as O


def test_pad_to_multiple_reference_arithmetic():
    from hvi_cidnet_amd.inference import pad_to_multiple
    for h, w in [(37, 51), (40, 48), (8, 9), (16, 17)]:
        x = torch.arange(3 * h * w, dtype=torch.float32).reshape(1, 3, h, w)
        y, (h0, w0) = pad_to_multiple(x, 8)
        assert (h0, w0) == (h, w)
        H = ((h + 8) // 8) * 8 if h % 8 else h
        W = ((w + 8) // 8) * 8 if w % 8 else w
        assert tuple(y.shape[-2:]) == (H, W)
        ref = F.pad(x, (0, W - w, 0, H - h), "reflect") if (H != h or W != w) else x
        assert torch.equal(y, ref)
    with pytest.raises(RuntimeError):      # reflect padding wider than the image: torch raises, as it does for the reference
        pad_to_multiple(torch.zeros(1, 3, 1, 17), 8)


def test_checkpoint_round_trips(tmp_path):
    """state_dict keys are the reference's (191 tensors); .pth is strict, the safetensors + config.json layout loads too"""
    import hvi_cidnet_amd as P
    m = P.CIDNet(channels=[12, 12, 24, 48])
    p = O.make_params(3, channels=(12, 12, 24, 48))
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
    pth = os.path.join(tmp_path, "epoch_1.pth")
    torch.save(m.state_dict(), pth)
    m2 = P.CIDNet(channels=[12, 12, 24, 48])
    assert P.load_weights(m2, pth) == ([], [])
    d = P.save_pretrained(m, os.path.join(tmp_path, "hf"), config={"channels": [12, 12, 24, 48], "heads": [1, 2, 4, 8], "norm": False})
    m3 = P.CIDNet(channels=[12, 12, 24, 48])
    assert P.load_weights(m3, d) == ([], [])
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]) and torch.equal(v, m3.state_dict()[k]), k
    assert len(m.state_dict()) == 191


@pytest.mark.gpu
@pytest.mark.parametrize("gamma,alpha_s,alpha_i,gated", [(1.0, 1.0, 1.0, False), (0.8, 1.3, 0.9, True)])
def test_enhance_matches_oracle_composition(dev, gamma, alpha_s, alpha_i, gated):
    import hvi_cidnet_amd as P
    chans = (12, 12, 24, 48)
    m = P.CIDNet(channels=list(chans))
    p = O.make_params(4, channels=chans)
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
    m.to(dev)
    m.trans.gated = gated
    m.trans.gated2 = gated
    img = O.synthetic_batch(141, (1, 3, 37, 51))[0]
    out = P.enhance(m, img.to(dev), gamma=gamma, alpha_s=alpha_s, alpha_i=alpha_i)
    assert tuple(out.shape) == (3, 37, 51)
    x = F.pad(img.unsqueeze(0), (0, 56 - 51, 0, 40 - 37), "reflect") ** gamma
    ref = O.cidnet_forward(p, x, gated=gated, alpha_s=alpha_s, gated2=gated, alpha=alpha_i)
    ref = torch.clamp(ref, 0, 1)[0, :, :37, :51]
    assert (out.cpu() - ref).abs().max().item() <= 1e-4


@pytest.mark.gpu
def test_no_grad_inference_stores_no_backward_tensors(dev):
    """Under torch.no_grad() with default (requires_grad=True) parameters, the tensors only the backward reads -- the IEL's
    u and the PReLU pre-activations of the down / up blocks -- are not allocated: the C ABI receives null pointers
    (ctx.needs_input_grad stays True for parameters under no_grad, so the modules decide)."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import _lib
    m = P.CIDNet(channels=[12, 12, 24, 48]).to(dev)
    assert all(p.requires_grad for p in m.parameters())
    L = _lib.lib()
    seen = {}
    orig = L.call

    def spy(name, *args):
        if name in ("cidnet_iel_dw_gate_fwd", "cidnet_down_prelu_fwd", "cidnet_pw_conv_up_prelu", "cidnet_iel_fwd"):
            seen.setdefault(name, []).append(args)
        return orig(name, *args)
    L.call = spy
    try:
        x = torch.rand(1, 3, 32, 48, device=dev)
        with torch.no_grad():
            m(x)
        infer = {k: list(v) for k, v in seen.items()}
        seen.clear()
        m(x)
        train = {k: list(v) for k, v in seen.items()}
    finally:
        L.call = orig

    def nulls(calls, idx):
        return [a[idx] is None or getattr(a[idx], "value", 1) is None for a in calls]
    assert infer.get("cidnet_down_prelu_fwd") and all(nulls(infer["cidnet_down_prelu_fwd"], 2))
    assert not any(nulls(train["cidnet_down_prelu_fwd"], 2))
    assert infer.get("cidnet_pw_conv_up_prelu") and all(nulls(infer["cidnet_pw_conv_up_prelu"], 8))
    assert not any(nulls(train["cidnet_pw_conv_up_prelu"], 8))
    if "cidnet_iel_dw_gate_fwd" in infer:
        assert all(nulls(infer["cidnet_iel_dw_gate_fwd"], 4)) and not any(nulls(train["cidnet_iel_dw_gate_fwd"], 4))
