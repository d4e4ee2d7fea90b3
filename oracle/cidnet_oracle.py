"""CPU oracle for the CIDNet forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain PyTorch CPU ops + autograd) of the
algorithm in the reference's `net/HVI_transform.py`, `net/transformer_utils.py`, `net/LCA.py`
and `net/CIDNet.py`.  It exists so that the hand-written HIP kernels in `hvi-cidnet_amd/csrc`
can be checked against something that (a) travels to the GPU box (the reference cannot) and
(b) has been pinned to the reference itself: `oracle/gen_golden.py` imports the reference in
the development container, asserts this restatement equals it (bit-exact forward for
HVIT/PHVIT and for the whole network on CPU; gradients to rounding) and writes the fixtures in
`tests/golden/`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`hvi-cidnet_amd/`) never does; it fails loudly without its HIP
library.

Parameters are passed as a flat dict keyed by the reference's `state_dict` names, e.g.
`"HV_LCA1.ffn.q.weight"`.  All functions are differentiable through torch autograd, and work
in float32 or float64 (`params_to(params, torch.float64)`).
"""
from __future__ import annotations

import math
import re
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

PI = 3.141592653589793  # reference: net/HVI_transform.py:4
EPS = 1e-8              # reference: net/HVI_transform.py:17,50

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# HVI colour transform
# --------------------------------------------------------------------------------------
def hvit(img: torch.Tensor, density_k: torch.Tensor) -> torch.Tensor:
    """RGB -> HVI.  Restates `RGB_HVI.HVIT`, net/HVI_transform.py:16-47.

    The reference fills an uninitialised hue buffer with three boolean-mask assignments in the
    order B, G, R (:23-25), so a later mask wins: tie priority is R > G > B; then the
    max==min (gray) mask zeroes hue (:27).  `torch.where` nests the same selection and routes
    gradient only through the winning branch, as the masked index_put does.
    """
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    value = img.max(1)[0]                       # :21 (first arg-max gets the gradient)
    img_min = img.min(1)[0]                     # :22
    denom = value - img_min + EPS
    hue_b = 4.0 + (r - g) / denom               # :23
    hue_g = 2.0 + (b - r) / denom               # :24
    hue_r = torch.remainder(0.0 + (g - b) / denom, 6)   # :25  (python-style modulo)
    hue = torch.where(r == value, hue_r, torch.where(g == value, hue_g, hue_b))
    hue = torch.where(img.min(1)[0] == value, torch.zeros_like(hue), hue)   # :27
    hue = hue / 6.0                             # :28

    sat = (value - img_min) / (value + EPS)     # :30
    sat = torch.where(value == 0, torch.zeros_like(sat), sat)               # :31

    hue, sat, value = hue.unsqueeze(1), sat.unsqueeze(1), value.unsqueeze(1)
    cs = ((value * 0.5 * PI).sin() + EPS).pow(density_k)                    # :40
    ch = (2.0 * PI * hue).cos()                 # :41
    cv = (2.0 * PI * hue).sin()                 # :42
    H = cs * sat * ch                           # :43
    V = cs * sat * cv                           # :44
    return torch.cat([H, V, value], dim=1)      # :45-46


def hvit_branch_code(img: torch.Tensor) -> torch.Tensor:
    """Integer restatement of HVIT's mask logic (net/HVI_transform.py:23-27,31) used for the
    bit-exact index test: bits 0-1 = hue branch (0 gray, 1 R, 2 G, 3 B), bits 2-3 = arg-max
    channel that receives d(value), bits 4-5 = arg-min channel, bit 6 = (value == 0)."""
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    value, amax = img.max(1)
    mn, amin = img.min(1)
    br = torch.where(r == value, 1, torch.where(g == value, 2, 3))
    br = torch.where(mn == value, 0, br)
    code = br + (amax << 2) + (amin << 4) + ((value == 0).to(torch.int64) << 6)
    return code.to(torch.uint8)


def phvit(hvi: torch.Tensor, k: float, gated: bool = False, alpha_s: float = 1.3,
          gated2: bool = False, alpha: float = 1.0) -> torch.Tensor:
    """HVI -> RGB.  Restates `RGB_HVI.PHVIT`, net/HVI_transform.py:49-122.  `k` is the python
    float `this_k` (:59): no gradient reaches `density_k` from here.  Pixels whose `h % 1`
    rounds to 1.0 get hi == 6, match none of the six masks (:85-90) and stay black."""
    H, V, I = hvi[:, 0], hvi[:, 1], hvi[:, 2]
    H = torch.clamp(H, -1, 1)                   # :54
    V = torch.clamp(V, -1, 1)                   # :55
    I = torch.clamp(I, 0, 1)                    # :56
    v = I
    cs = ((v * 0.5 * PI).sin() + EPS).pow(k)    # :60
    H = H / (cs + EPS)                          # :61
    V = V / (cs + EPS)                          # :62
    H = torch.clamp(H, -1, 1)                   # :63
    V = torch.clamp(V, -1, 1)                   # :64
    h = torch.atan2(V + EPS, H + EPS) / (2 * PI)  # :65
    h = h % 1                                   # :66
    s = torch.sqrt(H ** 2 + V ** 2 + EPS)       # :67
    if gated:
        s = s * alpha_s                         # :69-70
    s = torch.clamp(s, 0, 1)                    # :72
    v = torch.clamp(v, 0, 1)                    # :73

    hi = torch.floor(h * 6.0)                   # :79
    f = h * 6.0 - hi                            # :80
    p = v * (1. - s)                            # :81
    q = v * (1. - (f * s))                      # :82
    t = v * (1. - ((1. - f) * s))               # :83
    z = torch.zeros_like(h)                     # :75-77

    def pick(c0, c1, c2, c3, c4, c5):
        out = z
        for n, c in enumerate((c0, c1, c2, c3, c4, c5)):
            out = torch.where(hi == n, c, out)
        return out

    r = pick(v, q, p, p, t, v)                  # :92,96,100,104,108,112
    g = pick(t, v, v, q, p, p)                  # :93,97,101,105,109,113
    b = pick(p, p, t, v, v, q)                  # :94,98,102,106,110,114
    rgb = torch.stack([r, g, b], dim=1)         # :116-119
    if gated2:
        rgb = rgb * alpha                       # :120-121
    return rgb


def phvit_sextant(hvi: torch.Tensor, k: float) -> torch.Tensor:
    """floor(6h) of PHVIT (net/HVI_transform.py:65-66,79) as uint8 (6 == the black case)."""
    H, V, I = hvi[:, 0].clamp(-1, 1), hvi[:, 1].clamp(-1, 1), hvi[:, 2].clamp(0, 1)
    cs = ((I * 0.5 * PI).sin() + EPS).pow(k)
    H = (H / (cs + EPS)).clamp(-1, 1)
    V = (V / (cs + EPS)).clamp(-1, 1)
    h = (torch.atan2(V + EPS, H + EPS) / (2 * PI)) % 1
    return torch.floor(h * 6.0).to(torch.uint8)


# --------------------------------------------------------------------------------------
# transformer_utils blocks
# --------------------------------------------------------------------------------------
def layernorm_cf(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-6):
    """channels_first LayerNorm, net/transformer_utils.py:24-29 (biased variance over C)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    xn = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * xn + bias[:, None, None]


def bilinear_ac(x: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    """`nn.UpsamplingBilinear2d` == bilinear interpolate with align_corners=True
    (net/transformer_utils.py:40,59); output size is floor(in * scale)."""
    return F.interpolate(x, size=out_hw, mode="bilinear", align_corners=True)


def norm_downsample(x, p: Params, pre: str):
    """`NormDownsample.forward`, net/transformer_utils.py:38-48; the `use_norm` LayerNorm (:44-46, after the PReLU) is
    applied when the parameter set holds `<pre>.norm.*` (CIDNet(norm=True), net/CIDNet.py:12)."""
    y = F.conv2d(x, p[pre + ".down.0.weight"], padding=1)
    y = bilinear_ac(y, (int(math.floor(y.shape[2] * 0.5)), int(math.floor(y.shape[3] * 0.5))))
    y = F.prelu(y, p[pre + ".prelu.weight"])
    if pre + ".norm.weight" in p:
        y = layernorm_cf(y, p[pre + ".norm.weight"], p[pre + ".norm.bias"])
    return y


def norm_upsample(x, skip, p: Params, pre: str):
    """`NormUpsample.forward`, net/transformer_utils.py:62-70; with `<pre>.norm.*` present the `use_norm` LayerNorm (:67-68)."""
    y = F.conv2d(x, p[pre + ".up_scale.0.weight"], padding=1)
    y = bilinear_ac(y, (y.shape[2] * 2, y.shape[3] * 2))
    y = torch.cat([y, skip], dim=1)
    y = F.conv2d(y, p[pre + ".up.weight"])
    y = F.prelu(y, p[pre + ".prelu.weight"])
    if pre + ".norm.weight" in p:
        y = layernorm_cf(y, p[pre + ".norm.weight"], p[pre + ".norm.bias"])
    return y


def rep_conv3x3(x, w):
    """`ReplicationPad2d(1)` + valid 3x3 conv, net/CIDNet.py:21-24,32-35,39-42,50-53."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w)


# --------------------------------------------------------------------------------------
# LCA blocks
# --------------------------------------------------------------------------------------
def cab(x, y, p: Params, pre: str, heads: int):
    """Cross-attention block `CAB.forward`, net/LCA.py:19-41.  Attention is over channels:
    per head a (c/head x c/head) matrix from L2-normalised q,k rows of length H*W."""
    b, c, h, w = x.shape
    q = F.conv2d(F.conv2d(x, p[pre + ".q.weight"]), p[pre + ".q_dwconv.weight"], padding=1, groups=c)
    kv = F.conv2d(F.conv2d(y, p[pre + ".kv.weight"]), p[pre + ".kv_dwconv.weight"], padding=1,
                  groups=2 * c)
    k, v = kv.chunk(2, dim=1)
    q = q.reshape(b, heads, c // heads, h * w)
    k = k.reshape(b, heads, c // heads, h * w)
    v = v.reshape(b, heads, c // heads, h * w)
    q = F.normalize(q, dim=-1)                  # :30
    k = F.normalize(k, dim=-1)                  # :31
    attn = (q @ k.transpose(-2, -1)) * p[pre + ".temperature"]   # :33
    attn = attn.softmax(dim=-1)                 # :34
    out = (attn @ v).reshape(b, c, h, w)        # :36-38
    return F.conv2d(out, p[pre + ".project_out.weight"])         # :40


def iel(x, p: Params, pre: str):
    """Gated FFN `IEL.forward`, net/LCA.py:60-67."""
    x = F.conv2d(x, p[pre + ".project_in.weight"])
    c2 = x.shape[1]
    x = F.conv2d(x, p[pre + ".dwconv.weight"], padding=1, groups=c2)
    x1, x2 = x.chunk(2, dim=1)
    x1 = torch.tanh(F.conv2d(x1, p[pre + ".dwconv1.weight"], padding=1, groups=c2 // 2)) + x1
    x2 = torch.tanh(F.conv2d(x2, p[pre + ".dwconv2.weight"], padding=1, groups=c2 // 2)) + x2
    return F.conv2d(x1 * x2, p[pre + ".project_out.weight"])


def hv_lca(x, y, p: Params, pre: str, heads: int):
    """`HV_LCA.forward`, net/LCA.py:78-81: second stage has NO residual."""
    nw, nb = p[pre + ".norm.weight"], p[pre + ".norm.bias"]
    x = x + cab(layernorm_cf(x, nw, nb), layernorm_cf(y, nw, nb), p, pre + ".ffn", heads)
    return iel(layernorm_cf(x, nw, nb), p, pre + ".gdfn")


def i_lca(x, y, p: Params, pre: str, heads: int):
    """`I_LCA.forward`, net/LCA.py:90-93."""
    nw, nb = p[pre + ".norm.weight"], p[pre + ".norm.bias"]
    x = x + cab(layernorm_cf(x, nw, nb), layernorm_cf(y, nw, nb), p, pre + ".ffn", heads)
    return x + iel(layernorm_cf(x, nw, nb), p, pre + ".gdfn")


# --------------------------------------------------------------------------------------
# TNSM variant blocks (net/TNSM.py)
# --------------------------------------------------------------------------------------
def dynamic_noise_map(x, p: Params, pre: str):
    """`DynamicNoiseMap.forward`, net/TNSM.py:37-57 -> (B,1,H,W) noise map in (0,1)."""
    c = x.shape[1]
    avg = F.adaptive_avg_pool2d(x, 1)
    mx = F.adaptive_max_pool2d(x, 1)
    fc = lambda t: F.conv2d(F.relu(F.conv2d(t, p[pre + ".fc1.weight"])), p[pre + ".fc2.weight"])
    global_feat = torch.sigmoid(fc(avg) + fc(mx))
    local = F.conv2d(F.leaky_relu(F.conv2d(x, p[pre + ".noise_branch.0.weight"], padding=1, groups=c), 0.2),
                     p[pre + ".noise_branch.2.weight"])
    return torch.sigmoid(F.conv2d(global_feat * local, p[pre + ".final_conv.weight"]))


def noise_aware_attention(x, y, noise_map, p: Params, pre: str, heads: int):
    """`NoiseAwareAttentionCABStyle.forward`, net/TNSM.py:83-128: CAB without the L2 normalisation of
    q,k, with v modulated by sigmoid(conv1x1(noise_map))."""
    b, c, h, w = x.shape
    q = F.conv2d(F.conv2d(x, p[pre + ".q.weight"]), p[pre + ".q_dwconv.weight"], padding=1, groups=c)
    kv = F.conv2d(F.conv2d(y, p[pre + ".kv.weight"]), p[pre + ".kv_dwconv.weight"], padding=1, groups=2 * c)
    k, v = kv.chunk(2, dim=1)
    q = q.reshape(b, heads, c // heads, h * w)
    k = k.reshape(b, heads, c // heads, h * w)
    v = v.reshape(b, heads, c // heads, h * w)
    attn = ((q @ k.transpose(-2, -1)) * p[pre + ".temperature"]).softmax(dim=-1)
    keep = torch.sigmoid(F.conv2d(noise_map, p[pre + ".noise_scaler.0.weight"]))
    v = v * keep.reshape(b, heads, c // heads, h * w)
    out = (attn @ v).reshape(b, c, h, w)
    return F.conv2d(out, p[pre + ".project_out.weight"])


def adaptive_filter(x, noise_map, p: Params, pre: str):
    """`AdaptiveFilter.forward`, net/TNSM.py:155-173."""
    c = x.shape[1]
    nb = F.conv2d(F.leaky_relu(F.conv2d(x, p[pre + ".noise_process.0.weight"], padding=1, groups=c), 0.2),
                  p[pre + ".noise_process.2.weight"])
    db = F.conv2d(F.leaky_relu(F.conv2d(x, p[pre + ".detail_preserve.0.weight"]), 0.2),
                  p[pre + ".detail_preserve.2.weight"], padding=1, groups=c)
    fused = torch.cat([noise_map * nb, (1.0 - noise_map) * db], dim=1)
    out = F.conv2d(fused, p[pre + ".fusion.weight"])
    return layernorm_cf(out, p[pre + ".norm.weight"], p[pre + ".norm.bias"])


def tnsm_block(x, y, p: Params, pre: str, heads: int):
    """`TrainableNoiseSuppression.forward`, net/TNSM.py:196-215 (HV_TNSM / I_TNSM wrap it as `.tnsm`)."""
    pre = pre + ".tnsm"
    nm = dynamic_noise_map(x, p, pre + ".noise_map_generator")
    n1w, n1b = p[pre + ".norm1.weight"], p[pre + ".norm1.bias"]
    x = x + noise_aware_attention(layernorm_cf(x, n1w, n1b), layernorm_cf(y, n1w, n1b), nm, p, pre + ".noise_attention", heads)
    x = x + adaptive_filter(layernorm_cf(x, p[pre + ".norm2.weight"], p[pre + ".norm2.bias"]), nm, p, pre + ".adaptive_filter")
    return x, nm


def cidnet_tnsm_forward(p: Params, x: torch.Tensor, heads=(1, 2, 4, 8), training: bool = True, this_k: Optional[float] = None):
    """`CIDNet_TNSM.forward` (use_tnsm=True), net/CIDNet_TNSM.py:101-294 -> (rgb, fused_noise or None).
    Same wiring quirks as the base model (level-3 encoders see the pre-LCA2 tensors; ID_block2 is fed
    i_dec3, so I_LCA5 and I_TNSM5 are dead); twelve noise maps are resized (bilinear,
    align_corners=False) and fused by conv3x3(12->3)+sigmoid in training mode only (:248-266)."""
    _, h2, h3, h4 = heads
    k = p["trans.density_k"]
    hvi = hvit(x, k)
    if this_k is None:
        this_k = float(k.detach().reshape(-1)[0])
    maps = []
    i_enc0 = rep_conv3x3(hvi[:, 2:3], p["IE_block0.1.weight"])
    i_enc1 = norm_downsample(i_enc0, p, "IE_block1")
    hv_0 = rep_conv3x3(hvi, p["HVE_block0.1.weight"])
    hv_1 = norm_downsample(hv_0, p, "HVE_block1")
    i_jump0, hv_jump0 = i_enc0, hv_0

    def stage(i_in, hv_in, n, hd):
        """I_LCAn / HV_LCAn followed by I_TNSMn / HV_TNSMn (both TNSMs see the LCA outputs)"""
        i_l = i_lca(i_in, hv_in, p, f"I_LCA{n}", hd)
        hv_l = hv_lca(hv_in, i_in, p, f"HV_LCA{n}", hd)
        i_t, i_n = tnsm_block(i_l, hv_l, p, f"I_TNSM{n}", hd)
        hv_t, hv_n = tnsm_block(hv_l, i_l, p, f"HV_TNSM{n}", hd)
        maps.extend([i_n, hv_n])
        return i_t, hv_t

    i_enc2, hv_2 = stage(i_enc1, hv_1, 1, h2)
    v_jump1, hv_jump1 = i_enc2, hv_2
    i_enc2 = norm_downsample(i_enc2, p, "IE_block2")
    hv_2 = norm_downsample(hv_2, p, "HVE_block2")
    v_jump2, hv_jump2 = stage(i_enc2, hv_2, 2, h3)
    i_enc3 = norm_downsample(i_enc2, p, "IE_block3")
    hv_3 = norm_downsample(hv_2, p, "HVE_block3")
    i_enc4, hv_4 = stage(i_enc3, hv_3, 3, h4)
    i_dec4, hv_4 = stage(i_enc4, hv_4, 4, h4)
    hv_3 = norm_upsample(hv_4, hv_jump2, p, "HVD_block3")
    i_dec3 = norm_upsample(i_dec4, v_jump2, p, "ID_block3")
    i_dec2_dead, hv_2 = stage(i_dec3, hv_3, 5, h3)        # the I branch result is discarded (:213 vs :226)
    hv_2 = norm_upsample(hv_2, hv_jump1, p, "HVD_block2")
    i_dec2 = norm_upsample(i_dec3, v_jump1, p, "ID_block2")
    i_dec1, hv_1 = stage(i_dec2, hv_2, 6, h2)
    i_dec1 = norm_upsample(i_dec1, i_jump0, p, "ID_block1")
    i_dec0 = rep_conv3x3(i_dec1, p["ID_block0.1.weight"])
    hv_1 = norm_upsample(hv_1, hv_jump0, p, "HVD_block1")
    hv_0 = rep_conv3x3(hv_1, p["HVD_block0.1.weight"])
    rgb = phvit(torch.cat([hv_0, i_dec0], dim=1) + hvi, this_k)
    if not training:
        return rgb, None
    H, W = rgb.shape[-2:]
    resized = [m if m.shape[-2:] == (H, W) else F.interpolate(m, size=(H, W), mode="bilinear", align_corners=False) for m in maps]
    fused = torch.sigmoid(F.conv2d(torch.cat(resized, dim=1), p["noise_fusion.0.weight"], padding=1))
    return rgb, fused


def spatial_attention(x, w):
    """`SpatialAttention.forward` of the MSSA variant, net/CIDNet_MSSA.py:20-25:
    x * sigmoid(conv7x7([mean_c(x), max_c(x)])), zero pad 3, no bias."""
    avg = torch.mean(x, dim=1, keepdim=True)
    mx, _ = torch.max(x, dim=1, keepdim=True)
    y = F.conv2d(torch.cat([avg, mx], dim=1), w, padding=w.shape[-1] // 2)
    return x * torch.sigmoid(y)


# --------------------------------------------------------------------------------------
# training losses ("next" row f1)
# --------------------------------------------------------------------------------------
def ssim_window(window_size: int = 11, sigma: float = 1.5, channel: int = 3) -> torch.Tensor:
    """`create_window` / `gaussian`, loss/loss_utils.py:113-123: normalised Gaussian, outer product, one copy per channel."""
    import math
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    g = (g / torch.sum(g)).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim_map(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """`map_ssim` before its mean, loss/loss_utils.py:125-140 (depthwise Gaussian filtering with zero padding)."""
    c = img1.shape[1]
    w = ssim_window(window_size, 1.5, c).to(img1.dtype)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=c)
    mu2 = F.conv2d(img2, w, padding=pad, groups=c)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=c) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=c) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=c) - mu1_mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))


def ssim_loss(img1: torch.Tensor, img2: torch.Tensor, weight: float = 1.0, window_size: int = 11) -> torch.Tensor:
    """`SSIM.forward`, loss/losses.py:166-190 (size_average=True): (1 - mean(ssim_map)) * weight."""
    return (1.0 - ssim_map(img1, img2, window_size).mean()) * weight


def edge_laplacian(img: torch.Tensor) -> torch.Tensor:
    """`EdgeLoss.laplacian_kernel`, loss/losses.py:52-59 (with `conv_gauss`, :47-50): 5x5 blur outer([.05,.25,.4,.25,.05]),
    replicate padding, depthwise; blur -> keep even pixels x4 (zeros elsewhere) -> blur; current - that.
    RESTATED FROM THE TEXT, parity unpinned: loss.losses cannot be imported here (it imports torchvision for the VGG
    loss) and EdgeLoss.__init__ hard-codes .cuda(); only its reduction (`mse_loss`) is importable from loss_utils."""
    c = img.shape[1]
    k = torch.tensor([[.05, .25, .4, .25, .05]], dtype=torch.float32)
    kern = torch.matmul(k.t(), k).unsqueeze(0).repeat(c, 1, 1, 1).to(img.dtype)

    def conv_gauss(t):
        return F.conv2d(F.pad(t, (2, 2, 2, 2), mode="replicate"), kern, groups=c)

    filtered = conv_gauss(img)
    new_filter = torch.zeros_like(filtered)
    new_filter[:, :, ::2, ::2] = filtered[:, :, ::2, ::2] * 4
    return img - conv_gauss(new_filter)


def edge_loss(x: torch.Tensor, y: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
    """`EdgeLoss.forward`, loss/losses.py:61-63: mse_loss(laplacian(x), laplacian(y)) (mean) * weight."""
    return F.mse_loss(edge_laplacian(x), edge_laplacian(y)) * weight


def tnsm_noise_losses(noise_map: torch.Tensor, output_rgb: torch.Tensor, im1: torch.Tensor):
    """(noise_consistency_loss, noise_smoothing_loss) of the TNSM training script, train_tnsm.py:68-70.  PARITY UNPINNED:
    the formula lives inline in a script whose imports (torchvision datasets) fail here, so this is a restatement of
    those three lines from the text; F.sigmoid there is torch.sigmoid."""
    target = 1.0 - torch.sigmoid(torch.mean(torch.abs(output_rgb - im1), dim=1, keepdim=True))
    consistency = torch.mean(torch.abs(noise_map - target))
    smoothing = (torch.mean(torch.abs(noise_map[:, :, :, :-1] - noise_map[:, :, :, 1:]))
                 + torch.mean(torch.abs(noise_map[:, :, :-1, :] - noise_map[:, :, 1:, :])))
    return consistency, smoothing


# --------------------------------------------------------------------------------------
# whole network
# --------------------------------------------------------------------------------------
def cidnet_forward(p: Params, x: torch.Tensor, heads=(1, 2, 4, 8), this_k: Optional[float] = None,
                   gated=False, alpha_s=1.3, gated2=False, alpha=1.0, taps: Optional[dict] = None,
                   variant: str = "base"):
    """`CIDNet.forward`, net/CIDNet.py:71-122, including its two wiring quirks: level-3 encoders
    consume the PRE-LCA2 tensors (:94-95) and ID_block2 consumes i_dec3, so I_LCA5's result is
    dead (:105,109).  variant="mssa" restates net/CIDNet_MSSA.py:100-159 instead: a SpatialAttention
    gate after every up block (:133,135,142,144,150,153) and ID_block2 fed by I_LCA5's output (:143),
    so I_LCA5 is live there.  `this_k` defaults to the current value of density_k, which is what the
    reference's HVIT side effect (HVI_transform.py:38) leaves for PHVIT.  `taps`, if given,
    receives named intermediate activations (for per-stage parity tests)."""
    _, h2, h3, h4 = heads
    k = p["trans.density_k"]
    hvi = hvit(x, k)
    if this_k is None:
        this_k = float(k.detach().reshape(-1)[0])
    i = hvi[:, 2:3]
    i_enc0 = rep_conv3x3(i, p["IE_block0.1.weight"])
    i_enc1 = norm_downsample(i_enc0, p, "IE_block1")
    hv_0 = rep_conv3x3(hvi, p["HVE_block0.1.weight"])
    hv_1 = norm_downsample(hv_0, p, "HVE_block1")
    i_jump0, hv_jump0, hv_enc1 = i_enc0, hv_0, hv_1

    i_enc2 = i_lca(i_enc1, hv_1, p, "I_LCA1", h2)
    hv_2 = hv_lca(hv_1, i_enc1, p, "HV_LCA1", h2)
    v_jump1, hv_jump1 = i_enc2, hv_2
    i_enc2 = norm_downsample(i_enc2, p, "IE_block2")
    hv_2 = norm_downsample(hv_2, p, "HVE_block2")

    v_jump2 = i_lca(i_enc2, hv_2, p, "I_LCA2", h3)
    hv_jump2 = hv_lca(hv_2, i_enc2, p, "HV_LCA2", h3)
    i_enc3 = norm_downsample(i_enc2, p, "IE_block3")        # quirk 1: pre-LCA2 input
    hv_3 = norm_downsample(hv_2, p, "HVE_block3")

    i_enc4 = i_lca(i_enc3, hv_3, p, "I_LCA3", h4)
    hv_4 = hv_lca(hv_3, i_enc3, p, "HV_LCA3", h4)
    i_dec4 = i_lca(i_enc4, hv_4, p, "I_LCA4", h4)
    hv_4 = hv_lca(hv_4, i_enc4, p, "HV_LCA4", h4)

    mssa = variant == "mssa"
    sa = (lambda t, name: spatial_attention(t, p[name + ".conv1.weight"])) if mssa else (lambda t, name: t)
    hv_3 = sa(norm_upsample(hv_4, hv_jump2, p, "HVD_block3"), "sa_hv3")
    i_dec3 = sa(norm_upsample(i_dec4, v_jump2, p, "ID_block3"), "sa_i3")
    # base: I_LCA5(i_dec3, hv_3) is computed and discarded by the reference (:105); skipped here.
    i_dec2_in = i_lca(i_dec3, hv_3, p, "I_LCA5", h3) if mssa else i_dec3
    hv_2 = hv_lca(hv_3, i_dec3, p, "HV_LCA5", h3)

    hv_2 = sa(norm_upsample(hv_2, hv_jump1, p, "HVD_block2"), "sa_hv2")
    i_dec2 = sa(norm_upsample(i_dec2_in, v_jump1, p, "ID_block2"), "sa_i2")  # base quirk 2: i_dec3 goes in

    i_dec1 = i_lca(i_dec2, hv_2, p, "I_LCA6", h2)
    hv_1 = hv_lca(hv_2, i_dec2, p, "HV_LCA6", h2)

    i_dec1 = sa(norm_upsample(i_dec1, i_jump0, p, "ID_block1"), "sa_i1")
    i_dec0 = rep_conv3x3(i_dec1, p["ID_block0.1.weight"])
    hv_1 = sa(norm_upsample(hv_1, hv_jump0, p, "HVD_block1"), "sa_hv1")
    hv_0 = rep_conv3x3(hv_1, p["HVD_block0.1.weight"])

    out_hvi = torch.cat([hv_0, i_dec0], dim=1) + hvi
    if taps is not None:
        taps.update(hvi=hvi, i_enc1=i_enc1, hv_enc1=hv_enc1,
                    v_jump1=v_jump1, hv_jump1=hv_jump1, v_jump2=v_jump2, hv_jump2=hv_jump2,
                    i_dec4=i_dec4, hv_4=hv_4, i_dec3=i_dec3, i_dec2=i_dec2, i_dec1=i_dec1,
                    hv_dec1=hv_1, out_hvi=out_hvi)
    return phvit(out_hvi, this_k, gated, alpha_s, gated2, alpha)


# --------------------------------------------------------------------------------------
# deterministic, torch-RNG-independent parameters
# --------------------------------------------------------------------------------------
def param_shapes(channels=(36, 36, 72, 144), heads=(1, 2, 4, 8), variant: str = "base", norm: bool = False) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of the reference's 191 state_dict tensors (net/CIDNet.py:17-69),
    in the reference's registration order; variant="mssa" appends the six SpatialAttention convs
    (net/CIDNet_MSSA.py:91-97) for 197 tensors; norm=True adds the LayerNorm of every down / up block
    (CIDNet(norm=True): net/transformer_utils.py:35-36,54-55; 24 more tensors)."""
    c1, c2, c3, c4 = channels
    _, h2, h3, h4 = heads
    s: Dict[str, Tuple[int, ...]] = {}

    def down(pre, ci, co):
        if norm:
            s[pre + ".norm.weight"] = (co,)
            s[pre + ".norm.bias"] = (co,)
        s[pre + ".prelu.weight"] = (1,)
        s[pre + ".down.0.weight"] = (co, ci, 3, 3)

    def up(pre, ci, co):
        if norm:
            s[pre + ".norm.weight"] = (co,)
            s[pre + ".norm.bias"] = (co,)
        s[pre + ".prelu.weight"] = (1,)
        s[pre + ".up_scale.0.weight"] = (co, ci, 3, 3)
        s[pre + ".up.weight"] = (co, 2 * co, 1, 1)

    def iel_(pre, d):
        hid = int(d * 2.66)
        s[pre + ".project_in.weight"] = (2 * hid, d, 1, 1)
        s[pre + ".dwconv.weight"] = (2 * hid, 1, 3, 3)
        s[pre + ".dwconv1.weight"] = (hid, 1, 3, 3)
        s[pre + ".dwconv2.weight"] = (hid, 1, 3, 3)
        s[pre + ".project_out.weight"] = (d, hid, 1, 1)

    def cab_(pre, d, nh):
        s[pre + ".temperature"] = (nh, 1, 1)
        s[pre + ".q.weight"] = (d, d, 1, 1)
        s[pre + ".q_dwconv.weight"] = (d, 1, 3, 3)
        s[pre + ".kv.weight"] = (2 * d, d, 1, 1)
        s[pre + ".kv_dwconv.weight"] = (2 * d, 1, 3, 3)
        s[pre + ".project_out.weight"] = (d, d, 1, 1)

    def hv_lca_(pre, d, nh):          # registration order: gdfn, norm, ffn (net/LCA.py:74-76)
        iel_(pre + ".gdfn", d)
        s[pre + ".norm.weight"] = (d,)
        s[pre + ".norm.bias"] = (d,)
        cab_(pre + ".ffn", d, nh)

    def i_lca_(pre, d, nh):           # registration order: norm, gdfn, ffn (net/LCA.py:86-88)
        s[pre + ".norm.weight"] = (d,)
        s[pre + ".norm.bias"] = (d,)
        iel_(pre + ".gdfn", d)
        cab_(pre + ".ffn", d, nh)

    s["HVE_block0.1.weight"] = (c1, 3, 3, 3)
    down("HVE_block1", c1, c2); down("HVE_block2", c2, c3); down("HVE_block3", c3, c4)
    up("HVD_block3", c4, c3); up("HVD_block2", c3, c2); up("HVD_block1", c2, c1)
    s["HVD_block0.1.weight"] = (2, c1, 3, 3)
    s["IE_block0.1.weight"] = (c1, 1, 3, 3)
    down("IE_block1", c1, c2); down("IE_block2", c2, c3); down("IE_block3", c3, c4)
    up("ID_block3", c4, c3); up("ID_block2", c3, c2); up("ID_block1", c2, c1)
    s["ID_block0.1.weight"] = (1, c1, 3, 3)
    for n, (d, nh) in enumerate([(c2, h2), (c3, h3), (c4, h4), (c4, h4), (c3, h3), (c2, h2)], 1):
        hv_lca_(f"HV_LCA{n}", d, nh)
    for n, (d, nh) in enumerate([(c2, h2), (c3, h3), (c4, h4), (c4, h4), (c3, h3), (c2, h2)], 1):
        i_lca_(f"I_LCA{n}", d, nh)
    if variant == "tnsm":
        def tnsm_(pre, d, nh):
            r = max(8, d // 4)
            g = pre + ".tnsm.noise_map_generator"
            s[g + ".fc1.weight"] = (r, d, 1, 1)
            s[g + ".fc2.weight"] = (d, r, 1, 1)
            s[g + ".noise_branch.0.weight"] = (d, 1, 3, 3)
            s[g + ".noise_branch.2.weight"] = (d, d, 1, 1)
            s[g + ".final_conv.weight"] = (1, d, 1, 1)
            a = pre + ".tnsm.noise_attention"
            s[a + ".temperature"] = (nh, 1, 1)
            s[a + ".q.weight"] = (d, d, 1, 1)
            s[a + ".q_dwconv.weight"] = (d, 1, 3, 3)
            s[a + ".kv.weight"] = (2 * d, d, 1, 1)
            s[a + ".kv_dwconv.weight"] = (2 * d, 1, 3, 3)
            s[a + ".noise_scaler.0.weight"] = (d, 1, 1, 1)
            s[a + ".project_out.weight"] = (d, d, 1, 1)
            f = pre + ".tnsm.adaptive_filter"
            s[f + ".noise_process.0.weight"] = (d, 1, 3, 3)
            s[f + ".noise_process.2.weight"] = (d, d, 1, 1)
            s[f + ".detail_preserve.0.weight"] = (d, d, 1, 1)
            s[f + ".detail_preserve.2.weight"] = (d, 1, 3, 3)
            s[f + ".fusion.weight"] = (d, 2 * d, 1, 1)
            s[f + ".norm.weight"] = (d,)
            s[f + ".norm.bias"] = (d,)
            for nn_ in ("norm1", "norm2"):
                s[pre + f".tnsm.{nn_}.weight"] = (d,)
                s[pre + f".tnsm.{nn_}.bias"] = (d,)
        lv = [(c2, h2), (c3, h3), (c4, h4), (c4, h4), (c3, h3), (c2, h2)]
        for n, (d, nh) in enumerate(lv, 1):
            tnsm_(f"HV_TNSM{n}", d, nh)
        for n, (d, nh) in enumerate(lv, 1):
            tnsm_(f"I_TNSM{n}", d, nh)
    s["trans.density_k"] = (1,)
    if variant == "tnsm":
        s["noise_fusion.0.weight"] = (3, 12, 3, 3)
    if variant == "mssa":
        for n in ("sa_hv3", "sa_i3", "sa_hv2", "sa_i2", "sa_hv1", "sa_i1"):
            s[n + ".conv1.weight"] = (1, 2, 7, 7)
    return s


def _key_seed(seed: int, key: str) -> int:
    h = 1469598103934665603 ^ (seed * 1099511628211 & 0xFFFFFFFFFFFFFFFF)      # FNV-1a 64
    for ch in key.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def make_params(seed: int = 0, channels=(36, 36, 72, 144), heads=(1, 2, 4, 8), jitter: bool = True,
                dtype=torch.float32, variant: str = "base", norm: bool = False) -> Params:
    """Counter-based deterministic parameters w = f(seed, key, shape) from numpy's PCG64 (stable
    across numpy/torch versions).  Conv weights ~ U(+-1/sqrt(fan_in)) (PyTorch's default
    Kaiming-uniform a=sqrt(5) bound).  With `jitter` the unit/zero-initialised tensors (LayerNorm
    affine, PReLU slope, temperature, density_k) are perturbed so that parity tests see every
    parameter; without it they take the reference's init values (1/0, 0.25, 1, 0.2)."""
    import numpy as np
    out: Params = {}
    for key, shape in param_shapes(channels, heads, variant, norm).items():
        rng = np.random.Generator(np.random.PCG64(_key_seed(seed, key)))
        u = rng.random(shape, dtype=np.float64) * 2.0 - 1.0
        if re.search(r"norm\d*\.weight$", key):
            v = 1.0 + (0.2 * u if jitter else 0.0 * u)
        elif re.search(r"norm\d*\.bias$", key):
            v = 0.1 * u if jitter else 0.0 * u
        elif key.endswith("prelu.weight"):
            v = 0.25 + (0.1 * u if jitter else 0.0 * u)
        elif key.endswith("temperature"):
            v = 1.0 + (0.3 * u if jitter else 0.0 * u)
        elif key == "trans.density_k":
            v = 0.2 + (0.05 * u if jitter else 0.0 * u)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            v = u / math.sqrt(fan_in)
        out[key] = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
    return out


def params_to(p: Params, dtype=None, requires_grad: Optional[bool] = None) -> Params:
    q = {}
    for k, v in p.items():
        t = v.detach().clone()
        if dtype is not None:
            t = t.to(dtype)
        if requires_grad is not None:
            t.requires_grad_(requires_grad)
        q[k] = t
    return q


def synthetic_batch(seed: int, shape, quantised: bool = False) -> torch.Tensor:
    """Synthetic images in [0,1): U[0,1) fp32, or uint8-quantised k/255 (many ties and zeros)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(_key_seed(seed, "batch")))
    if quantised:
        a = rng.integers(0, 256, size=shape).astype(np.float32) / np.float32(255.0)
    else:
        a = rng.random(shape, dtype=np.float32)
    return torch.from_numpy(a)


VGG19_NAMES = ("conv1_1", "conv1_2", "pool", "conv2_1", "conv2_2", "pool", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "pool",
               "conv4_1", "conv4_2", "conv4_3", "conv4_4", "pool", "conv5_1", "conv5_2", "conv5_3", "conv5_4")


def vgg_features(x: torch.Tensor, convs: Dict[str, Tuple[torch.Tensor, torch.Tensor]], layer_names, range_norm: bool = True,
                 use_input_norm: bool = True) -> Dict[str, torch.Tensor]:
    """Restates VGGFeatureExtractor.forward (loss/vgg_arch.py:217-239): (x + 1) / 2 if range_norm, ImageNet mean / std,
    then conv(+bias) / ReLU / MaxPool2d(2, 2) in VGG19 order; the feature named 'convN_M' is the conv output BEFORE its
    ReLU.  PARITY UNPINNED: loss.vgg_arch imports torchvision (absent here) and fetches pretrained weights, so this
    is restated from the text; `convs` = {name: (weight, bias)}."""
    if range_norm:
        x = (x + 1) / 2
    if use_input_norm:
        mean = torch.tensor([0.485, 0.456, 0.406], dtype=x.dtype).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], dtype=x.dtype).view(1, 3, 1, 1)
        x = (x - mean) / std
    last = max(VGG19_NAMES.index(n) for n in layer_names)
    out = {}
    for i, name in enumerate(VGG19_NAMES[:last + 1]):
        if name == "pool":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
            continue
        w, b = convs[name]
        x = F.conv2d(x, w.to(x.dtype), b.to(x.dtype), padding=1)
        if name in layer_names:
            out[name] = x
        if i < last:
            x = F.relu(x)
    return out


def perceptual_loss(x, gt, convs, layer_weights: Dict[str, float], perceptual_weight: float = 1.0, range_norm: bool = True):
    """PerceptualLoss.forward with criterion 'mse' and no style term (loss/losses.py:126-142, train.py:192)."""
    names = list(layer_weights.keys())
    fx = vgg_features(x, convs, names, range_norm)
    fg = vgg_features(gt.detach(), convs, names, range_norm)
    loss = 0
    for k in names:
        loss = loss + F.mse_loss(fx[k], fg[k]) * layer_weights[k]
    return loss * perceptual_weight


def grad_fingerprint(t: torch.Tensor, max_sample: int = 2048):
    """Compact fixture form of a large tensor: (sums, sample).  sums = [sum, sum|.|, dot with a fixed
    non-periodic weight vector] in fp64 (the dot catches permuted / shifted elements that the plain sums
    miss); sample = every stride-th element of the flattened tensor (at most `max_sample`).  Used by
    oracle/gen_golden.py to store full-size gradients compactly and by the GPU parity tests to compare."""
    f = t.detach().reshape(-1).double().cpu()
    w = torch.cos(torch.arange(f.numel(), dtype=torch.float64) * 0.7390851332151607)
    sums = torch.stack([f.sum(), f.abs().sum(), (f * w).sum()])
    stride = max(1, -(-f.numel() // max_sample))
    return sums, f[::stride].float(), stride
