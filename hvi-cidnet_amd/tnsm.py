"""Drop-ins for the reference's net/TNSM.py (trainable noise-suppression blocks): identical module tree
and parameter names, forward on the HIP ops."""
import torch
import torch.nn as nn

from . import ops
from .transformer_utils import LayerNorm


class DynamicNoiseMap(nn.Module):
    """Reference: net/TNSM.py:7-57."""

    def __init__(self, dim, reduction=4, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("TNSM blocks are bias-free in CIDNet_TNSM")
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        reduced_dim = max(8, dim // reduction)
        self.fc1 = nn.Conv2d(dim, reduced_dim, kernel_size=1, bias=bias)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(reduced_dim, dim, kernel_size=1, bias=bias)
        self.noise_branch = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=3, padding=1, groups=dim, bias=bias),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(dim, dim, kernel_size=1, bias=bias))
        self.final_conv = nn.Conv2d(dim, 1, kernel_size=1, bias=bias)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return ops.NoiseMapFn.apply(x, self.fc1.weight, self.fc2.weight, self.noise_branch[0].weight,
                                    self.noise_branch[2].weight, self.final_conv.weight)


class NoiseAwareAttentionCABStyle(nn.Module):
    """Reference: net/TNSM.py:59-128."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("TNSM blocks are bias-free in CIDNet_TNSM")
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=bias)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=3, stride=1, padding=1, groups=dim * 2, bias=bias)
        self.noise_scaler = nn.Sequential(nn.Conv2d(1, dim, kernel_size=1, bias=bias), nn.Sigmoid())
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)

    def forward(self, x, y, noise_map=None, residual=None):
        if noise_map is None:
            raise NotImplementedError("NoiseAwareAttentionCABStyle is always called with a noise map in CIDNet_TNSM")
        if residual is None:
            residual = torch.zeros_like(x)
        return ops.NoiseCABResidualFn.apply(residual, x, y, noise_map, self.temperature, self.q.weight, self.q_dwconv.weight,
                                            self.kv.weight, self.kv_dwconv.weight, self.noise_scaler[0].weight,
                                            self.project_out.weight, self.num_heads)


class AdaptiveFilter(nn.Module):
    """Reference: net/TNSM.py:130-173.  The fusion 1x1 conv is distributed over the concat
    (fusion([nm*a ; (1-nm)*d]) = nm * (Wf1 a) + (1-nm) * (Wf2 d), nm being one scalar per pixel)."""

    def __init__(self, dim, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("TNSM blocks are bias-free in CIDNet_TNSM")
        self.noise_process = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(dim, dim, kernel_size=1, bias=bias))
        self.detail_preserve = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=1, bias=bias),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias))
        self.fusion = nn.Conv2d(dim * 2, dim, kernel_size=1, bias=bias)
        self.norm = LayerNorm(dim)

    def forward(self, x, noise_map):
        c = x.shape[1]
        nb = ops.PwConvFn.apply(ops.DwConvFn.apply(x, self.noise_process[0].weight, True), self.noise_process[2].weight, 0, c)
        db = ops.DwConvFn.apply(ops.UnaryFn.apply(ops.PwConvFn.apply(x, self.detail_preserve[0].weight, 0, c), "leaky"),
                                self.detail_preserve[2].weight, False)
        a = ops.PwConvFn.apply(nb, self.fusion.weight, 0, c)
        d = ops.PwConvFn.apply(db, self.fusion.weight, c, c)
        return self.norm(ops.BlendFn.apply(a, d, noise_map))


class TrainableNoiseSuppression(nn.Module):
    """Reference: net/TNSM.py:175-215."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.noise_map_generator = DynamicNoiseMap(dim, bias=bias)
        self.noise_attention = NoiseAwareAttentionCABStyle(dim, num_heads, bias=bias)
        self.adaptive_filter = AdaptiveFilter(dim, bias=bias)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)

    def forward(self, x, y=None):
        if y is None:
            y = x
        noise_map = self.noise_map_generator(x)
        x = self.noise_attention(self.norm1(x), self.norm1(y), noise_map, residual=x)
        x = ops.AddFn.apply(x, self.adaptive_filter(self.norm2(x), noise_map))
        return x, noise_map


class HV_TNSM(nn.Module):
    """Reference: net/TNSM.py:218-224."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.tnsm = TrainableNoiseSuppression(dim, num_heads, bias)

    def forward(self, x, y):
        return self.tnsm(x, y)


class I_TNSM(nn.Module):
    """Reference: net/TNSM.py:227-232."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.tnsm = TrainableNoiseSuppression(dim, num_heads, bias)

    def forward(self, x, y):
        return self.tnsm(x, y)
