"""`CIDNet` of the MSSA variant: drop-in for the reference's net/CIDNet_MSSA.py (the class that
train.py / eval.py actually import, train.py:10): CIDNet + six SpatialAttention gates after the up
blocks, and ID_block2 fed by I_LCA5's output (so I_LCA5 is live).  197 state_dict tensors."""
import torch.nn as nn

from . import ops
from .cidnet import CIDNet as _BaseCIDNet


class SpatialAttention(nn.Module):
    """Reference: net/CIDNet_MSSA.py:10-25."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 7), "kernel size must be 3 or 7"
        if kernel_size != 7:
            raise NotImplementedError("SpatialAttention: CIDNet_MSSA uses the 7x7 kernel; 3x3 is not implemented")
        self.conv1 = nn.Conv2d(2, 1, kernel_size, padding=3, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return ops.SpatialAttentionFn.apply(x, self.conv1.weight)


class CIDNet(_BaseCIDNet):
    """Reference: net/CIDNet_MSSA.py:28-159."""

    def __init__(self, channels=[36, 36, 72, 144], heads=[1, 2, 4, 8], norm=False):
        super().__init__(channels=channels, heads=heads, norm=norm)
        self.sa_hv3 = SpatialAttention()
        self.sa_i3 = SpatialAttention()
        self.sa_hv2 = SpatialAttention()
        self.sa_i2 = SpatialAttention()
        self.sa_hv1 = SpatialAttention()
        self.sa_i1 = SpatialAttention()

    def _gate(self, name, t):                    # net/CIDNet_MSSA.py:133,135,142,144,150,153
        return getattr(self, name)(t)

    def _stage5(self, i_dec3, hv_3):             # net/CIDNet_MSSA.py:137,143: ID_block2 is fed I_LCA5's output -- a live LCA pair
        return self._lca_pair(self.I_LCA5, self.HV_LCA5, i_dec3, hv_3)
