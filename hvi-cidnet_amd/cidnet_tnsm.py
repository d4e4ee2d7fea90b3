"""`CIDNet_TNSM`: drop-in for the reference's net/CIDNet_TNSM.py (CIDNet + twelve trainable
noise-suppression blocks + a noise-map fusion head).  forward returns (rgb, fused_noise) in training mode
and (rgb, None) in eval mode, as the reference does.  468 state_dict tensors."""
import torch.nn as nn

from . import ops
from .cidnet import CIDNet as _BaseCIDNet
from .tnsm import HV_TNSM, I_TNSM


class CIDNet_TNSM(_BaseCIDNet):
    """Reference: net/CIDNet_TNSM.py:10-297."""

    def __init__(self, channels=[36, 36, 72, 144], heads=[1, 2, 4, 8], norm=False, use_tnsm=True):
        super().__init__(channels=channels, heads=heads, norm=norm)
        self.use_tnsm = use_tnsm
        [ch1, ch2, ch3, ch4] = channels
        [head1, head2, head3, head4] = heads
        trans = self.trans
        del self.trans                     # re-registered below so the key order equals the reference's
        if self.use_tnsm:
            lv = [(ch2, head2), (ch3, head3), (ch4, head4), (ch4, head4), (ch3, head3), (ch2, head2)]
            for n, (d, h) in enumerate(lv, 1):
                setattr(self, f"HV_TNSM{n}", HV_TNSM(d, h))
            for n, (d, h) in enumerate(lv, 1):
                setattr(self, f"I_TNSM{n}", I_TNSM(d, h))
        self.trans = trans
        if self.use_tnsm:
            self.noise_fusion = nn.Sequential(nn.Conv2d(12, 3, kernel_size=3, padding=1, bias=False), nn.Sigmoid())

    def _stage(self, n, i_in, hv_in, maps):
        """I_LCAn / HV_LCAn, then (use_tnsm) I_TNSMn / HV_TNSMn on the LCA outputs (CIDNet_TNSM.py:124-136 etc.);
        the I and HV halves of each pair run on the two branch streams"""
        i_l, hv_l = self._par(lambda: getattr(self, f"I_LCA{n}")(i_in, hv_in),
                              lambda: getattr(self, f"HV_LCA{n}")(hv_in, i_in), (i_in, hv_in))
        if not self.use_tnsm:
            return i_l, hv_l
        (i_t, i_n), (hv_t, hv_n) = self._par(lambda: getattr(self, f"I_TNSM{n}")(i_l, hv_l),
                                             lambda: getattr(self, f"HV_TNSM{n}")(hv_l, i_l), (i_l, hv_l))
        maps.extend([i_n, hv_n])
        return i_t, hv_t

    def forward(self, x):
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise RuntimeError(f"CIDNet_TNSM: H and W must be multiples of 8 (got {tuple(x.shape[2:])})")
        maps = []
        hvi = self.trans.HVIT(x)
        i = hvi[:, 2:3, :, :].contiguous()
        i_enc0 = self.IE_block0(i)
        i_enc1 = self.IE_block1(i_enc0)
        hv_0 = self.HVE_block0(hvi)
        hv_1 = self.HVE_block1(hv_0)
        i_jump0, hv_jump0 = i_enc0, hv_0

        i_enc2, hv_2 = self._stage(1, i_enc1, hv_1, maps)
        v_jump1, hv_jump1 = i_enc2, hv_2
        i_enc2 = self.IE_block2(i_enc2)
        hv_2 = self.HVE_block2(hv_2)
        v_jump2, hv_jump2 = self._stage(2, i_enc2, hv_2, maps)
        i_enc3 = self.IE_block3(i_enc2)                  # pre-LCA2 tensors, as in the base model
        hv_3 = self.HVE_block3(hv_2)
        i_enc4, hv_4 = self._stage(3, i_enc3, hv_3, maps)
        i_dec4, hv_4 = self._stage(4, i_enc4, hv_4, maps)
        hv_3 = self.HVD_block3(hv_4, hv_jump2)
        i_dec3 = self.ID_block3(i_dec4, v_jump2)
        _i_dead, hv_2 = self._stage(5, i_dec3, hv_3, maps)   # I branch result unused (CIDNet_TNSM.py:213 vs :226)
        hv_2 = self.HVD_block2(hv_2, hv_jump1)
        i_dec2 = self.ID_block2(i_dec3, v_jump1)
        i_dec1, hv_1 = self._stage(6, i_dec2, hv_2, maps)
        i_dec1 = self.ID_block1(i_dec1, i_jump0)
        i_dec0 = self.ID_block0(i_dec1)
        hv_1 = self.HVD_block1(hv_1, hv_jump0)
        hv_0 = self.HVD_block0(hv_1)
        rgb = self.trans.PHVIT_residual(hv_0, i_dec0, hvi)
        if self.use_tnsm and self.training:
            if not maps:
                raise ValueError("noise_maps list is empty during training with use_tnsm=True")
            stacked = ops.ResizeCatFn.apply(rgb.shape[-2], rgb.shape[-1], *maps)
            fused = ops.UnaryFn.apply(ops.Conv3x3Fn.apply(stacked, self.noise_fusion[0].weight), "sigmoid")
            return rgb, fused
        return rgb, None
