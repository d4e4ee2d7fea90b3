"""`RGB_HVI`: drop-in for the reference's net/HVI_transform.py module (same attributes and
state_dict key `density_k`), backed by the K1/K2 HIP kernels."""
import torch
import torch.nn as nn

from . import ops


class RGB_HVI(nn.Module):
    """Reference: `RGB_HVI`, net/HVI_transform.py:6-122.

    Attribute contract kept: `density_k` (Parameter, init 0.2), `gated`, `gated2`, `alpha`,
    `alpha_s`, `this_k` (= value of k at the most recent HVIT call, 0 before any; the eval scripts
    read/mutate these around inference, eval.py:46-55).  Unlike the reference, HVIT does not call
    `k.item()`: the kernels read k from device memory and `this_k` is materialised on the host only
    when somebody reads the attribute.
    """

    def __init__(self):
        super().__init__()
        self.density_k = nn.Parameter(torch.full([1], 0.2))
        self.gated = False
        self.gated2 = False
        self.alpha = 1.0
        self.alpha_s = 1.3
        self._this_k_host = 0          # python number once known on the host
        self._this_k_dev = None        # device snapshot of k taken by the last HVIT

    # -- this_k: lazily synchronised view of the last k ------------------------------------------
    @property
    def this_k(self):
        if self._this_k_dev is not None:
            self._this_k_host = self._this_k_dev.item()
            self._this_k_dev = None
        return self._this_k_host

    @this_k.setter
    def this_k(self, value):
        self._this_k_host = value
        self._this_k_dev = None

    def _k_for_phvit(self):
        return self._this_k_dev if self._this_k_dev is not None else float(self._this_k_host)

    def HVIT(self, img):
        out = ops.HVITFn.apply(img, self.density_k)
        self._this_k_dev = ops.snapshot(self.density_k)         # device copy by our own kernel, no host sync
        return out

    def PHVIT(self, img):
        return ops.PHVITFn.apply(img, None, None, self._k_for_phvit(), self.gated, self.alpha_s, self.gated2,
                                 self.alpha)

    def PHVIT_residual(self, hv, iv, hvi):
        """PHVIT(cat([hv, iv], 1) + hvi) with the add fused (net/CIDNet.py:119-120)."""
        return ops.PHVITFn.apply(hvi, hv, iv, self._k_for_phvit(), self.gated, self.alpha_s, self.gated2, self.alpha)
