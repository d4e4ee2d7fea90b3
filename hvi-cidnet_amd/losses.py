"""Training losses on device ("next" row f1 of SURVEY section 8): drop-ins for the reference's `L1Loss` and `SSIM`
(loss/losses.py:10-38, 166-190) and `EdgeLoss` (:41-65) with the same constructor arguments.  The VGG19
PerceptualLoss (:68-161) needs downloaded weights and is not built; `CIDNetLoss` is the reference's training objective
(train.py:61-65) without that term."""
import torch.nn as nn

from . import ops


class L1Loss(nn.Module):
    """mean(|pred - target|) * loss_weight (reduction='mean', the only mode train.py uses)."""

    def __init__(self, loss_weight=1.0, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise ValueError(f"Unsupported reduction mode: {reduction}. hvi-cidnet_amd builds 'mean' (train.py:189)")
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred, target, weight=None, **kwargs):
        if weight is not None:
            raise NotImplementedError("element-wise loss weights are not used by the reference's training loop")
        loss = ops.L1LossFn.apply(pred, target)
        return loss if self.loss_weight == 1.0 else loss * self.loss_weight


class SSIM(nn.Module):
    """(1 - mean SSIM map) * weight with the reference's 11x11 Gaussian window (sigma 1.5, zero padding)."""

    def __init__(self, window_size=11, size_average=True, weight=1.0):
        super().__init__()
        if window_size != 11 or not size_average:
            raise NotImplementedError("hvi-cidnet_amd builds the SSIM loss as train.py constructs it: window 11, size_average=True")
        self.window_size = window_size
        self.size_average = size_average
        self.weight = weight

    def forward(self, img1, img2):
        return ops.SSIMLossFn.apply(img1, img2, float(self.weight))


class EdgeLoss(nn.Module):
    """Laplacian-pyramid edge loss: mse(laplacian(x), laplacian(y)) * loss_weight (5x5 blur, replicate padding)."""

    def __init__(self, loss_weight=1.0, reduction="mean"):
        super().__init__()
        self.weight = loss_weight

    def forward(self, x, y):
        return ops.EdgeLossFn.apply(x, y, float(self.weight))


class CIDNetLoss(nn.Module):
    """loss_rgb + HVI_weight * loss_hvi with loss_* = L1 + SSIM + Edge (train.py:61-65 without the perceptual term);
    defaults are data/options.py:56-59.  `model` supplies HVIT for the HVI-space terms, exactly as train.py calls
    `model.HVIT` on the output and on the ground truth."""

    def __init__(self, model, L1_weight=1.0, D_weight=0.5, E_weight=50.0, HVI_weight=1.0):
        super().__init__()
        self.l1 = L1Loss(loss_weight=L1_weight)
        self.ssim = SSIM(weight=D_weight)
        self.edge = EdgeLoss(loss_weight=E_weight)
        self.hvi_weight = HVI_weight
        self._hvit = model.HVIT

    def _terms(self, a, b):
        return self.l1(a, b) + self.ssim(a, b) + self.edge(a, b)

    def forward(self, output_rgb, gt_rgb):
        loss_hvi = self._terms(self._hvit(output_rgb), self._hvit(gt_rgb))
        loss_rgb = self._terms(output_rgb, gt_rgb)
        return loss_rgb + self.hvi_weight * loss_hvi
