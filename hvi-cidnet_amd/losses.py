"""Training losses on device ("next" rows f1 and f4 of SURVEY section 8): drop-ins for the reference's `L1Loss` and `SSIM`
(loss/losses.py:10-38, 166-190), `EdgeLoss` (:41-65) and `PerceptualLoss` (:68-161 over loss/vgg_arch.py:133-239) with
the same constructor arguments.  `CIDNetLoss` is the reference's training objective (train.py:61-65).  The pretrained
VGG19 weights cannot be fetched here (torchvision / network absent): `VGGFeatureExtractor` starts from build-owned
deterministic random weights and loads a torchvision `vgg19` state_dict when one is supplied."""
from collections import OrderedDict

import torch
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


class VGGFeatureExtractor(nn.Module):
    """Parameter container with the reference's layout (loss/vgg_arch.py:133-239): `vgg_net.<convN_M>.weight / .bias`
    for the VGG19 prefix that reaches the deepest requested layer, buffers `mean` / `std`, everything frozen.
    torchvision's pretrained weights are not available offline; `load_torchvision_state_dict` takes a `vgg19().state_dict()`
    (keys `features.<i>.weight`) when the user has one."""

    def __init__(self, layer_name_list, vgg_type="vgg19", use_input_norm=True, range_norm=False, requires_grad=False,
                 remove_pooling=False, pooling_stride=2, seed=19):
        super().__init__()
        if vgg_type != "vgg19" or requires_grad or remove_pooling or pooling_stride != 2:
            raise NotImplementedError("hvi-cidnet_amd builds the extractor as train.py:192 constructs it: frozen vgg19, MaxPool2d(2, 2)")
        self.layer_name_list = list(layer_name_list)
        self.use_input_norm = use_input_norm
        self.range_norm = range_norm
        self.plan = ops.vgg_plan(self.layer_name_list)
        gen = torch.Generator().manual_seed(seed)          # identical on every rank, independent of the global RNG
        mods = OrderedDict()
        n_pool = 0
        for layer in self.plan:
            if layer == "pool":
                n_pool += 1
                mods[f"pool{n_pool}"] = nn.MaxPool2d(kernel_size=2, stride=2)
                continue
            name, ci, co = layer
            conv = nn.Conv2d(ci, co, kernel_size=3, padding=1)
            with torch.no_grad():                           # He-uniform, the scale real VGG activations live at
                bound = (6.0 / (9 * ci)) ** 0.5
                conv.weight.copy_((torch.rand(conv.weight.shape, generator=gen) * 2 - 1) * bound)
                conv.bias.copy_((torch.rand(co, generator=gen) * 2 - 1) * 0.05)
            mods[name] = conv
            mods[name.replace("conv", "relu")] = nn.ReLU(inplace=True)
        last = self.plan[-1][0].replace("conv", "relu")     # the reference cuts the stack at the deepest requested layer
        mods.pop(last)
        self.vgg_net = nn.Sequential(mods)
        for p in self.parameters():
            p.requires_grad = False
        if use_input_norm:
            self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
            self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def conv_params(self):
        out = []
        for layer in self.plan:
            if layer != "pool":
                conv = getattr(self.vgg_net, layer[0])
                out += [conv.weight, conv.bias]
        return out

    def load_torchvision_state_dict(self, sd):
        """sd: torchvision.models.vgg19().state_dict() (or its `features.` part)"""
        idx, mapped = 0, {}
        for layer in ops.VGG19_LAYERS:
            if layer == "pool":
                idx += 1
                continue
            mapped[layer[0]] = idx
            idx += 2                                        # conv, relu
        with torch.no_grad():
            for layer in self.plan:
                if layer == "pool":
                    continue
                conv = getattr(self.vgg_net, layer[0])
                i = mapped[layer[0]]
                conv.weight.copy_(sd.get(f"features.{i}.weight", sd.get(f"{i}.weight")))
                conv.bias.copy_(sd.get(f"features.{i}.bias", sd.get(f"{i}.bias")))


class PerceptualLoss(nn.Module):
    """Reference: PerceptualLoss, loss/losses.py:68-161, as train.py:192 builds it: layer_weights over 'convN_M' features
    (before the ReLU), criterion 'mse', no style term.  forward returns (percep_loss, None) like the reference."""

    def __init__(self, layer_weights, vgg_type="vgg19", use_input_norm=True, range_norm=True, perceptual_weight=1.0,
                 style_weight=0., criterion="l1"):
        super().__init__()
        if style_weight > 0:
            raise NotImplementedError("the style (Gram) term is never enabled by the reference's training loop (style_weight=0)")
        if criterion != "mse":
            raise NotImplementedError("hvi-cidnet_amd builds the criterion train.py:192 uses: 'mse'")
        self.perceptual_weight = perceptual_weight
        self.style_weight = style_weight
        self.layer_weights = dict(layer_weights)
        self.criterion_type = criterion
        self.vgg = VGGFeatureExtractor(list(layer_weights.keys()), vgg_type=vgg_type, use_input_norm=use_input_norm,
                                       range_norm=range_norm)

    def forward(self, x, gt):
        if not self.perceptual_weight > 0:
            return None, None
        names = list(self.layer_weights.keys())
        loss = ops.PerceptualLossFn.apply(x, gt.detach(), names, [float(self.layer_weights[k]) for k in names],
                                          float(self.perceptual_weight), self.vgg.range_norm, self.vgg.use_input_norm,
                                          *self.vgg.conv_params())
        return loss, None


class CIDNetLoss(nn.Module):
    """loss_rgb + HVI_weight * loss_hvi with loss_* = L1 + SSIM + Edge + P_weight * Perceptual (train.py:61-65);
    defaults are data/options.py:56-59 (P_weight 1e-2 there; 0 here leaves the VGG term out, the round-1 objective).
    `model` supplies HVIT for the HVI-space terms, exactly as train.py calls `model.HVIT` on the output and on the ground
    truth.  The perceptual term is built as train.py:192 does: conv1_2 / conv2_2 / conv3_4 / conv4_4, 'mse'."""

    def __init__(self, model, L1_weight=1.0, D_weight=0.5, E_weight=50.0, HVI_weight=1.0, P_weight=0.0, tnsm_weight=0.0):
        super().__init__()
        self.tnsm_weight = tnsm_weight
        self.l1 = L1Loss(loss_weight=L1_weight)
        self.ssim = SSIM(weight=D_weight)
        self.edge = EdgeLoss(loss_weight=E_weight)
        self.hvi_weight = HVI_weight
        self.p_weight = P_weight
        self.perceptual = PerceptualLoss({"conv1_2": 1, "conv2_2": 1, "conv3_4": 1, "conv4_4": 1}, perceptual_weight=1.0,
                                         criterion="mse") if P_weight > 0 else None
        self._hvit = model.HVIT

    def _terms(self, a, b):
        t = self.l1(a, b) + self.ssim(a, b) + self.edge(a, b)
        if self.perceptual is not None:
            t = t + self.p_weight * self.perceptual(a, b)[0]
        return t

    # dp.DataParallelTrainer passes the step's input as `im1=` to a loss function with this attribute
    wants_input = True

    def forward(self, output_rgb, gt_rgb, noise_map=None, im1=None):
        """`output_rgb`: the model's result -- a tensor, or CIDNet_TNSM's train-mode pair (rgb, fused noise map), which is
        unpacked here.  `noise_map`, `im1`: that second result and the RAW low-light image -- train_tnsm.py:69 uses im1 as
        loaded, before the optional `im1 ** gamma` the network is fed with (:55) -- with tnsm_weight > 0 they add
        train_tnsm.py:68-72's  tnsm_weight * (noise_consistency_loss + noise_smoothing_loss).  A positive tnsm_weight without
        a noise map raises: the reference's default is tnsm_weight = 1 (options.py:61) and silently training without the
        terms would be a different objective."""
        if isinstance(output_rgb, (tuple, list)):
            output_rgb, model_noise = output_rgb[0], output_rgb[1]
            noise_map = model_noise if noise_map is None else noise_map
        if self.tnsm_weight > 0 and noise_map is None:
            raise ValueError("CIDNetLoss(tnsm_weight > 0) needs the fused noise map: pass CIDNet_TNSM's train-mode result (rgb, noise) "
                             "or noise_map=...; a model in eval mode returns (rgb, None)")
        loss_hvi = self._terms(self._hvit(output_rgb), self._hvit(gt_rgb))
        loss_rgb = self._terms(output_rgb, gt_rgb)
        loss = loss_rgb + self.hvi_weight * loss_hvi
        if self.tnsm_weight > 0:
            if im1 is None:
                raise ValueError("CIDNetLoss: the TNSM noise terms need the network input im1 (train_tnsm.py:69)")
            loss = loss + tnsm_noise_loss(noise_map, output_rgb, im1, self.tnsm_weight)
        return loss


def tnsm_noise_loss(noise_map, output_rgb, im1, weight=1.0):
    """weight * (noise_consistency_loss + noise_smoothing_loss), train_tnsm.py:68-72 (options.py:61: tnsm_weight = 1)"""
    return ops.TNSMNoiseLossFn.apply(noise_map, output_rgb, im1.detach(), float(weight))
