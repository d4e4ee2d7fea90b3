"""Drop-ins for the reference's net/transformer_utils.py (LayerNorm, NormDownsample, NormUpsample):
same constructor arguments, sub-module tree and state_dict keys; forward runs the HIP ops."""
import torch
import torch.nn as nn

from . import ops


class LayerNorm(nn.Module):
    """Reference: net/transformer_utils.py:5-29.  Only `channels_first` is on the CIDNet hot path;
    `channels_last` (never used by the reference models) is rejected rather than routed to ATen."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_first"):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if self.data_format not in ["channels_last", "channels_first"]:
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)
        self._use = ops.LNUse()            # multi-use bookkeeping (ops.LNUse): in-place summing of this module's gradients
        # the norm of an LCA block feeds nothing but 1x1 convs (net/LCA.py:22-23,60): in the bf16 mode its OUTPUT is stored
        # as bf16 (ops.STORAGE); set by HV_LCA / I_LCA, never for the optional norms of the down / up blocks
        self.lca_internal = False

    def _out_dtype(self, x):
        return ops.lca_dtype(x) if self.lca_internal else torch.float32

    def forward(self, x):
        if self.data_format != "channels_first":
            raise NotImplementedError("hvi-cidnet_amd implements the channels_first LayerNorm of the CIDNet hot path")
        return ops.LayerNormCFFn.apply(x, self.weight, self.bias, self.eps,
                                       self._use if ops.needs_grad(x, self.weight, self.bias) else None, self._out_dtype(x))

    def forward_res(self, x):
        """(norm(x), x) for a pre-norm residual block: use the second value as the residual input (see
        ops.LayerNormResFn: the residual's gradient is then added inside the LayerNorm backward kernel)."""
        if self.data_format != "channels_first":
            raise NotImplementedError("hvi-cidnet_amd implements the channels_first LayerNorm of the CIDNet hot path")
        return ops.LayerNormResFn.apply(x, self.weight, self.bias, self.eps,
                                        self._use if ops.needs_grad(x, self.weight, self.bias) else None, self._out_dtype(x))

    def forward_dual(self, x, other):
        """(self(x), other(x), x): this module's norm (x-norm of its block; the third value is the block's residual input, as
        in forward_res) and the partner block's y-norm of the same tensor in one pass (ops.LayerNormDualFn).  Falls back
        to two passes when the shape has no dual kernel or the eps differ."""
        if not (ops.ln_dual_supported(x) and self.eps == other.eps and other.data_format == "channels_first"):
            n, xr = self.forward_res(x)
            return n, other(x), xr
        grad = ops.needs_grad(x, self.weight, self.bias, other.weight, other.bias)
        return ops.LayerNormDualFn.apply(x, self.weight, self.bias, other.weight, other.bias, self.eps, self._use if grad else None)


class NormDownsample(nn.Module):
    """Reference: net/transformer_utils.py:31-48.  `down` keeps the reference's Sequential(Conv2d,
    UpsamplingBilinear2d) as the parameter container (key `down.0.weight`); the fused HIP path
    conv3x3 -> bilinear(align_corners) -> PReLU replaces calling it."""

    def __init__(self, in_ch, out_ch, scale=0.5, use_norm=False):
        super().__init__()
        if scale != 0.5:
            raise NotImplementedError("NormDownsample: only scale=0.5 (the CIDNet configuration) is implemented")
        self.use_norm = use_norm
        if self.use_norm:
            self.norm = LayerNorm(out_ch)
        self.prelu = nn.PReLU()
        self.down = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=False),
            nn.UpsamplingBilinear2d(scale_factor=scale))

    def forward(self, x):
        ws = (self.down[0].weight, self.prelu.weight)
        x = ops.DownFn.apply(x, *ws, ops.needs_grad(x, *ws))
        return self.norm(x) if self.use_norm else x

    def forward_res(self, x):
        """-> (self(x), x).  For an input that also feeds a skip connection: hand the second result to the skip's
        consumer and its gradient is folded into this block's data-gradient kernel (ops.DownResFn).  CIDNet.forward uses
        it for the four skip connections that leave a down block's input (cidnet.CIDNet._down_skip)."""
        y, x = ops.DownResFn.apply(x, self.down[0].weight, self.prelu.weight)
        return (self.norm(y) if self.use_norm else y), x


class NormUpsample(nn.Module):
    """Reference: net/transformer_utils.py:50-70."""

    def __init__(self, in_ch, out_ch, scale=2, use_norm=False):
        super().__init__()
        if scale != 2:
            raise NotImplementedError("NormUpsample: only scale=2 (the CIDNet configuration) is implemented")
        self.use_norm = use_norm
        if self.use_norm:
            self.norm = LayerNorm(out_ch)
        self.prelu = nn.PReLU()
        self.up_scale = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=False),
            nn.UpsamplingBilinear2d(scale_factor=scale))
        self.up = nn.Conv2d(out_ch * 2, out_ch, kernel_size=1, stride=1, padding=0, bias=False)

    def forward(self, x, y):
        ws = (self.up_scale[0].weight, self.up.weight, self.prelu.weight)
        x = ops.UpFn.apply(x, y, *ws, ops.needs_grad(x, y, *ws))
        return self.norm(x) if self.use_norm else x
