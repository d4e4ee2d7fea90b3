"""hvi-cidnet_amd: MI355X-native CIDNet forward/backward hot path.

Hand-written gfx950 HIP kernels behind a C ABI (include/cidnet_hip.h, csrc/), wrapped as
torch.autograd.Function ops (ops.py) and assembled into modules whose names, constructor arguments,
attributes and state_dict keys equal the reference's net/CIDNet.py (cidnet.py, lca.py,
transformer_utils.py, hvi_transform.py).  There is no CPU or ATen fallback: tensors must live on a
ROCm device and libcidnet_hip.so must be built (`python hvi-cidnet_amd/build.py`).
"""
from . import _lib  # noqa: F401
from .cidnet import CIDNet
from .cidnet_mssa import CIDNet as CIDNet_MSSA, SpatialAttention
from .cidnet_tnsm import CIDNet_TNSM
from .tnsm import HV_TNSM, I_TNSM, TrainableNoiseSuppression
from .hvi_transform import RGB_HVI
from .lca import CAB, IEL, HV_LCA, I_LCA
from .transformer_utils import LayerNorm, NormDownsample, NormUpsample
from .losses import L1Loss, SSIM, EdgeLoss, CIDNetLoss, PerceptualLoss, VGGFeatureExtractor, tnsm_noise_loss
from .inference import enhance, load_weights, save_pretrained, pad_to_multiple
from .schedule import WarmupCosineLR
from .ops import set_storage_dtype, set_precision, set_math_levels

__all__ = ["CIDNet", "CIDNet_MSSA", "SpatialAttention", "CIDNet_TNSM", "HV_TNSM", "I_TNSM", "TrainableNoiseSuppression", "RGB_HVI", "CAB", "IEL", "HV_LCA", "I_LCA", "LayerNorm", "NormDownsample", "NormUpsample", "L1Loss", "SSIM", "EdgeLoss", "CIDNetLoss", "tnsm_noise_loss", "PerceptualLoss", "VGGFeatureExtractor", "enhance", "load_weights", "save_pretrained", "pad_to_multiple",
           "WarmupCosineLR", "set_storage_dtype", "set_precision", "set_math_levels"]
