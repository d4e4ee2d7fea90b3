"""Drop-ins for the reference's net/LCA.py (CAB, IEL, HV_LCA, I_LCA): identical parameter names and
shapes (nn.Conv2d modules are kept as parameter containers), forward on the HIP ops."""
import torch
import torch.nn as nn

from . import ops
from .transformer_utils import LayerNorm


class CAB(nn.Module):
    """Cross-attention block.  Reference: net/LCA.py:7-41."""

    def __init__(self, dim, num_heads, bias):
        super().__init__()
        if bias:
            raise NotImplementedError("CAB: the CIDNet hot path is bias-free (net/LCA.py:73,87)")
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.q = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)
        self.q_dwconv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim, bias=bias)
        self.kv = nn.Conv2d(dim, dim * 2, kernel_size=1, bias=bias)
        self.kv_dwconv = nn.Conv2d(dim * 2, dim * 2, kernel_size=3, stride=1, padding=1, groups=dim * 2, bias=bias)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=bias)

    def forward(self, x, y, residual=None):
        """CAB(x, y) [+ residual].  With residual=None a zero tensor stands in for it."""
        if residual is None:
            residual = torch.zeros_like(x)
        return ops.CABResidualFn.apply(residual, x, y, self.temperature, self.q.weight, self.q_dwconv.weight,
                                       self.kv.weight, self.kv_dwconv.weight, self.project_out.weight, self.num_heads)


class IEL(nn.Module):
    """Gated feed-forward.  Reference: net/LCA.py:45-67."""

    def __init__(self, dim, ffn_expansion_factor=2.66, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("IEL: the CIDNet hot path is bias-free")
        hidden_features = int(dim * ffn_expansion_factor)
        self.project_in = nn.Conv2d(dim, hidden_features * 2, kernel_size=1, bias=bias)
        self.dwconv = nn.Conv2d(hidden_features * 2, hidden_features * 2, kernel_size=3, stride=1, padding=1,
                                groups=hidden_features * 2, bias=bias)
        self.dwconv1 = nn.Conv2d(hidden_features, hidden_features, kernel_size=3, stride=1, padding=1,
                                 groups=hidden_features, bias=bias)
        self.dwconv2 = nn.Conv2d(hidden_features, hidden_features, kernel_size=3, stride=1, padding=1,
                                 groups=hidden_features, bias=bias)
        self.project_out = nn.Conv2d(hidden_features, dim, kernel_size=1, bias=bias)
        self.Tanh = nn.Tanh()

    def forward(self, x, residual=None):
        ws = (self.project_in.weight, self.dwconv.weight, self.dwconv1.weight, self.dwconv2.weight, self.project_out.weight)
        return ops.IELFn.apply(x, residual, *ws, ops.needs_grad(x, residual, *ws))


class HV_LCA(nn.Module):
    """Reference: net/LCA.py:71-81 -- the IEL stage has NO residual."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.gdfn = IEL(dim)
        self.norm = LayerNorm(dim)
        self.norm.lca_internal = True
        self.ffn = CAB(dim, num_heads, bias)

    def forward(self, x, y):
        xn, xr = self.norm.forward_res(x)          # xr is x: the residual's gradient is added inside the LN backward
        return self.body(xn, self.norm(y), xr)

    def body(self, xn, yn, xr):
        """the block after its two input norms: xn = norm(x), yn = norm(y), xr = x as the residual input"""
        x = self.ffn(xn, yn, residual=xr)
        return self.gdfn(self.norm(x))


class I_LCA(nn.Module):
    """Reference: net/LCA.py:83-93."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.norm.lca_internal = True
        self.gdfn = IEL(dim)
        self.ffn = CAB(dim, num_heads, bias=bias)

    def forward(self, x, y):
        xn, xr = self.norm.forward_res(x)          # xr is x: the residual's gradient is added inside the LN backward
        return self.body(xn, self.norm(y), xr)

    def body(self, xn, yn, xr):
        """the block after its two input norms: xn = norm(x), yn = norm(y), xr = x as the residual input"""
        x = self.ffn(xn, yn, residual=xr)
        xn, xr = self.norm.forward_res(x)
        return self.gdfn(xn, residual=xr)
