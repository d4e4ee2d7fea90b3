"""`CIDNet`: drop-in for the reference's net/CIDNet.py module (same constructor, attribute `trans`,
method `HVIT`, 191 state_dict keys) running entirely on the hand-written gfx950 kernels."""
import weakref

import torch
import torch.nn as nn

from . import ops
from .hvi_transform import RGB_HVI
from .lca import HV_LCA, I_LCA
from .transformer_utils import NormDownsample, NormUpsample

def _allow_cross_stream_param_grads():
    """With `dual_norms` a LayerNorm module's weight / bias receive one gradient contribution from the OTHER branch's stream
    (the partner block's y-norm is computed next to this tensor's x-norm): intentional, and autograd synchronises the
    streams for it -- only its "AccumulateGrad node's stream does not match" warning is switched off, and only once a
    model actually runs with dual_norms (not at import time)."""
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


try:  # the reference mixes this in for from_pretrained/save_pretrained (net/CIDNet.py:6,8)
    from huggingface_hub import PyTorchModelHubMixin as _HubMixin
except Exception:  # pragma: no cover - hub client not installed
    class _HubMixin:  # type: ignore
        pass


# Per-module runtime objects (side stream, back-pressure events) live OUTSIDE the module: HIP streams and events cannot be
# pickled or deep-copied, and a model must stay copy.deepcopy()-able (EMA copies) and torch.save()-able after its first
# forward.  Keyed weakly by the module; a copy starts with fresh state on whatever device it runs on.
_RUNTIME = weakref.WeakKeyDictionary()


def _runtime(module, device):
    st = _RUNTIME.get(module)
    if st is None or st["device"] != device:
        st = {"device": device, "side": None, "events": []}
        _RUNTIME[module] = st
    return st


class _RepConv(nn.Sequential):
    """ReplicationPad2d(1) + Conv2d(3x3, padding=0) container (keys `<name>.1.weight`) whose forward
    is the fused replicate-border conv kernel."""

    def __init__(self, cin, cout):
        super().__init__(nn.ReplicationPad2d(1), nn.Conv2d(cin, cout, 3, stride=1, padding=0, bias=False))

    def forward(self, x):
        return ops.RepConv3x3Fn.apply(x, self[1].weight)


class CIDNet(nn.Module, _HubMixin):
    """Reference: net/CIDNet.py:8-126."""

    def __init__(self, channels=[36, 36, 72, 144], heads=[1, 2, 4, 8], norm=False):
        super().__init__()
        [ch1, ch2, ch3, ch4] = channels
        [head1, head2, head3, head4] = heads

        # HV_ways
        self.HVE_block0 = _RepConv(3, ch1)
        self.HVE_block1 = NormDownsample(ch1, ch2, use_norm=norm)
        self.HVE_block2 = NormDownsample(ch2, ch3, use_norm=norm)
        self.HVE_block3 = NormDownsample(ch3, ch4, use_norm=norm)

        self.HVD_block3 = NormUpsample(ch4, ch3, use_norm=norm)
        self.HVD_block2 = NormUpsample(ch3, ch2, use_norm=norm)
        self.HVD_block1 = NormUpsample(ch2, ch1, use_norm=norm)
        self.HVD_block0 = _RepConv(ch1, 2)

        # I_ways
        self.IE_block0 = _RepConv(1, ch1)
        self.IE_block1 = NormDownsample(ch1, ch2, use_norm=norm)
        self.IE_block2 = NormDownsample(ch2, ch3, use_norm=norm)
        self.IE_block3 = NormDownsample(ch3, ch4, use_norm=norm)

        self.ID_block3 = NormUpsample(ch4, ch3, use_norm=norm)
        self.ID_block2 = NormUpsample(ch3, ch2, use_norm=norm)
        self.ID_block1 = NormUpsample(ch2, ch1, use_norm=norm)
        self.ID_block0 = _RepConv(ch1, 1)

        self.HV_LCA1 = HV_LCA(ch2, head2)
        self.HV_LCA2 = HV_LCA(ch3, head3)
        self.HV_LCA3 = HV_LCA(ch4, head4)
        self.HV_LCA4 = HV_LCA(ch4, head4)
        self.HV_LCA5 = HV_LCA(ch3, head3)
        self.HV_LCA6 = HV_LCA(ch2, head2)

        self.I_LCA1 = I_LCA(ch2, head2)
        self.I_LCA2 = I_LCA(ch3, head3)
        self.I_LCA3 = I_LCA(ch4, head4)
        self.I_LCA4 = I_LCA(ch4, head4)
        self.I_LCA5 = I_LCA(ch3, head3)     # parameters kept for checkpoint parity; see forward()
        self.I_LCA6 = I_LCA(ch2, head2)

        self.trans = RGB_HVI()

    # hooks the MSSA variant overrides (net/CIDNet_MSSA.py); identity / dead-block skip for the base net
    def _gate(self, name, t):
        return t

    def _stage5(self, i_dec3, hv_3):
        """-> (input of ID_block2, HV_LCA5(hv_3, i_dec3)).  net/CIDNet.py:105 evaluates I_LCA5(i_dec3, hv_3) and :109 then
        ignores it (ID_block2 is fed i_dec3): the result never reaches the output and its 13 parameters get no gradient, so
        the dead block is not executed here.  i_dec3 has two consumers, HV_LCA5's y-norm and ID_block2; it is threaded
        through the norm (ops.LayerNormResFn hands it on as a view) so that ID_block2's gradient arrives in the LayerNorm
        backward kernel instead of being summed by a separate autograd accumulation pass."""
        blk = self.HV_LCA5
        if not (self.chain_lca_inputs and torch.is_grad_enabled() and i_dec3.requires_grad):
            return self._par(lambda: i_dec3, lambda: blk(hv_3, i_dec3), (i_dec3, hv_3))

        def hv_side():
            hv_n, hv_r = blk.norm.forward_res(hv_3)
            i_n, i_pass = blk.norm.forward_res(i_dec3)
            return blk.body(hv_n, i_n, hv_r), i_pass
        _, (out_hv, i_pass) = self._par(lambda: None, hv_side, (i_dec3, hv_3))
        return i_pass, out_hv

    # A down block whose input also feeds a skip connection (net/CIDNet.py:80-81,85-86): with `fold_skip_grads` the skip's
    # gradient is added in the epilogue of the block's data-gradient conv (ops.DownResFn) instead of by a separate autograd
    # accumulation pass over the tensor (four per step: two at 400x600, two at 200x300).  Round 1 switched this off after
    # multi-second host stalls; their cause was the unbounded run-ahead of the host (dp.DataParallelTrainer), not this.
    fold_skip_grads = True

    def _down_skip(self, block, t):
        """-> (tensor for the skip connection, block(t))"""
        if self.fold_skip_grads and torch.is_grad_enabled() and t.requires_grad:
            y, t_skip = block.forward_res(t)
            return t_skip, y
        return t, block(t)

    # ---- two-stream execution: the I branch and the HV branch of every stage are independent ----
    two_streams = True

    def _two(self, t):
        # (Round 2 serialised the branches while a bf16-MFMA kernel was enabled: beside LDS-fed bf16 MFMAs, packed-fp32
        # instructions with op_sel in another kernel's waves were seen to drop products.  Since round 3 the library is built
        # without packed-fp32 and SDWA instructions -- hvi-cidnet_amd/build.py, checked by tests/test_abi.py -- and the
        # split-product 3x3 conv is on by default, on both streams.)
        return self.two_streams and t.is_cuda

    def _side(self, device):
        st = _runtime(self, device)
        if st["side"] is None:
            st["side"] = torch.cuda.Stream(device=device)
        return st["side"]

    @property
    def _side_stream(self):
        """the HV branch's stream once a two-stream forward has run on the current device (None before)"""
        st = _RUNTIME.get(self)
        return st["side"] if st is not None else None

    def _par(self, f_i, f_hv, shared):
        """Run f_i on the current stream and f_hv on a side stream, then join.  `shared` = tensors read by
        both (allocator bookkeeping for cross-stream use)."""
        if not self._two(shared[0]):
            return f_i(), f_hv()
        main = torch.cuda.current_stream()
        side = self._side(shared[0].device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out_hv = f_hv()
        out_i = f_i()
        main.wait_stream(side)
        for t in shared:
            t.record_stream(side)
        for t in (out_hv if isinstance(out_hv, (tuple, list)) else (out_hv,)):
            if isinstance(t, torch.Tensor):
                t.record_stream(main)
        return out_i, out_hv

    # An LCA pair (net/CIDNet.py:83-84 etc.): I_LCA(i, hv) and HV_LCA(hv, i).  Each of the two inputs feeds three consumers --
    # the x-norm of its own block (which also hands it on as the CAB residual) and the y-norm of the other block -- and
    # autograd would sum their gradients with separate passes over the tensor (ten per step).  With `chain_lca_inputs` the
    # input is threaded through the norms instead: t -> HV.norm -> (n, t1) -> I.norm -> (n', t2) -> residual consumer, every
    # link being ops.LayerNormResFn whose backward kernel adds the gradient that arrives for its pass-through output.  Each
    # module's norm still runs on its own branch's stream (the in-place summing of LayerNorm parameter gradients relies on
    # that), and the forward needs no extra synchronisation: t1 / t2 are views of t.
    chain_lca_inputs = True
    # the two norms that read the same LCA input as ONE pass (ops.LayerNormDualFn): x read once in the forward, once in the backward
    # Off by default since round 4: alternating runs on one box showed no step gain (313.0 / 311.2 off, 313.4 / 312.8 on),
    # and module b's gradients go through autograd (27 ATen adds and 20 copies per step on the other branch's stream).
    dual_norms = False

    def _lca_pair(self, I_blk, HV_blk, i, hv):
        if not (self.chain_lca_inputs and torch.is_grad_enabled() and (i.requires_grad or hv.requires_grad)):
            return self._par(lambda: I_blk(i, hv), lambda: HV_blk(hv, i), (i, hv))
        two = self._two(i)
        main = side = None
        if two:
            main = torch.cuda.current_stream()
            side = self._side(i.device)
            side.wait_stream(main)

        def on_side(f):
            if not two:
                return f()
            with torch.cuda.stream(side):
                return f()
        if self.dual_norms and ops.ln_dual_supported(i) and ops.ln_dual_supported(hv):
            _allow_cross_stream_param_grads()
            # one pass per input: its own block's x-norm (+ residual hand-over) and the partner block's y-norm of it
            hv_nhv, hv_ni, hv2 = on_side(lambda: HV_blk.norm.forward_dual(hv, I_blk.norm))
            i_ni, i_nhv, i2 = I_blk.norm.forward_dual(i, HV_blk.norm)
            if two:                              # each branch now consumes a tensor the other branch's stream produced
                main.wait_stream(side)
                side.wait_stream(main)
                hv_ni.record_stream(main)
                i_nhv.record_stream(side)
        else:
            (hv_nhv, hv1), (i_nhv, i1) = on_side(lambda: (HV_blk.norm.forward_res(hv), HV_blk.norm.forward_res(i)))
            i_ni, i2 = I_blk.norm.forward_res(i1)
            hv_ni, hv2 = I_blk.norm.forward_res(hv1)
        out_hv = on_side(lambda: HV_blk.body(hv_nhv, i_nhv, hv2))
        out_i = I_blk.body(i_ni, hv_ni, i2)
        if two:
            main.wait_stream(side)
            for t in (i, hv):
                t.record_stream(side)
            out_hv.record_stream(main)
        return out_i, out_hv

    # Back-pressure for callers that drive model(x) / backward() themselves (the trainer has its own): the host enqueues a
    # forward+backward about twice as fast as the GPU runs it, and every queued pass pins several GiB in the caching
    # allocator (cross-stream frees wait for their events), which ends in hipMalloc storms and multi-second stalls once
    # the host is far ahead (dp.DataParallelTrainer, tools/stall_probe.py).  At most `max_queued_forwards` passes may be
    # in flight; 0 disables the wait.
    max_queued_forwards = 4
    # ... and at most this fraction of the device's memory may be pinned for queued passes (bytes the allocator holds as
    # "active" beyond what the host still references): a 32 x 3 x 1024 x 1024 inference forward cycles ~40 GB, and four of them
    # in flight made the allocator hipMalloc a second and a third footprint (302 instead of 117 ms per batch, round 4)
    max_queued_fraction = 0.25

    def _backpressure(self, x):
        if not (x.is_cuda and self.max_queued_forwards > 0) or torch.cuda.is_current_stream_capturing():
            return
        st = _runtime(self, x.device)
        q = st["events"]
        while len(q) >= self.max_queued_forwards:
            q.pop(0).synchronize()
        if len(q) > 1 and self.max_queued_fraction > 0:
            cap = st.get("cap")
            if cap is None:
                cap = st["cap"] = int(self.max_queued_fraction * torch.cuda.get_device_properties(x.device).total_memory)
            while len(q) > 1:
                ms = torch.cuda.memory_stats_as_nested_dict(x.device)
                if ms["active_bytes"]["all"]["current"] - ms["allocated_bytes"]["all"]["current"] <= cap:
                    break
                q.pop(0).synchronize()
        ev = torch.cuda.Event()
        ev.record()
        q.append(ev)

    def forward(self, x):
        self._backpressure(x)
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise RuntimeError(f"CIDNet: H and W must be multiples of 8 (got {tuple(x.shape[2:])}); the reference "
                               "fails in NormUpsample's cat for other sizes (net/transformer_utils.py:64)")
        hvi = self.trans.HVIT(x)
        if torch.is_grad_enabled() and hvi.requires_grad:
            # three consumers (HV stem, I stem on plane 2, output residual): one kernel sums their gradients
            hvi_hv, i, hvi = ops.HviFanoutFn.apply(hvi)
        else:
            hvi_hv, i = hvi, hvi[:, 2:3, :, :]                     # the I stem reads the plane in place (no copy)
        # low
        (i_enc0, i_enc1), (hv_0, hv_1) = self._par(
            lambda: self._down_skip(self.IE_block1, self.IE_block0(i)),
            lambda: self._down_skip(self.HVE_block1, self.HVE_block0(hvi_hv)), (hvi,))
        i_jump0 = i_enc0
        hv_jump0 = hv_0

        i_enc2, hv_2 = self._lca_pair(self.I_LCA1, self.HV_LCA1, i_enc1, hv_1)
        (v_jump1, i_enc2), (hv_jump1, hv_2) = self._par(lambda: self._down_skip(self.IE_block2, i_enc2),
                                                        lambda: self._down_skip(self.HVE_block2, hv_2), (i_enc2, hv_2))

        # reference quirk: level-3 encoders take the PRE-LCA2 tensors (net/CIDNet.py:94-95).  They run first here so that
        # the LCA pair's gradient for those tensors enters the down blocks' data-gradient conv as an epilogue addend
        # (ops.DownResFn) instead of a separate autograd accumulation pass; the values are the reference's either way.
        (i_enc2, i_enc3), (hv_2, hv_3) = self._par(lambda: self._down_skip(self.IE_block3, i_enc2),
                                                   lambda: self._down_skip(self.HVE_block3, hv_2), (i_enc2, hv_2))
        v_jump2, hv_jump2 = self._lca_pair(self.I_LCA2, self.HV_LCA2, i_enc2, hv_2)

        i_enc4, hv_4 = self._lca_pair(self.I_LCA3, self.HV_LCA3, i_enc3, hv_3)
        i_dec4, hv_4b = self._lca_pair(self.I_LCA4, self.HV_LCA4, i_enc4, hv_4)

        i_dec3, hv_3 = self._par(lambda: self._gate("sa_i3", self.ID_block3(i_dec4, v_jump2)),
                                 lambda: self._gate("sa_hv3", self.HVD_block3(hv_4b, hv_jump2)),
                                 (i_dec4, hv_4b, v_jump2, hv_jump2))
        i_dec2, hv_2 = self._stage5(i_dec3, hv_3)

        i_dec2, hv_2 = self._par(lambda: self._gate("sa_i2", self.ID_block2(i_dec2, v_jump1)),
                                 lambda: self._gate("sa_hv2", self.HVD_block2(hv_2, hv_jump1)),
                                 (i_dec2, hv_2, v_jump1, hv_jump1))

        i_dec1, hv_1 = self._lca_pair(self.I_LCA6, self.HV_LCA6, i_dec2, hv_2)

        i_dec0, hv_0 = self._par(
            lambda: self.ID_block0(self._gate("sa_i1", self.ID_block1(i_dec1, i_jump0))),
            lambda: self.HVD_block0(self._gate("sa_hv1", self.HVD_block1(hv_1, hv_jump0))),
            (i_dec1, hv_1, i_jump0, hv_jump0))

        # cat([hv_0, i_dec0], 1) + hvi -> PHVIT, fused (net/CIDNet.py:119-120)
        return self.trans.PHVIT_residual(hv_0, i_dec0, hvi)

    def HVIT(self, x):
        return self.trans.HVIT(x)
