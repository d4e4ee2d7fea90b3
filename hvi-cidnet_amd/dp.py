"""Data-parallel training step for CIDNet: one process per GPU, RCCL (torch.distributed backend
"nccl") over xGMI, gradients all-reduced in a few flat buckets that overlap with the backward.

The reference trains on a single GPU (train.py:34 pins CUDA_VISIBLE_DEVICES='0') and has no
distributed code; this harness restates its step semantics (train.py:56-73: forward, loss,
zero_grad, backward, Adam.step) for N ranks:

  * parameters live in ONE flat fp32 buffer (the modules' Parameters are views into it); weight
    gradients are written by the HIP backward kernels straight into the matching flat gradient
    buffer (ops.set_grad_arena), so there is no per-tensor packing before communication;
  * the flat gradient buffer is cut into `n_buckets` contiguous buckets.  Parameters are laid out
    in the order in which their gradients become ready (learned from one probing backward), so
    bucket b is complete while the backward of earlier layers is still running; its all-reduce is
    launched from a post-accumulate-grad hook and runs on RCCL's stream concurrently;
  * the message is small (7.9 MB for CIDNet) and xGMI is point-to-point, so the collective is
    latency-bound: few large buckets, never one collective per tensor;
  * after the last bucket lands, one fused Adam kernel updates the flat parameter buffer
    (1/world_size folded into the gradient scale).

Parameters that receive no gradient (the reference's dead I_LCA5.* block, SURVEY.md quick fact 1)
are detected in the probing backward and left out of buckets and optimizer, as torch.optim does
for grad=None parameters.  Nothing here is CIDNet-specific: any nn.Module works, which is what the
gloo/CPU tests use.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import os

import torch
import torch.distributed as dist


def _default_loss(out, gt):
    from . import ops
    return ops.L1LossFn.apply(out, gt)


class FlatAdam:
    """torch.optim.Adam semantics on one flat buffer; `kernel` selects the fused HIP update."""

    def __init__(self, flat_p, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, kernel: bool = True):
        self.p = flat_p
        self.m = torch.zeros_like(flat_p)
        self.v = torch.zeros_like(flat_p)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        self.kernel = kernel

    def step(self, flat_g, n_live: int, grad_scale: float = 1.0):
        self.t += 1
        b1, b2 = self.betas
        p, g, m, v = self.p[:n_live], flat_g[:n_live], self.m[:n_live], self.v[:n_live]
        if self.kernel:
            from . import ops
            ops.adam_step(p, g, m, v, self.lr, b1, b2, self.eps, self.wd, self.t, grad_scale)
            return
        # plain-torch restatement of the same update (used by the CPU/gloo tests of the harness only)
        g = g * grad_scale
        if self.wd:
            g = g + self.wd * p
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        p.sub_((self.lr / bc1) * (m / (v.sqrt() / bc2 ** 0.5 + self.eps)))


def rank_cpu_set(cpus, local_rank: int, local_world: int):
    """The contiguous share of `cpus` (sorted ids) that local rank `local_rank` of `local_world` ranks on this node gets:
    equal chunks, the remainder to the first ranks; every rank at least one CPU (ranks then share when CPUs are fewer)."""
    cpus = sorted(cpus)
    n = len(cpus)
    if local_world <= 1 or n == 0:
        return list(cpus)
    if n < local_world:
        return [cpus[local_rank % n]]
    base, rem = divmod(n, local_world)
    lo = local_rank * base + min(local_rank, rem)
    return cpus[lo:lo + base + (1 if local_rank < rem else 0)]


def pin_rank_to_cpus(local_rank: Optional[int] = None, local_world: Optional[int] = None) -> Optional[List[int]]:
    """One process per GPU: give every local rank its own slice of the CPUs this job may use (sched_setaffinity) and size
    torch's intra-op pool to it.  Without it the N enqueue threads (13 ms of Python per 26 ms step each) and their helper
    threads migrate over all cores and evict each other.  LOCAL_RANK / LOCAL_WORLD_SIZE from the launcher are the
    defaults; CIDNET_CPU_AFFINITY=0 disables.  Returns the CPU list that was set (None when nothing was done)."""
    if os.environ.get("CIDNET_CPU_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    lr = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else local_world
    if lw <= 1:
        return None
    mine = rank_cpu_set(os.sched_getaffinity(0), lr, lw)
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(len(mine), 8)))
    return mine


class DataParallelTrainer:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, n_buckets: int = 4, loss_fn: Optional[Callable] = None,
                 process_group=None, use_hip_kernels: bool = True, wgrad_stream: bool = True, use_graph: bool = False,
                 max_steps_in_flight: int = 3, max_queued_bytes: Optional[int] = None,
                 prepared_weights: Optional[bool] = None):
        self.model = model
        self.prepared_weights = (os.environ.get("CIDNET_PREPARED_WEIGHTS", "1") == "1") if prepared_weights is None else bool(prepared_weights)
        # Back-pressure.  Nothing in a training step synchronises host and device, and the host enqueues a step in ~13 ms
        # while the GPU needs ~31 ms.  Root cause of the multi-second stalls of round 1 (tools/stall_probe.py,
        # profiles/r02_stall_probe_*.txt): every tensor the host frees after it was used on a side stream
        # (record_stream) stays "active" in the caching allocator until the GPU reaches that point, so nothing of a
        # still-queued step can be reused -- each queued step pins ~6 GiB, and the allocator hipMallocs that much in NEW
        # segments per step the host is ahead (reserved memory 30 -> 130 GiB over 24 unthrottled steps, 20-40 device
        # mallocs per step, single hipMalloc calls of 60 ms once the queue is deep; at ~45 steps ahead the 288 GB are
        # gone and the allocator falls into its synchronise-free-retry path).  Bounding the run-ahead removes the cause;
        # the bound is in steps AND in bytes the allocator holds for queued work, so a configuration with more launches
        # or larger activations per step (TNSM, the full objective, bigger batches) is covered as well.
        # Three steps, not two: a queue of two steps (62 ms of work) is drained by an ordinary 65 ms host hiccup.
        self.max_steps_in_flight = max(1, int(max_steps_in_flight))
        # None: a third of the device's memory, resolved at the first step (96 GiB on a 288 GB MI355X)
        self.max_queued_bytes = None if max_queued_bytes is None else int(max_queued_bytes)
        self._step_events = []
        self.loss_fn = loss_fn or _default_loss
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # rehearsal switch: run the bucket all-reduces even on a 1-rank group (exercises the RCCL path on one GPU)
        self._force_comm = dist.is_initialized() and os.environ.get("CIDNET_DP_FORCE_ALLREDUCE") == "1"
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.n_buckets = max(1, n_buckets)
        self.use_hip = use_hip_kernels
        # weight-gradient GEMMs on a third stream (they feed nothing but the optimizer): 35.9 -> 34.4 ms/step at
        # bs=8 400x600 -- the data-gradient chain of short, latency-bound launches no longer waits behind them
        self.wgrad_stream = wgrad_stream
        self._opt_args = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._ready = False
        self._handles: List = []
        self._launch_stream = None
        self._main_stream = None
        # replay forward + loss + backward as ONE hipGraph (captured over all three streams) instead of ~700 launches
        # issued from Python: at bs=8 about 5 ms of every 34 ms step are launch-bound phases (the 50x75 level)
        self.use_graph = use_graph
        self._graph = None
        self._capturing = False
        self.params = [p for p in model.parameters() if p.requires_grad]

    # ---- one-time setup: probe gradient order, flatten, hook -----------------------------------
    def _setup(self, x, gt):
        # 1) probing backward: which parameters get gradients, in which order, and how often
        order, fired = [], {}
        hooks = []
        for p in self.params:
            def h(param, _p=p):
                if id(_p) not in fired:
                    fired[id(_p)] = 0
                    order.append(_p)
                fired[id(_p)] += 1
            hooks.append(p.register_post_accumulate_grad_hook(h))
        ln_uses = [m._use for m in self.model.modules() if hasattr(m, "_use")] if self.use_hip else []
        if self.use_hip:
            from . import ops
            ops.set_grad_arena(None, None)
            ops.start_grad_probe()
            for st in ln_uses:
                st.acc, st.fwd, st.bwd, st.probe = False, 0, 0, [0, 0]
        self.model.zero_grad(set_to_none=True)
        loss = self._loss(self.model(x), gt, x)
        loss.backward()
        for h in hooks:
            h.remove()
        # a parameter whose gradient is produced by several kernel launches per step (LayerNorm affine:
        # 3 uses per LCA) must be summed by autograd, not written in place
        counts = ops.stop_grad_probe() if self.use_hip else {}
        multi_ids = {id(p) for p in self.params if counts.get(p.data_ptr(), 0) > 1}
        # LayerNorm modules applied several times per step whose every forward use came back in the probing backward: their
        # uses sum the weight / bias gradients in place in the arena (ops.LNUse), so they count as arena parameters
        ln_acc_ids = set()
        for m in (self.model.modules() if self.use_hip else ()):
            st = getattr(m, "_use", None)
            if st is not None and st.probe[0] > 0 and st.probe[0] == st.probe[1]:
                st.acc, st.fwd, st.bwd = True, 0, 0
                ln_acc_ids.update((id(m.weight), id(m.bias)))
        multi_ids -= ln_acc_ids
        # single-use parameters whose gradient a kernel writes in place into the arena, possibly on the weight-gradient stream
        self._arena_ids = {id(p) for p in self.params if counts.get(p.data_ptr(), 0) == 1} | ln_acc_ids
        self.model.zero_grad(set_to_none=True)
        live = order
        dead = [p for p in self.params if id(p) not in fired]
        # 2) flat buffers in gradient-ready order; dead parameters parked at the tail
        layout = live + dead
        n_live = sum(p.numel() for p in live)
        n_all = sum(p.numel() for p in layout)
        dev = layout[0].device
        flat_p = torch.empty(n_all, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(n_all, device=dev, dtype=torch.float32)
        off = 0
        self._slices = {}
        for p in layout:
            n = p.numel()
            flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + n].view(p.shape)
            self._slices[id(p)] = (off, n)
            off += n
        self.flat_p, self.n_live = flat_p, n_live
        # identical initial weights on every rank
        if self.world > 1 or self._force_comm:
            dist.broadcast(self.flat_p, src=0, group=self.pg)
        # 3) buckets: contiguous, roughly equal, cut at parameter boundaries
        target = (n_live + self.n_buckets - 1) // self.n_buckets
        self.buckets, cur_start, cur_n, members = [], 0, 0, []
        for p in live:
            members.append(p)
            cur_n += p.numel()
            if cur_n >= target and len(self.buckets) < self.n_buckets - 1:
                self.buckets.append((cur_start, cur_n, members))
                cur_start += cur_n
                cur_n, members = 0, []
        if members:
            self.buckets.append((cur_start, cur_n, members))
        self._bucket_of = {}
        for bi, (_, _, mem) in enumerate(self.buckets):
            for p in mem:
                self._bucket_of[id(p)] = bi
        self._pending = [0] * len(self.buckets)
        self._ln_uses = [st for st in ln_uses if st.acc]
        self._check_layout_across_ranks(layout, n_live)
        # 4) gradients written in place by the HIP backward kernels (single-use parameters only)
        multi = [p.data_ptr() for p in live if id(p) in multi_ids]      # pointers AFTER re-homing into flat_p
        if self.use_hip:
            from . import ops
            ops.set_grad_arena(self.flat_p, self.flat_g, exclude_ptrs=multi)
            ops.enable_wgrad_stream(self.wgrad_stream)
            # this trainer's fused Adam is the only writer of flat_p: inside its passes (_pass_scope) prepared (split,
            # fragment-ordered) weight operands are kept across launches, and re-prepared in two launches after every update
            # (ops.refresh_prepared_weights); anything else that writes the weights without torch's version counter seeing
            # it (p.data.copy_, raw kernels) must call weights_changed()
            ops.clear_prepared_weights()
        for p in live:
            p.register_post_accumulate_grad_hook(self._on_grad)
        self.opt = FlatAdam(self.flat_p, kernel=self.use_hip, **self._opt_args)
        self._ready = True

    def _check_layout_across_ranks(self, layout, n_live):
        """Every rank derives the arena layout from its OWN probing backward (gradient-ready order).  The order is a
        property of the graph, so it is the same everywhere -- but if it ever were not (a data-dependent branch, another
        library version on one node), the all-reduce would silently add different parameters into each other.  Rank 0
        broadcasts a digest of (parameter name, offset, size) in layout order plus the bucket cuts; a mismatch raises."""
        if not (self.world > 1 or self._force_comm):
            return
        import hashlib
        names = {id(p): n for n, p in self.model.named_parameters()}
        h = hashlib.sha256()
        for p in layout:
            off, n = self._slices[id(p)]
            h.update(f"{names.get(id(p), '?')}:{off}:{n};".encode())
        h.update(f"live={n_live};buckets={[(s0, c) for s0, c, _ in self.buckets]}".encode())
        mine = torch.tensor(list(h.digest()), dtype=torch.int64, device=self.flat_g.device)
        ref = mine.clone()
        dist.broadcast(ref, src=0, group=self.pg)
        same = torch.equal(mine.cpu(), ref.cpu())
        flag = torch.tensor([0 if same else 1], dtype=torch.int64, device=self.flat_g.device)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.pg)
        if int(flag.item()) != 0:
            raise RuntimeError(f"rank {self.rank}: gradient-arena layout differs between ranks "
                               f"({'this rank differs from rank 0' if not same else 'another rank differs'}); "
                               "the bucket all-reduce would mix parameters")

    def _begin_pass(self, x):
        """state every forward+backward starts from: all gradients of every bucket pending, no collective in flight, and
        the in-place LayerNorm gradient counters at zero -- a stray grad-mode forward without a backward (validation
        without no_grad, model(x) for logging, an exception between forward and backward) would otherwise leave
        `fwd` ahead of `bwd` for good and the LayerNorm gradients would never be handed over again."""
        for bi, (_, _, mem) in enumerate(self.buckets):
            self._pending[bi] = len(mem)
        self._handles = []
        self._main_stream = torch.cuda.current_stream() if x.is_cuda else None
        for st in self._ln_uses:
            st.fwd = st.bwd = 0

    def _end_pass(self):
        """every live parameter's gradient must have arrived in the arena (and its bucket been launched)"""
        if any(self._pending):
            missing = [i for i, n in enumerate(self._pending) if n]
            raise RuntimeError(f"backward left gradients pending in buckets {missing}: a parameter that received a gradient "
                               "in the probing step got none now (graph changed?), or a LayerNorm's uses were not all "
                               "run backward")

    def _join_wgrad_stream(self):
        """make the current stream wait for the weight-gradient stream (ops._offload_wgrad)"""
        if self.use_hip:
            from . import ops
            torch.cuda.current_stream().wait_stream(ops.wgrad_stream(self.flat_g.device))

    def _on_grad(self, p):
        off, n = self._slices[id(p)]
        slot = self.flat_g[off:off + n]
        if p.grad.data_ptr() != slot.data_ptr():
            # multi-use parameters and gradients not produced through grad_like() (summed / allocated by autograd on the
            # branch streams) -- or a single-use arena gradient that autograd cloned instead of stealing: that one may
            # still be in flight on the weight-gradient stream, so the copy waits for it (never seen in practice;
            # without the wait it would read a stale gradient)
            if self.use_hip and self.wgrad_stream and p.grad.is_cuda and id(p) in self._arena_ids:
                self._join_wgrad_stream()
            slot.copy_(p.grad.reshape(-1))
        p.grad = None                               # the arena is the single home of gradients
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and (self.world > 1 or self._force_comm) and not self._capturing:
            self._launch_bucket(bi)

    def _launch_bucket(self, bi):
        """All-reduce one finished bucket without stalling the compute streams.  The bucket's gradients were written
        by kernels on up to three streams (the two branch streams and the weight-gradient stream) and this hook only
        runs in CPU order, so the collective is issued from a dedicated launch stream that first waits for the
        current tail of every producer stream; RCCL orders its own stream after the launch stream."""
        start, cnt, _ = self.buckets[bi]
        dev = self.flat_g.device
        if dev.type != "cuda":                    # gloo / CPU rehearsal: no streams
            self._handles.append(dist.all_reduce(self.flat_g[start:start + cnt], op=dist.ReduceOp.SUM, group=self.pg,
                                                 async_op=True))
            return
        if self._launch_stream is None:
            self._launch_stream = torch.cuda.Stream(device=dev)
        st = self._launch_stream
        producers = [torch.cuda.current_stream(), self._main_stream, getattr(self.model, "_side_stream", None)]
        if self.use_hip and self.wgrad_stream:
            from . import ops
            producers.append(ops.wgrad_stream(dev))
        for ps in producers:
            if ps is not None:
                st.wait_stream(ps)
        with torch.cuda.stream(st):
            self._handles.append(dist.all_reduce(self.flat_g[start:start + cnt], op=dist.ReduceOp.SUM, group=self.pg,
                                                 async_op=True))

    def set_lr(self, lr: float):
        """learning rate of the fused Adam from the next step on (see schedule.WarmupCosineLR)"""
        self._opt_args["lr"] = float(lr)
        if getattr(self, "opt", None) is not None:
            self.opt.lr = float(lr)

    # ---- hipGraph replay of forward + loss + backward ---------------------------------------------
    def _backward(self, loss):
        """loss.backward() with a cached unit gradient: autograd would otherwise launch an ATen fill for its implicit
        ones_like(loss) every step (the step launches only cidnet:: kernels, tests/test_trainer_gpu.py)"""
        one = getattr(self, "_unit_grad", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype or one.shape != loss.shape:
            one = self._unit_grad = torch.ones_like(loss)
        loss.backward(gradient=one)

    def _pass_scope(self):
        if self.use_hip and self.prepared_weights:
            from . import ops
            return ops.prepared_weights(True)
        import contextlib
        return contextlib.nullcontext()

    def _loss(self, out, gt, x):
        """loss_fn(model(x), gt); a loss function with a true `wants_input` attribute (losses.CIDNetLoss: the TNSM noise terms
        compare the output with the network input, train_tnsm.py:69) also gets the input as im1=x.  Tuple results of the
        model (CIDNet_TNSM in train mode) are handed to the loss function as they are."""
        if getattr(self.loss_fn, "wants_input", False):
            return self.loss_fn(out, gt, im1=x)
        return self.loss_fn(out, gt)

    def _fwd_bwd(self, x, gt):
        with self._pass_scope():
            self._begin_pass(x)
            loss = self._loss(self.model(x), gt, x)
            self._backward(loss)
            self._end_pass()
        return loss

    def _capture(self, x, gt):
        """Static input buffers, two eager passes on the capture stream (scratch buffers reach their final size), then
        capture.  The side stream of the model and the weight-gradient stream fork from / join the capture stream
        exactly as in eager mode, so the graph keeps the three-stream concurrency.  Collectives and Adam stay outside
        (Adam's bias correction is a host scalar that changes every step)."""
        self._gx, self._ggt = x.clone(), gt.clone()
        st = torch.cuda.Stream(device=x.device)
        st.wait_stream(torch.cuda.current_stream())
        self._capturing = True
        try:
            with torch.cuda.stream(st):
                for _ in range(2):
                    self._fwd_bwd(self._gx, self._ggt)
                    self._join_wgrad_stream()
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                loss = self._fwd_bwd(self._gx, self._ggt)
                self._join_wgrad_stream()
                self._gloss = loss.detach()
            self._graph = g
        finally:
            self._capturing = False

    def _graph_step(self, x, gt):
        if self._graph is None:
            self._capture(x, gt)
        self._gx.copy_(x)
        self._ggt.copy_(gt)
        self._graph.replay()
        if self.world > 1 or self._force_comm:
            dist.all_reduce(self.flat_g[:self.n_live], op=dist.ReduceOp.SUM, group=self.pg)
        self.opt.step(self.flat_g, self.n_live, grad_scale=1.0 / self.world)
        self._weights_changed()          # the captured pass reads prepared operands that the eager passes before it created
        return self._gloss

    # ---- the step ---------------------------------------------------------------------------------
    def step(self, x, gt):
        """forward, loss, backward (+ overlapped bucket all-reduce), fused Adam.  Returns the loss."""
        if not self._ready:
            self._setup(x, gt)
        if x.is_cuda:
            while len(self._step_events) >= self.max_steps_in_flight:
                self._step_events.pop(0).synchronize()
            # bytes pinned for queued work = "active" (in use or waiting for a stream event) minus what the host still holds;
            # only looked at while more than one step is queued (the stats call walks the allocator's counters)
            if self.max_queued_bytes is None:
                self.max_queued_bytes = torch.cuda.get_device_properties(x.device).total_memory // 3
            while len(self._step_events) > 1 and self._queued_bytes(x.device) > self.max_queued_bytes:
                self._step_events.pop(0).synchronize()
        if self.use_graph and x.is_cuda:
            loss = self._graph_step(x, gt)
        else:
            loss = self._fwd_bwd(x, gt)
            for h in self._handles:
                h.wait()
            self._join_wgrad_stream()
            self.opt.step(self.flat_g, self.n_live, grad_scale=1.0 / self.world)
            self._weights_changed()
            loss = loss.detach()
        if x.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._step_events.append(ev)
        return loss

    def _weights_changed(self):
        if self.use_hip and self.prepared_weights and self.flat_p.is_cuda:
            from . import ops
            ops.refresh_prepared_weights(self.flat_p.device)

    def weights_changed(self):
        """call after anything that writes the parameters outside torch's version tracking (p.data.copy_, raw kernels): the
        prepared weight operands are rebuilt.  load_state_dict / in-place torch ops are seen without it."""
        if self._ready:
            self._weights_changed()

    @staticmethod
    def _queued_bytes(device):
        st = torch.cuda.memory_stats_as_nested_dict(device)
        return st["active_bytes"]["all"]["current"] - st["allocated_bytes"]["all"]["current"]

    def forward_backward(self, x, gt):
        """forward + loss + backward only (gradients left in the flat arena); for timing splits."""
        if not self._ready:
            self._setup(x, gt)
        loss = self._fwd_bwd(x, gt)
        for h in self._handles:
            h.wait()
        self._join_wgrad_stream()
        return loss.detach()
