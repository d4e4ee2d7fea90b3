"""GPU: the training-step harness (hvi-cidnet_amd/dp.py: flat gradient arena written in place by the backward
kernels, weight-gradient GEMMs on a third stream, two branch streams, fused Adam) must produce exactly the
gradients / update of plain autograd + torch.optim.Adam on the same model."""
import pytest
import torch

from oracle import cidnet_oracle as O

pytestmark = pytest.mark.gpu


def _model(dev, chans):
    import hvi_cidnet_amd as P
    m = P.CIDNet(channels=list(chans))
    p = O.make_params(21, channels=chans)
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
    return m.to(dev)


@pytest.mark.parametrize("chans,shape", [((12, 12, 24, 48), (2, 3, 32, 48)), ((36, 36, 72, 144), (2, 3, 64, 96))])
def test_trainer_gradients_and_update_match_plain_autograd(dev, chans, shape):
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    x = O.synthetic_batch(31, shape).to(dev)
    gt = O.synthetic_batch(32, shape).to(dev)
    # plain autograd reference (no arena, single stream)
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    ref = _model(dev, chans)
    ref.two_streams = False
    loss_r = ops.L1LossFn.apply(ref(x), gt)
    loss_r.backward()
    grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
    opt = torch.optim.Adam([p for p in ref.parameters() if p.grad is not None], lr=1e-3)
    opt.step()
    # harness
    m = _model(dev, chans)
    tr = DataParallelTrainer(m, lr=1e-3, n_buckets=3, wgrad_stream=True)
    loss_t = tr.forward_backward(x, gt)
    torch.cuda.synchronize()
    assert abs(loss_t.item() - loss_r.item()) < 1e-6
    names = {id(p): n for n, p in m.named_parameters()}
    checked = 0
    for p in tr.params:
        n = names[id(p)]
        if n not in grads:
            continue
        off, cnt = tr._slices[id(p)]
        got = tr.flat_g[off:off + cnt].view(p.shape)
        ref_g = grads[n]
        tol = 1e-5 * ref_g.abs().max().item() + 1e-8
        err = (got - ref_g).abs().max().item()
        assert err <= tol, f"{n}: {err:.3e} > {tol:.3e}"
        checked += 1
    assert checked > 150
    # LayerNorm modules used several times per step sum their gradients in place in the arena (ops.LNUse): every live
    # norm is in that mode after the probe, the dead I_LCA5.norm is not, and the counters are back at rest
    uses = {n: mod._use for n, mod in m.named_modules() if hasattr(mod, "_use")}
    assert len(uses) == 12 and all(st.acc == (not n.startswith("I_LCA5")) for n, st in uses.items()), {n: st.acc for n, st in uses.items()}
    assert all(st.fwd == 0 and st.bwd == 0 for st in uses.values())
    # one full step == torch Adam fed with the harness's own gradients.  (Fed with the plain-autograd gradients
    # instead, first-step Adam ~ lr * g / (|g| + eps) amplifies summation-order noise on elements with
    # |g| ~ eps into differences of the order of lr, so that comparison would test luck.)
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    m2 = _model(dev, chans)
    before = {n: p.detach().clone() for n, p in m2.named_parameters()}
    tr2 = DataParallelTrainer(m2, lr=1e-3, n_buckets=3)
    tr2.step(x, gt)
    torch.cuda.synchronize()
    twins = []
    for n, p in m2.named_parameters():
        if n not in grads:
            assert torch.equal(p, before[n]), n              # dead I_LCA5.*: untouched
            continue
        off, cnt = tr2._slices[id(p)]
        g2 = tr2.flat_g[off:off + cnt].view(p.shape)
        assert (g2 - grads[n]).abs().max().item() <= 1e-5 * grads[n].abs().max().item() + 1e-8, n
        t = before[n].clone().requires_grad_(True)
        t.grad = g2.clone()
        twins.append((n, p, t))
    torch.optim.Adam([t for _, _, t in twins], lr=1e-3).step()
    for n, p, t in twins:
        assert torch.allclose(p, t, atol=1e-7, rtol=1e-6), n
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)


def test_trainer_survives_stray_grad_mode_forward(dev):
    """a grad-enabled forward WITHOUT a backward between two trainer steps (validation without
    no_grad, model(x) for logging) used to leave the in-place LayerNorm gradient counters with fwd ahead of bwd for good:
    the 22 live LayerNorm gradients were then never handed to autograd, their bucket never completed and stale values
    piled up in the arena -- silently.  The trainer zeroes the counters at the start of every pass and raises when a
    bucket is left pending; the gradients of the step after the stray forward must equal plain autograd's."""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    chans, shape = (12, 12, 24, 48), (2, 3, 32, 48)
    x = O.synthetic_batch(41, shape).to(dev)
    gt = O.synthetic_batch(42, shape).to(dev)
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    ref = _model(dev, chans)
    ref.two_streams = False
    ops.L1LossFn.apply(ref(x), gt).backward()
    grads = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
    m = _model(dev, chans)
    tr = DataParallelTrainer(m, lr=0.0, n_buckets=3, wgrad_stream=True)       # lr 0: the weights stay the reference's
    tr.step(x, gt)
    y_stray = m(x)                                                           # grad mode, never run backward
    assert y_stray.requires_grad
    uses = [mod._use for mod in m.modules() if hasattr(mod, "_use")]
    assert any(st.fwd != st.bwd for st in uses)                              # the counters ARE out of balance now
    tr.forward_backward(x, gt)
    torch.cuda.synchronize()
    names = {id(p): n for n, p in m.named_parameters()}
    for p in tr.params:
        n = names[id(p)]
        if n not in grads:
            continue
        off, cnt = tr._slices[id(p)]
        got = tr.flat_g[off:off + cnt].view(p.shape)
        err = (got - grads[n]).abs().max().item()
        assert err <= 1e-5 * grads[n].abs().max().item() + 1e-8, f"{n}: {err:.3e}"
    del y_stray
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)


def test_model_deepcopy_and_save_after_forward(dev, tmp_path):
    """streams / events used by the two-stream forward and its back-pressure live outside the module, so
    a model that has run stays copy.deepcopy()-able (EMA copies) and torch.save()-able; the copy computes the same."""
    import copy
    m = _model(dev, (12, 12, 24, 48))
    x = O.synthetic_batch(43, (1, 3, 32, 48)).to(dev)
    with torch.no_grad():
        y = m(x)
        m2 = copy.deepcopy(m)
        torch.save(m, tmp_path / "model.pt")
        assert torch.equal(m2(x), y)
    assert m._side_stream is not None


def test_trainer_rccl_path_on_one_rank(dev, monkeypatch):
    """The multi-GPU code path (bucket all-reduces launched from gradient hooks, joined with the weight-gradient
    stream, 1/world folded into Adam) exercised on ONE rank over RCCL: a 1-rank all-reduce is the identity, so
    gradients and the update must equal the no-communication run bit for bit."""
    import torch.distributed as dist
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    chans, shape = (12, 12, 24, 48), (2, 3, 32, 48)
    x = O.synthetic_batch(41, shape).to(dev)
    gt = O.synthetic_batch(42, shape).to(dev)
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    m0 = _model(dev, chans)
    t0 = DataParallelTrainer(m0, lr=1e-3, n_buckets=3)
    t0.step(x, gt)
    torch.cuda.synchronize()
    ref_g = t0.flat_g.clone()
    ref_p = t0.flat_p.clone()
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    monkeypatch.setenv("CIDNET_DP_FORCE_ALLREDUCE", "1")
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29571", world_size=1, rank=0)
    try:
        m1 = _model(dev, chans)
        t1 = DataParallelTrainer(m1, lr=1e-3, n_buckets=3)
        t1.step(x, gt)
        torch.cuda.synchronize()
        assert t1._handles == [] or all(h.is_completed() for h in t1._handles)
        assert torch.equal(t1.flat_g, ref_g)
        assert torch.equal(t1.flat_p, ref_p)
    finally:
        dist.destroy_process_group()
        ops.set_grad_arena(None, None)
        ops.enable_wgrad_stream(False)


@pytest.mark.parametrize("chans,shape", [((12, 12, 24, 48), (2, 3, 32, 48)), ((36, 36, 72, 144), (1, 3, 64, 96))])
def test_trainer_graph_replay_equals_eager(dev, chans, shape):
    """forward + loss + backward replayed as one hipGraph (three streams captured) must give the eager step's
    gradients and parameter updates bit for bit, step after step (new inputs are copied into the static buffers).  The
    full-width case runs the split-product kernels inside the capture: their prepared weight operands are cache hits there
    (created by the eager passes before the capture) and are refreshed after every Adam update outside the graph."""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    batches = [(O.synthetic_batch(51 + i, shape).to(dev), O.synthetic_batch(61 + i, shape).to(dev)) for i in range(3)]
    res = []
    for use_graph in (False, True):
        ops.set_grad_arena(None, None)
        ops.enable_wgrad_stream(False)
        m = _model(dev, chans)
        tr = DataParallelTrainer(m, lr=1e-3, n_buckets=3, use_graph=use_graph)
        losses = [float(tr.step(x, gt).item()) for x, gt in batches]
        torch.cuda.synchronize()
        res.append((losses, tr.flat_g.clone(), tr.flat_p.clone()))
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2])


def test_trainer_with_reference_objective(dev):
    """the training step driven by the reference's objective (L1 + SSIM + Edge on RGB and on HVIT(out), train.py:61-65
    without the VGG term): gradients reach every live parameter, including density_k through the loss-side HVIT,
    and a few steps reduce the loss"""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)
    chans, shape = (12, 12, 24, 48), (2, 3, 32, 48)
    m = _model(dev, chans)
    x = O.synthetic_batch(71, shape).to(dev)
    gt = O.synthetic_batch(72, shape).to(dev)
    tr = DataParallelTrainer(m, lr=2e-3, n_buckets=3, loss_fn=P.CIDNetLoss(m))
    losses = [float(tr.step(x, gt).item()) for _ in range(6)]
    torch.cuda.synchronize()
    assert all(l == l for l in losses), losses
    assert losses[-1] < losses[0], losses
    assert tr.flat_g[:tr.n_live].abs().max().item() > 0
    ops.set_grad_arena(None, None)
    ops.enable_wgrad_stream(False)


@pytest.mark.parametrize("two_streams", [True, False])
def test_training_step_launches_only_cidnet_kernels(dev, two_streams):
    """beside LDS-fed bf16 MFMAs, packed-fp32 instructions with op_sel in ANOTHER kernel's waves were seen
    to lose products; libcidnet_hip.so is built without them, ATen's elementwise kernels are not
    (profiles/r04_a_audit_aten_add_gfx950_packed_fp32.json: v_pk_fma_f32 ... op_sel_hi:[0,1,1] in every
    vectorized_elementwise_kernel<CUDAFunctor_add<float>>).  So one training step of the shipped configuration -- forward,
    L1 loss, backward into the arena, fused Adam -- must launch nothing but cidnet:: kernels (at N > 1: plus RCCL's
    FuncSum kernels, audited in profiles/r04_a_audit_rccl_gfx950_packed_fp32.json): no autograd accumulation adds, no
    slice copies, no fills, no device-to-device memcpy."""
    import os
    import sys
    from hvi_cidnet_amd.dp import DataParallelTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        from foreign_kernels_in_step import foreign_kernels
    finally:
        sys.path.pop(0)
    chans, shape = (36, 36, 72, 144), (2, 3, 64, 96)
    x = O.synthetic_batch(41, shape).to(dev)
    gt = O.synthetic_batch(42, shape).to(dev)
    m = _model(dev, chans)
    m.two_streams = two_streams
    tr = DataParallelTrainer(m, lr=1e-3, wgrad_stream=two_streams)
    for _ in range(3):
        tr.step(x, gt)
    fk = foreign_kernels(tr, x, gt, steps=1)
    assert not fk, {k: [(op, shp) for op, shp, _ in v][:3] for k, v in fk.items()}


def test_trainer_prepared_weights_bit_identical(dev):
    """DataParallelTrainer(prepared_weights=True): the bf16x3 convs read weight operands prepared once per optimizer step
    (two batched launches after the fused Adam) instead of once per launch -- parameters after three steps are bit-identical
    to the per-call preparation, and the cache is actually hit"""
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    chans, shape = (36, 36, 72, 144), (2, 3, 64, 96)
    x = O.synthetic_batch(41, shape).to(dev)
    gt = O.synthetic_batch(42, shape).to(dev)
    finals = []
    try:
        for prepared in (False, True):
            m = _model(dev, chans)
            tr = DataParallelTrainer(m, lr=1e-3, prepared_weights=prepared)
            ops._PREP["stats"][:] = [0, 0]
            for _ in range(3):
                tr.step(x, gt)
            torch.cuda.synchronize()
            finals.append(tr.flat_p.clone())
            if prepared:
                hits, misses = ops._PREP["stats"]
                assert misses > 50 and hits >= 2 * misses - 10, (hits, misses)    # prepared in the first pass, reused afterwards
                assert not ops._PREP["on"]             # the cache is consulted inside the trainer's passes only
        assert torch.equal(finals[0], finals[1])
    finally:
        ops.enable_prepared_weights(False)
        ops.set_grad_arena(None, None)
        ops.enable_wgrad_stream(False)


def test_trainer_with_tnsm_model_and_its_objective(dev):
    """CIDNet_TNSM in train mode returns (rgb, fused noise map): DataParallelTrainer hands the pair and the step's input to
    losses.CIDNetLoss(tnsm_weight=1) (train_tnsm.py:56-72; options.py:61), whose noise terms then reach the noise branch;
    without a noise map a positive tnsm_weight raises instead of silently dropping the terms."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    chans, shape = (12, 12, 24, 48), (2, 3, 32, 48)
    torch.manual_seed(3)
    m = P.CIDNet_TNSM(channels=list(chans)).to(dev).train()
    x = O.synthetic_batch(51, shape).to(dev)
    gt = O.synthetic_batch(52, shape).to(dev)
    lf = P.CIDNetLoss(m, tnsm_weight=1.0).to(dev)
    try:
        with pytest.raises(ValueError, match="noise map"):
            lf(torch.rand(shape, device=dev), gt)
        # manual composition of the same objective
        rgb, noise = m(x)
        manual = P.CIDNetLoss(m, tnsm_weight=0.0).to(dev)(rgb, gt) + P.tnsm_noise_loss(noise, rgb, x, 1.0)
        tr = DataParallelTrainer(m, lr=1e-4, loss_fn=lf)
        loss = tr.forward_backward(x, gt)
        torch.cuda.synchronize()
        assert abs(loss.item() - manual.item()) <= 1e-5 * abs(manual.item()) + 1e-6
        names = {id(p): n for n, p in m.named_parameters()}
        noise_params = [p for p in tr.params if "noise_map_generator" in names[id(p)] and id(p) in tr._slices
                        and not names[id(p)].startswith("I_TNSM5")]
        assert noise_params
        off, n = tr._slices[id(noise_params[0])]
        assert tr.flat_g[off:off + n].abs().max().item() > 0
        l2 = tr.step(x, gt)
        assert torch.isfinite(l2)
    finally:
        ops.clear_prepared_weights()
        ops.set_grad_arena(None, None)
        ops.enable_wgrad_stream(False)


def test_bf16_mode_training_tracks_fp32(dev):
    """BASELINE.json configs[2] trains in bf16: 25 Adam steps of the full-width model on a fixed synthetic pair in the bf16 mode
    (bf16 matrix-core operands, bf16 LCA-internal tensors) must follow the fp32 run -- the loss falls (0.361 -> 0.106 / 0.105), and stays within
    6 % of the fp32 loss at every step (measured: 2.1 %; two Adam trajectories drift apart); parameters stay finite.  Gradient quality is pinned separately against
    fp64 (tests/test_fullsize_gpu.py::test_cidnet_400x600_bf16_mode_vs_reference)."""
    import hvi_cidnet_amd as P
    from hvi_cidnet_amd import ops
    from hvi_cidnet_amd.dp import DataParallelTrainer
    chans, shape = (36, 36, 72, 144), (2, 3, 64, 96)
    x = (O.synthetic_batch(61, shape) * 0.3).to(dev)           # a dark input ...
    gt = O.synthetic_batch(61, shape).to(dev)                   # ... and its bright version
    curves = {}
    try:
        for prec in ("f32", "bf16"):
            P.set_precision(prec)
            m = _model(dev, chans)
            tr = DataParallelTrainer(m, lr=2e-4)
            losses = [tr.step(x, gt) for _ in range(25)]
            torch.cuda.synchronize()
            curves[prec] = [float(l.item()) for l in losses]
            assert torch.isfinite(tr.flat_p).all()
            del tr, m
    finally:
        P.set_precision("f32")
        ops.clear_prepared_weights()
        ops.set_grad_arena(None, None)
        ops.enable_wgrad_stream(False)
    f, b = curves["f32"], curves["bf16"]
    assert f[-1] < 0.9 * f[0] and b[-1] < 0.9 * b[0], (f[0], f[-1], b[0], b[-1])
    worst = max(abs(p - q) / q for p, q in zip(b, f))
    print(f"bf16 vs fp32 training: loss {f[0]:.4f} -> {f[-1]:.4f} (fp32), {b[0]:.4f} -> {b[-1]:.4f} (bf16); worst relative gap {worst:.2e}")
    assert worst <= 6e-2, worst
