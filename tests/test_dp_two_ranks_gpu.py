"""Two data-parallel ranks SHARING the one GPU of the test box (gloo backend, CUDA tensors): the trainer's real path --
HIP kernels, the two branch streams, the weight-gradient stream, gradient-ready-order buckets all-reduced from the
post-accumulate hooks on the launch stream -- with a real exchange between two processes.  RCCL refuses two ranks on
one device, so the collective here is gloo's; everything on our side of `dist.all_reduce` is what runs at N > 1.
Checked: the all-reduced mean gradient equals a single process's gradient on the concatenated batch; after Adam steps
the two ranks hold bit-identical parameters."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (3, 64, 96)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(world, per_rank):
    g = torch.Generator().manual_seed(5)
    x = torch.rand(world * per_rank, *SHAPE, generator=g) * 0.6 + 0.05
    gt = torch.rand(world * per_rank, *SHAPE, generator=g)
    return x, gt


def _model():
    sys.path.insert(0, ROOT)
    import hvi_cidnet_amd as P
    from oracle import cidnet_oracle as O
    m = P.CIDNet()
    p = O.make_params(7)
    m.load_state_dict({k: p[k] for k in m.state_dict().keys()})
    return m.to("cuda:0")


def _worker(rank, world, port, per_rank, q, precision="f32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hvi_cidnet_amd.dp import DataParallelTrainer
    import hvi_cidnet_amd as P
    P.set_precision(precision)
    m = _model()
    tr = DataParallelTrainer(m, lr=1e-4, n_buckets=4)
    x, gt = _data(world, per_rank)
    xs = x[rank * per_rank:(rank + 1) * per_rank].cuda()
    gs = gt[rank * per_rank:(rank + 1) * per_rank].cuda()
    tr.forward_backward(xs, gs)
    torch.cuda.synchronize()
    g_mean = (tr.flat_g[:tr.n_live] / world).cpu().numpy().copy()
    for _ in range(2):
        tr.step(xs, gs)
    torch.cuda.synchronize()
    q.put((rank, g_mean, tr.flat_p[:tr.n_live].cpu().numpy().copy(), len(tr.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_two_ranks_on_one_gpu_match_single_process(precision):
    """precision = "bf16": BASELINE.json configs[2]'s numeric mode (bf16 matrix-core operands, bf16 storage of the LCA-internal
    tensors) through the same two-rank path: every sample's arithmetic is independent of the batch it sits in, so the
    reduced mean gradient still equals the single-process gradient on the concatenated batch to fp32 summation order."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, per_rank = 2, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q, precision)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, g, p_, nb = q.get(timeout=500)
        res[r] = (torch.from_numpy(g), torch.from_numpy(p_), nb)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][2] >= 2                                                   # more than one bucket in flight
    assert torch.equal(res[0][0], res[1][0])                                # both ranks saw the same reduced gradient
    assert torch.equal(res[0][1], res[1][1])                                # ... and made the same updates
    # single process, concatenated batch (mean loss over the global batch = mean of the per-rank means)
    sys.path.insert(0, ROOT)
    from hvi_cidnet_amd.dp import DataParallelTrainer
    from hvi_cidnet_amd import ops
    import hvi_cidnet_amd as P
    P.set_precision(precision)
    try:
        m = _model()
        tr = DataParallelTrainer(m, lr=1e-4, n_buckets=4)
        x, gt = _data(world, per_rank)
        tr.forward_backward(x.cuda(), gt.cuda())
        torch.cuda.synchronize()
        ref = tr.flat_g[:tr.n_live].cpu()
    finally:
        P.set_precision("f32")
        ops.clear_prepared_weights()
        ops.set_grad_arena(None, None)
        ops.enable_wgrad_stream(False)
    d = (res[0][0] - ref).abs().max().item()
    assert d <= 2e-5 * ref.abs().max().item() + 1e-9, (d, ref.abs().max().item())
