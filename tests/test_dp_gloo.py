"""CPU / gloo, world_size=2: the data-parallel harness (hvi-cidnet_amd/dp.py) -- gradient-ready-order
flat buckets, overlapped all-reduce from post-accumulate hooks, dead-parameter exclusion, flat Adam --
is checked against single-process torch.optim.Adam on the concatenated batch.  The harness is
model-agnostic; a small plain-torch network stands in for CIDNet here because the product kernels
have no CPU path (the N>1 GPU path differs only in the backend string and the fused kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.norm_w = nn.Parameter(torch.ones(8))       # used twice per step (multi-use parameter)
        self.b = nn.Conv2d(8, 3, 1)
        self.dead = nn.Linear(4, 4)                      # never used: grad stays None (like I_LCA5)

    def forward(self, x):
        h = torch.tanh(self.a(x)) * self.norm_w[None, :, None, None]
        h = h * self.norm_w[None, :, None, None]
        return self.b(h)


class OracleCIDNet(nn.Module):
    """CPU stand-in with CIDNet's exact parameter tree (the reference's 191 state_dict names, reduced width) whose
    forward is the oracle's functional restatement: the same multi-use LayerNorm parameters (one norm module applied
    2-3 times per LCA, so autograd accumulates before the post-accumulate hook fires) and the same dead I_LCA5.*
    block (13 tensors with grad None) that the trainer must bucket / exclude on the real model."""

    def __init__(self, channels=(12, 12, 24, 48), seed=3):
        super().__init__()
        from oracle import cidnet_oracle as O
        self._names = []
        for name, v in O.make_params(seed, channels=channels).items():
            mod = self
            *path, leaf = name.split(".")
            for part in path:
                key = "m_" + part if part.isdigit() else part
                if not hasattr(mod, key):
                    setattr(mod, key, nn.Module())
                mod = getattr(mod, key)
            mod.register_parameter(leaf, nn.Parameter(v.clone()))
            self._names.append((name, mod, leaf))

    def forward(self, x):
        from oracle import cidnet_oracle as O
        return O.cidnet_forward({n: getattr(m, leaf) for n, m, leaf in self._names}, x)


def l1(out, gt):
    return (out - gt).abs().mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, steps, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hvi_cidnet_amd.dp import DataParallelTrainer
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    net = Net()
    tr = DataParallelTrainer(net, lr=1e-2, n_buckets=3, loss_fn=l1, use_hip_kernels=False)
    g = torch.Generator().manual_seed(7)
    xs = torch.rand(steps, world * 2, 3, 8, 8, generator=g)
    gts = torch.rand(steps, world * 2, 3, 8, 8, generator=g)
    for s in range(steps):
        tr.step(xs[s, rank * 2:(rank + 1) * 2], gts[s, rank * 2:(rank + 1) * 2])
    q.put((rank, {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}, len(tr.buckets), tr.n_live))  # by value
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_two_ranks_match_single_process_adam():
    world, steps = 2, 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, sd, nb, n_live = q.get(timeout=240)
        res[r] = ({k: torch.from_numpy(v) for k, v in sd.items()}, nb, n_live)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sd0, nb, n_live = res[0]
    assert nb >= 2
    for k in sd0:                                         # ranks stay bit-identical
        assert torch.equal(sd0[k], res[1][0][k]), k
    # single-process reference: rank 0's initial weights, full batch, torch Adam
    torch.manual_seed(100)
    ref = Net()
    live = [p for n, p in ref.named_parameters() if not n.startswith("dead")]
    assert n_live == sum(p.numel() for p in live)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    xs = torch.rand(steps, world * 2, 3, 8, 8, generator=g)
    gts = torch.rand(steps, world * 2, 3, 8, 8, generator=g)
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        # mean over the global batch == average of the per-rank means (equal shard sizes)
        l1(ref(xs[s]), gts[s]).backward()
        opt.step()
    for k, v in ref.state_dict().items():
        assert torch.allclose(sd0[k], v, atol=2e-6, rtol=1e-5), (k, (sd0[k] - v).abs().max())


def _worker_cidnet(rank, world, port, steps, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hvi_cidnet_amd.dp import DataParallelTrainer
    net = OracleCIDNet(seed=3 + rank)                   # different init per rank: broadcast must fix it
    tr = DataParallelTrainer(net, lr=1e-3, n_buckets=4, loss_fn=l1, use_hip_kernels=False)
    g = torch.Generator().manual_seed(11)
    xs = torch.rand(steps, world, 3, 16, 24, generator=g)
    gts = torch.rand(steps, world, 3, 16, 24, generator=g)
    for s in range(steps):
        tr.step(xs[s, rank:rank + 1], gts[s, rank:rank + 1])
    dead = sorted(n for n, m, leaf in net._names if id(getattr(m, leaf)) not in tr._bucket_of)
    sd = {n: getattr(m, leaf).detach().numpy().copy() for n, m, leaf in net._names}
    q.put((rank, sd, len(tr.buckets), tr.n_live, dead))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp_two_ranks_cidnet_tree_match_single_process_adam():
    """the REAL trainer (probe, gradient-ready-order flat layout, 4 buckets launched from hooks, dead-block exclusion,
    flat Adam) on CIDNet's parameter tree, 2 gloo ranks, against single-process torch.optim.Adam on the global batch"""
    world, steps = 2, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cidnet, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, sd, nb, n_live, dead = q.get(timeout=500)
        res[r] = (sd, nb, n_live, dead)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sd0, nb, n_live, dead = res[0]
    assert nb == 4
    assert len(dead) == 13 and all(n.startswith("I_LCA5.") for n in dead), dead
    for k in sd0:
        assert (sd0[k] == res[1][0][k]).all(), k         # ranks stay bit-identical (dead block included: broadcast)
    ref = OracleCIDNet(seed=3)
    assert n_live == sum(getattr(m, leaf).numel() for n, m, leaf in ref._names if not n.startswith("I_LCA5."))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(11)
    xs = torch.rand(steps, world, 3, 16, 24, generator=g)
    gts = torch.rand(steps, world, 3, 16, 24, generator=g)
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        l1(ref(xs[s]), gts[s]).backward()
        opt.step()
    worst = 0.0
    for n, m, leaf in ref._names:
        v = getattr(m, leaf).detach()
        d = (torch.from_numpy(sd0[n]) - v).abs().max().item()
        worst = max(worst, d)
        # Adam's first steps move every weight by ~lr regardless of gradient scale: compare against lr
        assert d <= 2e-2 * 1e-3 + 1e-7, (n, d)
    print("worst |delta| vs single-process Adam:", worst)


class NetSwapped(Net):
    """same parameters as Net, but the two halves of the forward are ordered so that gradients become ready in another
    order: a rank running this module derives a different arena layout"""

    def forward(self, x):
        return super().forward(x) + 0.0 * self.dead(torch.zeros(1, 4)).sum()      # `dead` now receives a (zero) gradient


def _worker_layout(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hvi_cidnet_amd.dp import DataParallelTrainer
    torch.manual_seed(5)
    net = Net() if rank == 0 else NetSwapped()
    tr = DataParallelTrainer(net, lr=1e-2, n_buckets=2, loss_fn=l1, use_hip_kernels=False)
    x, gt = torch.rand(2, 3, 8, 8), torch.rand(2, 3, 8, 8)
    try:
        tr.step(x, gt)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_layout_mismatch_between_ranks_raises():
    """every rank derives the gradient-arena layout from its own probing backward; if two ranks ever
    disagree, the bucket all-reduce would add different parameters into each other silently.  The trainer now compares a
    digest of the layout across ranks in _setup and raises on every rank."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_layout, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert "layout differs between ranks" in res[0], res
    assert "layout differs between ranks" in res[1], res


def test_rank_cpu_sets_partition_the_job():
    """per-rank CPU affinity (dp.pin_rank_to_cpus): the local ranks' slices are disjoint, contiguous, cover every CPU and
    differ by at most one in size; with fewer CPUs than ranks every rank still gets one"""
    from hvi_cidnet_amd.dp import rank_cpu_set
    for cpus, world in ((list(range(8)), 2), (list(range(3, 131, 2)), 8), (list(range(16)), 3), ([5], 1)):
        parts = [rank_cpu_set(cpus, r, world) for r in range(world)]
        assert sorted(c for p in parts for c in p) == sorted(cpus)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        for p in parts:
            assert p == sorted(cpus)[sorted(cpus).index(p[0]):sorted(cpus).index(p[0]) + len(p)]
    assert [rank_cpu_set([0, 1], r, 4) for r in range(4)] == [[0], [1], [0], [1]]


def _affinity_worker(rank, world, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hvi_cidnet_amd.dp import pin_rank_to_cpus
    before = sorted(os.sched_getaffinity(0))
    mine = pin_rank_to_cpus(rank, world)
    q.put((rank, before, mine, sorted(os.sched_getaffinity(0)), torch.get_num_threads()))


@pytest.mark.timeout(120)
def test_pin_rank_to_cpus_two_processes():
    if not hasattr(os, "sched_setaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs two usable CPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_affinity_worker, args=(r, 2, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=100) for _ in ps)
    for p in ps:
        p.join(30)
    (_, before0, mine0, after0, nt0), (_, before1, mine1, after1, nt1) = got
    assert before0 == before1 and mine0 == after0 and mine1 == after1
    assert not set(mine0) & set(mine1) and sorted(mine0 + mine1) == before0
    assert 1 <= nt0 <= len(mine0) and 1 <= nt1 <= len(mine1)
