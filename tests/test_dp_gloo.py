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
