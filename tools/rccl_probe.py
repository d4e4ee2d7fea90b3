"""How long does a 1-rank RCCL all-reduce of the gradient buckets take on an idle GPU? (dev tool)"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
for n in (500_000, 2_000_000):
    t = torch.ones(n, device=dev)
    for _ in range(5): dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dist.all_reduce(t)
    e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): dist.all_reduce(t)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{n * 4 / 1e6:.1f} MB: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per all_reduce on the GPU, {1e6 * (t1 - t0) / 20:.1f} us host enqueue")
    hs = []
    e0.record()
    for _ in range(20): hs.append(dist.all_reduce(t, async_op=True))
    for h in hs: h.wait()
    e1.record(); torch.cuda.synchronize()
    print(f"   async: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us each")
dist.destroy_process_group()
