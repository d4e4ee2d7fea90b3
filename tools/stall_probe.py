"""Root cause of the multi-second host stalls seen when the trainer runs unthrottled: one run with
`max_steps_in_flight` lifted, logging per step the host enqueue time and the caching allocator's counters
(reserved / active / allocated bytes, device mallocs and frees, allocation retries, pending cross-stream events)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer

dev = torch.device("cuda:0")
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
torch.manual_seed(0)
model = P.CIDNet().to(dev)
tr = DataParallelTrainer(model, lr=1e-4, n_buckets=4, max_steps_in_flight=inflight)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.rand((8, 3, 400, 600), device=dev, generator=g)
gt = torch.rand((8, 3, 400, 600), device=dev, generator=g)
for _ in range(3):
    tr.step(x, gt)
torch.cuda.synchronize()
keys = ["reserved_bytes.all.current", "active_bytes.all.current", "allocated_bytes.all.current", "inactive_split_bytes.all.current",
        "num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "num_ooms"]
rows = []
ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
t_prev = time.perf_counter()
for i in range(steps):
    tr.step(x, gt)
    t = time.perf_counter()
    st = torch.cuda.memory_stats(dev)
    rows.append({"step": i, "enqueue_ms": round(1e3 * (t - t_prev), 2), **{k.split(".")[0]: st.get(k, 0) for k in keys}})
    t_prev = t
ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
torch.cuda.synchronize()
tot = ev0.elapsed_time(ev1)
print(f"max_steps_in_flight={inflight}: {steps} steps in {tot:.1f} ms GPU-side ({tot / steps:.2f} ms/step)")
for r in rows:
    print(f"  step {r['step']:2d} enqueue {r['enqueue_ms']:8.2f} ms  reserved {r['reserved_bytes'] / 2**30:7.2f} GiB  active {r['active_bytes'] / 2**30:7.2f}  "
          f"allocated {r['allocated_bytes'] / 2**30:7.2f}  inactive-split {r['inactive_split_bytes'] / 2**30:6.2f}  dev-mallocs {r['num_device_alloc']:5d}  "
          f"dev-frees {r['num_device_free']:4d}  retries {r['num_alloc_retries']}  sync-all {r['num_sync_all_streams']}  ooms {r['num_ooms']}")
