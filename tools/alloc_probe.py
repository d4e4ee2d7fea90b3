"""How long does a fresh device allocation take on this stack? (dev tool: explains multi-second host stalls when the
caching allocator has to grow in the middle of a step)"""
import time, torch
torch.cuda.init(); torch.cuda.synchronize()
for gb in (0.25, 0.5, 1, 2, 4, 8, 0.5, 1):
    n = int(gb * (1 << 30))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    t.fill_(1); torch.cuda.synchronize(); t2 = time.perf_counter()
    del t
    torch.cuda.synchronize(); t3 = time.perf_counter()
    torch.cuda.empty_cache(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"{gb:5.2f} GiB: alloc {1e3 * (t1 - t0):8.2f} ms, first touch {1e3 * (t2 - t1):8.2f} ms, free-to-cache {1e3 * (t3 - t2):6.2f} ms, empty_cache {1e3 * (t4 - t3):8.2f} ms", flush=True)
