"""What is in the captured whole-step hipGraph, and how does its replay compare with eager issue? (dev tool)

Captures DataParallelTrainer(use_graph=True)'s step graph with debug mode on, dumps it (hipGraphDebugDotPrint through
torch.cuda.CUDAGraph.debug_dump), counts node kinds and dependency edges, and times replay against the eager step --
with the three-stream plan and with everything on one stream."""
import os, re, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvi_cidnet_amd as P
from hvi_cidnet_amd.dp import DataParallelTrainer

dev = torch.device("cuda:0")
B, H, W = 8, 400, 600
x = torch.rand(B, 3, H, W, device=dev); gt = torch.rand(B, 3, H, W, device=dev)
OUT = os.environ.get("GRAPH_PROBE_OUT", "gpurun_out")


def timed(tr, n=12):
    for _ in range(4): tr.step(x, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): tr.step(x, gt)
    t1 = time.perf_counter()                      # host enqueue done
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t2 - t0) / n, 1e3 * (t1 - t0) / n


for two, wg in ((True, True), (False, False)):
    label = "three streams" if two else "one stream"
    torch.manual_seed(0)
    m = P.CIDNet().to(dev); m.two_streams = two
    te = DataParallelTrainer(m, lr=1e-4, wgrad_stream=wg)
    ms_e, enq_e = timed(te)
    del te
    torch.manual_seed(0)
    m = P.CIDNet().to(dev); m.two_streams = two
    tg = DataParallelTrainer(m, lr=1e-4, wgrad_stream=wg, use_graph=True)
    # debug mode keeps the graph for dumping
    orig = torch.cuda.CUDAGraph
    class G(orig):
        def __new__(cls, *a, **k):
            g = orig.__new__(cls)
            return g
        def __init__(self, *a, **k):
            super().__init__()
            self.enable_debug_mode()
    torch.cuda.CUDAGraph = G
    try:
        ms_g, enq_g = timed(tg)
    finally:
        torch.cuda.CUDAGraph = orig
    dot = os.path.join(OUT, f"step_graph_{'3s' if two else '1s'}.dot")
    kinds, edges, nodes = collections.Counter(), 0, 0
    try:
        tg._graph.debug_dump(dot)
        txt = open(dot).read()
        for line in txt.splitlines():
            if "->" in line:
                edges += 1
            elif "label" in line:
                nodes += 1
                lab = re.search(r'label="([^"]*)"', line)
                s = lab.group(1) if lab else ""
                k = "kernel" if ("KERNEL" in s.upper() or "cidnet" in s or "void" in s) else "memcpy" if "MEMCPY" in s.upper() or "Memcpy" in s else \
                    "memset" if "MEMSET" in s.upper() or "Memset" in s else "empty" if "EMPTY" in s.upper() or "Empty" in s else "other"
                kinds[k] += 1
        os.remove(dot)                             # large; the counts are what is kept
    except Exception as e:                         # debug_dump may be unavailable on this stack
        kinds["(dump failed: %s)" % type(e).__name__] = 1
    print(f"{label}: eager {ms_e:6.2f} ms/step (host enqueue {enq_e:5.2f}) | graph replay {ms_g:6.2f} ms/step (host {enq_g:5.2f}) | "
          f"graph: {nodes} nodes {dict(kinds)}, {edges} edges", flush=True)
    del tg
    torch.cuda.empty_cache()
