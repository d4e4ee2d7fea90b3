"""Summarise a rocprofv3 kernel trace of bench.py: per step, GPU idle time (no kernel running) and the largest gaps (dev tool).

    rocprofv3 --kernel-trace -d out -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
    python tools/trace_gaps.py out/t_kernel_trace.csv
"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps are delimited by the fused Adam kernel
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
print("kernels", len(rows), "adam steps", len(adam))
for a, b in zip(adam[:-1], adam[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], seg[-1][1]
    busy_end, idle, gaps = t0, 0, []
    for s, e, n in seg:
        if s > busy_end:
            idle += s - busy_end
            gaps.append((s - busy_end, n[:60]))
        busy_end = max(busy_end, e)
    gaps.sort(reverse=True)
    print(f"step: wall {(t1 - t0) / 1e6:.2f} ms, kernels {len(seg)}, idle {idle / 1e6:.2f} ms in {len(gaps)} gaps; largest: " +
          "; ".join(f"{g / 1e3:.0f}us before {n}" for g, n in gaps[:5]))
