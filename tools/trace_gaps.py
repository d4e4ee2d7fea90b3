"""How much of a step is the GPU idle between kernels?  (dev tool)
Reads a rocprofv3 --kernel-trace CSV (b_kernel_trace.csv), takes the last `steps` training steps (delimited by the fused
Adam kernel), and reports per step: wall time, union of kernel intervals (busy), idle time, number of launches, and the
idle time by gap size.   python tools/trace_gaps.py <b_kernel_trace.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
adam = [i for i, e in enumerate(ev) if "adam" in e[2].lower()]
print(f"{len(ev)} kernels, {len(adam)} optimizer steps in the trace")
for s in range(max(1, len(adam) - steps), len(adam)):
    seg = ev[adam[s - 1] + 1:adam[s] + 1]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_end, gaps = 0, t0, []
    for a, b, _ in seg:
        if a > cur_end:
            gaps.append(a - cur_end)
            busy += b - a
            cur_end = b
        elif b > cur_end:
            busy += b - cur_end
            cur_end = b
    idle = sum(gaps)
    small = sum(g for g in gaps if g < 5000); mid = sum(g for g in gaps if 5000 <= g < 50000); big = sum(g for g in gaps if g >= 50000)
    ksum = sum(b - a for a, b, _ in seg)
    print(f"step {s}: wall {1e-6 * (t1 - t0):6.2f} ms  busy (union) {1e-6 * busy:6.2f}  idle {1e-6 * idle:5.2f} in {len(gaps)} gaps "
          f"(<5us: {1e-6 * small:4.2f} ms, 5-50us: {1e-6 * mid:4.2f}, >50us: {1e-6 * big:4.2f})  launches {len(seg)}  sum of kernel times {1e-6 * ksum:6.2f}")
