"""Which part of a step runs with the chip under-filled?  (dev tool)
Reads a rocprofv3 --kernel-trace CSV, takes the last `steps` training steps (delimited by the fused Adam kernel) and splits
each step's wall time by the LARGEST grid (in workgroups) resident at that moment: idle, only launches below 64 / 256 /
1024 workgroups, or at least one big launch.  Also lists which small kernels own the under-filled time.
    python tools/trace_fill.py <b_kernel_trace.csv> [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def blocks(r):
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    return max(1, g // max(w, 1))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], blocks(r), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows), key=lambda t: t[0])
adam = [i for i, e in enumerate(ev) if "adam" in e[2].lower()]
print(f"{len(ev)} kernels, {len(adam)} optimizer steps in the trace")
def short(n):
    n = n.replace("cidnet::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]
for s in range(max(1, len(adam) - steps), len(adam)):
    seg = ev[adam[s - 1] + 1:adam[s] + 1]
    pts = sorted(set([e[0] for e in seg] + [e[1] for e in seg]))
    t0, t1 = pts[0], pts[-1]
    # sweep
    starts = collections.defaultdict(list); ends = collections.defaultdict(list)
    for i, e in enumerate(seg):
        starts[e[0]].append(i); ends[e[1]].append(i)
    live = set()
    cls = collections.Counter(); owner = collections.Counter()
    qbusy = collections.Counter()
    for a, b in zip(pts[:-1], pts[1:]):
        for i in ends.get(a, ()): live.discard(i)
        for i in starts.get(a, ()): live.add(i)
        d = b - a
        if not live:
            cls["idle"] += d; continue
        mx = max(seg[i][3] for i in live)
        c = "<64" if mx < 64 else "<256" if mx < 256 else "<1024" if mx < 1024 else ">=1024"
        cls[c] += d
        if mx < 256:
            for i in live: owner[short(seg[i][2])] += d / len(live)
        for q in set(seg[i][4] for i in live): qbusy[q] += d
    tot = t1 - t0
    print(f"step {s}: wall {1e-6*tot:6.2f} ms  " + "  ".join(f"{k}: {1e-6*cls[k]:5.2f}" for k in ("idle", "<64", "<256", "<1024", ">=1024")) + f"  launches {len(seg)}")
    print("   busy per queue: " + "  ".join(f"{q}: {1e-6*v:5.2f}" for q, v in sorted(qbusy.items())))
    print("   under-filled (<256 workgroups) time by kernel: " + ", ".join(f"{k} {1e-6*v:.2f}" for k, v in owner.most_common(12)))

# ---- idle gaps of the last step: which kernel ended before, which started after ----
if len(adam) >= 4:
    s = len(adam) - 3                      # the last three-stream step (the two after it are the instrumented single-stream ones)
    seg = ev[adam[s - 1] + 1:adam[s] + 1]
    cur_end, cur_name, gaps = seg[0][0], "(start)", []
    for a, b, name, blk, q in seg:
        if a > cur_end:
            gaps.append((a - cur_end, cur_name, short(name), q))
        if b > cur_end:
            cur_end, cur_name = b, short(name)
    agg = collections.Counter(); cnt = collections.Counter()
    for g, before, after, q in gaps:
        agg[(before, after)] += g; cnt[(before, after)] += 1
    print(f"idle gaps of step {s}: {len(gaps)} gaps, {1e-6 * sum(g for g, *_ in gaps):.2f} ms; largest contributors (ended before -> started after):")
    for (before, after), v in agg.most_common(25):
        print(f"   {1e-3 * v:7.1f} us in {cnt[(before, after)]:3d}:  {before[:48]:48s} -> {after[:48]}")
