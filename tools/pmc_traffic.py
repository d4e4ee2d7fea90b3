"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/<tag>_pmc_conv3_traffic.json.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [kernel substring]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE
reports half of wide coalesced reads (guide: /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections, csv, json, sys

fetch_csv, write_csv, out = sys.argv[1:4]
needle = sys.argv[4] if len(sys.argv) > 4 else "conv3_kernel"

def per_dispatch(path, counter):
    acc = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(path)):
        if needle not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = int(r["Grid_Size"])
    return acc, meta

f, fm = per_dispatch(fetch_csv, "FETCH_SIZE")
w, wm = per_dispatch(write_csv, "WRITE_SIZE")
# the two passes run the same deterministic launch sequence: group by grid size
def by_grid(acc, meta):
    g = collections.defaultdict(list)
    for d, v in acc.items():
        g[meta[d]].append(v)
    return g
fg, wg = by_grid(f, fm), by_grid(w, wm)
shapes, tot_bytes, tot_n = [], 0.0, 0
for grid in sorted(fg):
    fa = sum(fg[grid]) / len(fg[grid])
    wa = sum(wg.get(grid, [0.0])) / max(len(wg.get(grid, [])), 1)
    n = len(fg[grid])
    hb = (2 * fa + wa) * 1024
    shapes.append({"grid_threads": grid, "dispatches": n, "fetch_size_kb_avg": round(fa, 1), "write_size_kb_avg": round(wa, 1),
                   "hbm_bytes_per_launch": int(hb)})
    tot_bytes += hb * n; tot_n += n
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --single-stream",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 FETCH_SIZE reports half of wide coalesced reads)",
           "kernel_family": needle, "avg_hbm_bytes_per_launch": int(tot_bytes / max(tot_n, 1)), "launches": tot_n, "per_shape": shapes},
          open(out, "w"), indent=1)
print(out, "avg HBM bytes/launch", int(tot_bytes / max(tot_n, 1)), "over", tot_n, "launches")
