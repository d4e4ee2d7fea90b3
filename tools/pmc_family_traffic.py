"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of one single-stream bench step into per-family HBM traffic:

    python tools/pmc_family_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [launches-per-step divisor]

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE reports half of wide
coalesced reads (guide: /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).  Raw (uncorrected) totals are kept
next to the corrected ones.  Families are matched on kernel-name substrings."""
import collections, csv, json, sys

fetch_csv, write_csv, out = sys.argv[1:4]
FAMILIES = collections.OrderedDict([
    ("conv3x", ("conv3x_kernel",)), ("conv3xw", ("conv3xw_",)), ("conv3", ("conv3_kernel",)), ("conv3_wgrad", ("conv3_wgrad",)),
    ("pw", ("pw_conv", "pwx_kernel", "pw_wgrad", "pw_splitk", "pw_up", "pwb_kernel")), ("dw_iel", ("dw3x3", "iel_")), ("ln", ("ln_",)),
])


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            n[k] += 1
    return tot, n


f, fn = per_kernel(fetch_csv, "FETCH_SIZE")
w, wn = per_kernel(write_csv, "WRITE_SIZE")
fam = collections.OrderedDict()
for name, needles in FAMILIES.items():
    ks = [k for k in f if any(s in k for s in needles)]
    fetch, write, launches = sum(f[k] for k in ks), sum(w.get(k, 0.0) for k in ks), sum(fn[k] for k in ks)
    if launches:
        fam[name] = {"launches": launches, "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
                     "raw_bytes": int((fetch + write) * 1024), "hbm_bytes": int((2 * fetch + write) * 1024),
                     "avg_hbm_bytes_per_launch": int((2 * fetch + write) * 1024 / launches)}
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("cidnet::", "")
    return k.split("(")[0][:90]


top = sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0.0)))[:30]
kernels = collections.OrderedDict((short(k), {"launches": fn[k], "avg_hbm_bytes_per_launch": int((2 * f[k] + w.get(k, 0.0)) * 1024 / max(fn[k], 1)),
                                              "avg_fetch_kib": round(f[k] / max(fn[k], 1), 1), "avg_write_kib": round(w.get(k, 0.0) / max(fn[k], 1), 1)})
                                  for k in top)
allf, allw = sum(f.values()), sum(w.values())
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-inference-leg --single-stream (all launches of the process: set-up steps, 1 warm-up, 1 timed, 2 instrumented)",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 FETCH_SIZE reports half of wide coalesced reads); raw_bytes = (FETCH + WRITE) * 1024",
           "families": fam,
           "top_kernels_by_hbm_bytes": kernels,
           "all_kernels": {"fetch_kib": round(allf, 1), "write_kib": round(allw, 1), "raw_bytes": int((allf + allw) * 1024),
                           "hbm_bytes": int((2 * allf + allw) * 1024), "dispatches": sum(fn.values())}},
          open(out, "w"), indent=1)
for k, v in fam.items():
    print(f"{k:12s} launches {v['launches']:6d}  avg HBM bytes/launch {v['avg_hbm_bytes_per_launch']:>12d}")
