"""Summarise the SQ counter pass of tools/pmc_mfma.sh: MFMA-busy share of each kernel family's run time.

    python tools/pmc_mfma.py <calibration dir> <step dir> <out.json>

SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs/XCDs in units the tool does not document, so it is calibrated:
the pure v_mfma_f32_16x16x4_f32 loop of tools/mfma_peak.hip (155 of 157.3 TFLOP/s = 0.987 busy) gives the counter value
per nanosecond of kernel time at full rate; a kernel's busy share is its own counter per nanosecond over that."""
import collections, csv, glob, json, os, re, sys

def load(d):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    for p in kt:
        for r in csv.DictReader(open(p)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for p in cc:
        for r in csv.DictReader(open(p)):
            vals[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
            if r["Dispatch_Id"] not in dur and "Start_Timestamp" in r:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    return vals, names, dur

def family(n):
    n = re.sub(r"cidnet::\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].split("<")[0]

cal_dir, step_dir, out = sys.argv[1:4]
cv, cn, cd = load(cal_dir)
# calibration: the longest kernels of the pure-MFMA tool
rates = sorted(((cv[d]["SQ_VALU_MFMA_BUSY_CYCLES"] / cd[d][0], cd[d][0]) for d in cv if d in cd and cd[d][0] > 0), reverse=True)
full = rates[0][0] / 0.987 if rates else None
sv, sn, sd = load(step_dir)
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for d, c in sv.items():
    if d not in sd: continue
    f = family(sn[d])
    fam[f]["ns"] += sd[d][0]; fam[f]["launches"] += 1
    for k, v in c.items(): fam[f][k] += v
rows = []
tot_ns = sum(f["ns"] for f in fam.values())
for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
    if f["ns"] < 0.002 * tot_ns: continue
    row = {"kernel": name, "launches": int(f["launches"]), "ms": round(f["ns"] / 1e6, 3), "share_of_kernel_time": round(f["ns"] / tot_ns, 4)}
    if full: row["mfma_busy"] = round(f["SQ_VALU_MFMA_BUSY_CYCLES"] / f["ns"] / full, 4)
    if f.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in f: row[k.lower() + "_per_wave_cycle"] = round(f[k] / f["SQ_WAVE_CYCLES"], 4)
    rows.append(row)
json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ counters> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --single-stream "
                     "(tools/pmc_mfma.sh); counter pass only, no --stats",
           "calibration": {"kernel": "tools/mfma_peak.hip (pure v_mfma_f32_16x16x4_f32, 0.987 of peak)",
                           "mfma_busy_counter_per_ns_at_full_rate": full, "top_rates": [round(r[0], 3) for r in rates[:4]]},
           "note": "kernel times under counter collection are longer than in the timed bench; shares are within this pass",
           "kernels": rows}, open(out, "w"), indent=1)
for r in rows[:14]: print(r)
