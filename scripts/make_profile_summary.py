"""Turns the rocprofv3 CSVs of a profiling run of bench.py into the committed summaries under profiles/<tag>/ and the
entry of profiles/hbm_traffic.json that bench.py reads for roofline.traffic (keyed by the exact operating point).

    python scripts/make_profile_summary.py TAG KEY STEPS TRACE_DIR [COUNTER=DIR ...]

TRACE_DIR: output of `rocprofv3 --kernel-trace --stats --output-format csv -d TRACE_DIR -- python3 bench.py ...`
COUNTER=DIR: output of a `--pmc COUNTER[,COUNTER...]` pass of the same command (separate passes: FETCH_SIZE and
WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").

One sweep of all chains is a SET of up to three concurrent pmdi_sweep_kernel launches (heaviest chains / heavy /
light: DESIGN.md section 4.4).  A set's duration is the interval from the first start to the last end of its launches,
which is what bench.py's HIP events bracket; counters are summed over the launches of a set.  Only the last STEPS
sets (the timed region) enter the averages.
"""
import csv, glob, json, os, shutil, sys
import numpy as np

tag, key, steps, d_trace = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
passes = dict(a.split("=", 1) for a in sys.argv[5:])
# bench.py's latency-form leg (K > 1) launches the one-dataset-per-workgroup builds `<T, WPS, true>` after the timed region:
# PMDI_KERNEL_FILTER="false>" keeps the throughput-form launches only
FLT = os.environ.get("PMDI_KERNEL_FILTER", "")
out = os.path.join("profiles", tag); os.makedirs(out, exist_ok=True)


def is_sweep(name):
    return "pmdi_sweep2_kernel" in name or ("pmdi_sweep_kernel" in name and FLT in name)


def f1(d, *pats):
    for pat in pats:
        hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
        if hits:
            return hits[0]
    raise FileNotFoundError(f"{d}: none of {pats}")


def sweep_sets(d):
    rows = [r for r in csv.DictReader(open(f1(d, "*kernel_trace.csv"))) if is_sweep(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sets, cur = [], []
    for r in rows:      # a sweep launches every build at most once (round 3: heaviest / heavy / settled-chain kernel / re-run of the
        # chains that kernel gave back, the last one behind the settled launch): a kernel name seen again opens the next set
        if cur and any(x["Kernel_Name"] == r["Kernel_Name"] for x in cur):
            sets.append(cur); cur = []
        cur.append(r)
    if cur:
        sets.append(cur)
    return rows, sets


try:
    shutil.copy(f1(d_trace, "*kernel_stats.csv", "*kernels_summary.csv"), os.path.join(out, "kernel_stats.csv"))
except FileNotFoundError:
    pass
rows, sets = sweep_sets(d_trace)
set_ms = np.array([(max(int(x["End_Timestamp"]) for x in s) - min(int(x["Start_Timestamp"]) for x in s)) / 1e6 for s in sets])
timed = set_ms[-steps:]
per_kernel = {}
for s in sets[-steps:]:
    for x in s:
        per_kernel.setdefault(x["Kernel_Name"], []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6)


def counters(d):
    """{counter: mean over the timed sets of the sum over the set's dispatches}; under counter collection the launches of a
    set run one after the other, so sets are rebuilt from that pass's own kernel trace by dispatch order."""
    rs = [r for r in csv.DictReader(open(f1(d, "*counter_collection.csv", "*counter_collection_trace.csv"))) if is_sweep(r["Kernel_Name"])]
    per = {}
    for r in rs:
        per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    n_per_set = max(1, len(rows) // len(sets))
    res = {}
    for name, disp in per.items():
        vals = [disp[k] for k in sorted(disp)]
        per_set = [sum(vals[i:i + n_per_set]) for i in range(0, len(vals) - n_per_set + 1, n_per_set)]
        res[name] = float(np.mean(per_set[-steps:]))
    return res


pmc = {}
for name, d in passes.items():
    pmc.update(counters(d))
summary = {
    "operating_point": key, "launches_per_sweep": len(rows) // max(len(sets), 1), "sweeps_total": len(sets), "timed_sweeps": steps,
    "avg_ms_timed_sweeps": float(timed.mean()), "min_ms_timed": float(timed.min()), "max_ms_timed": float(timed.max()),
    "kernels_of_a_sweep": [{"kernel": k, "avg_ms_timed": float(np.mean(v)), "launches_timed": len(v),
                            "workgroup": next(x["Workgroup_Size_X"] for x in sets[-1] if x["Kernel_Name"] == k),
                            "grid_workgroups": next(int(x["Grid_Size_X"]) // int(x["Workgroup_Size_X"]) for x in sets[-1] if x["Kernel_Name"] == k),
                            "vgpr": next(x.get("VGPR_Count") or x.get("Vgpr_Count") for x in sets[-1] if x["Kernel_Name"] == k),
                            "scratch": next(x.get("Scratch_Size") for x in sets[-1] if x["Kernel_Name"] == k),
                            "lds": next(x.get("LDS_Block_Size") or x.get("Lds_Block_Size") for x in sets[-1] if x["Kernel_Name"] == k)}
                           for k, v in per_kernel.items()],
    "pmc_per_timed_sweep": pmc,
    "per_sweep_ms": [round(float(x), 2) for x in set_ms],
}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, i.e. it
    # reports 1/2 of the bytes of wide streaming reads -> doubled.  (This kernel's reads are 4-16 B gathers, for which the guide
    # has no calibration; the raw sum is kept alongside.)
    raw = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    corrected = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
    summary["hbm_bytes_per_sweep_raw"] = raw
    summary["hbm_bytes_per_sweep"] = corrected
    tpath = os.path.join("profiles", "hbm_traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if "workload" in tj:       # round-1 format
        tj = {}
    tj[key] = {"hbm_bytes_per_sweep": corrected, "hbm_bytes_per_sweep_raw": raw, "fetch_kb": pmc["FETCH_SIZE"], "write_kb": pmc["WRITE_SIZE"],
               "sweep_ms_under_trace": float(timed.mean()),
               "source": f"profiles/{tag}/sweep_kernel_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of this exact "
                         "command; mean over the timed sweeps, summed over the concurrent launches of a sweep; gfx950 correction: FETCH_SIZE x 2)"}
    json.dump(tj, open(tpath, "w"), indent=1)
json.dump(summary, open(os.path.join(out, "sweep_kernel_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_sweep_ms"}, indent=1))
