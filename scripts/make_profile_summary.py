"""Turns the rocprofv3 CSVs of a round into the committed summaries under profiles/rNN/ and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

    python scripts/make_profile_summary.py r01 gpurun_out/r1_stats gpurun_out/r1_fetch gpurun_out/r1_write <steps>

One sweep of all chains is a SET of up to three concurrent pmdi_sweep_kernel launches (heaviest
chains / heavy / light: DESIGN.md section 4.4).  A set's duration is the interval from the first
start to the last end of its launches, which is what bench.py's HIP events bracket; counters are
summed over the launches of a set.
"""
import csv, glob, json, os, shutil, sys
import numpy as np
tag, d_stats, d_fetch, d_write, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
out = os.path.join("profiles", tag); os.makedirs(out, exist_ok=True)
def f1(d, *pats):
    for pat in pats:
        hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
        if hits:
            return hits[0]
    raise FileNotFoundError(f"{d}: none of {pats}")


# rocprofv3 of ROCm 7.2 writes a rocpd database; `rocpd2csv` / `rocpd2summary --format csv` turn it into these files
shutil.copy(f1(d_stats, "*kernel_stats.csv", "*kernels_summary.csv"), os.path.join(out, "kernel_stats.csv"))
rows = [r for r in csv.DictReader(open(f1(d_stats, "*kernel_trace.csv"))) if "pmdi_sweep_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a set = launches that overlap or start within 2 ms of the running set's start (the three launches of a sweep are
# issued back to back; consecutive sweeps are separated by the host's hyper-parameter update)
sets, cur = [], []
for r in rows:
    a = int(r["Start_Timestamp"])
    if cur and a > max(int(x["End_Timestamp"]) for x in cur) and a - int(cur[0]["Start_Timestamp"]) > 2_000_000:
        sets.append(cur); cur = []
    cur.append(r)
if cur: sets.append(cur)
set_ms = np.array([(max(int(x["End_Timestamp"]) for x in s) - min(int(x["Start_Timestamp"]) for x in s)) / 1e6 for s in sets])
timed = set_ms[-steps:]
per_kernel = {}
for s in sets[-steps:]:
    for x in s:
        k = x["Kernel_Name"]
        per_kernel.setdefault(k, []).append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6)


def counter(d, name):
    rs = [r for r in csv.DictReader(open(f1(d, "*counter_collection.csv", "*counter_collection_trace.csv"))) if "pmdi_sweep_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    per_dispatch = {}
    for r in rs:
        per_dispatch[int(r["Dispatch_Id"])] = per_dispatch.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    vals = [per_dispatch[k] for k in sorted(per_dispatch)]
    per_set = len(rows) // len(sets)           # launches per sweep (3 with the chain split, else 1)
    return np.array([sum(vals[i:i + per_set]) for i in range(0, len(vals) - per_set + 1, per_set)])


fetch, write = counter(d_fetch, "FETCH_SIZE"), counter(d_write, "WRITE_SIZE")
fetch_t, write_t = fetch[-steps:].mean(), write[-steps:].mean()
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes
# of WIDE (16 B/lane) streaming reads.  This kernel's reads are 4-16 B gathers, uncalibrated for that correction,
# so both the raw and the doubled figure are recorded; bench.py reports the raw sum.
hbm_raw = (fetch_t + write_t) * 1024.0
hbm_doubled = (2 * fetch_t + write_t) * 1024.0
chains = max(int(x["Grid_Size_X"]) // int(x["Workgroup_Size_X"]) for x in sets[-1])
summary = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu   (defaults: 2048 chains, 30 warm-up + 10 timed sweeps)",
    "launches_per_sweep": len(rows) // len(sets), "sweeps_total": len(sets), "timed_sweeps": steps,
    "avg_ms_timed_sweeps": float(timed.mean()), "min_ms_timed": float(timed.min()), "max_ms_timed": float(timed.max()),
    "kernels_of_a_sweep": [{"kernel": k, "avg_ms_timed": float(np.mean(v)), "workgroup": next(x["Workgroup_Size_X"] for x in sets[-1] if x["Kernel_Name"] == k),
                            "grid_workgroups": next(int(x["Grid_Size_X"]) // int(x["Workgroup_Size_X"]) for x in sets[-1] if x["Kernel_Name"] == k),
                            "vgpr": next(x.get("VGPR_Count") or x.get("Vgpr_Count") for x in sets[-1] if x["Kernel_Name"] == k),
                            "scratch": next(x.get("Scratch_Size") for x in sets[-1] if x["Kernel_Name"] == k)}
                           for k, v in per_kernel.items()],
    "FETCH_SIZE_KB_per_timed_sweep": float(fetch_t), "WRITE_SIZE_KB_per_timed_sweep": float(write_t),
    "hbm_bytes_per_launch_raw": hbm_raw, "hbm_bytes_per_launch_fetch_doubled": hbm_doubled,
    "per_sweep_ms": [round(float(x), 2) for x in set_ms],
}
json.dump(summary, open(os.path.join(out, "sweep_kernel_summary.json"), "w"), indent=1)
json.dump({"workload": "cfg2", "chains_per_gpu": chains, "groups": 1,
           "hbm_bytes_per_launch": hbm_raw, "hbm_bytes_per_launch_fetch_doubled": hbm_doubled,
           "source": f"profiles/{tag}/sweep_kernel_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, timed sweeps; "
                     "summed over the concurrent launches of a sweep)"},
          open(os.path.join("profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_sweep_ms"}, indent=1))
