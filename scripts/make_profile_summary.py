"""Turns the rocprofv3 CSVs of a round into the committed summaries under profiles/rNN/ and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

    python scripts/make_profile_summary.py r01 gpurun_out/r1_stats gpurun_out/r1_fetch gpurun_out/r1_write <steps>
"""
import csv, glob, json, os, shutil, sys
import numpy as np
tag, d_stats, d_fetch, d_write, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
out = os.path.join("profiles", tag); os.makedirs(out, exist_ok=True)
f1 = lambda d, pat: glob.glob(os.path.join(d, "**", pat), recursive=True)[0]
shutil.copy(f1(d_stats, "*kernel_stats.csv"), os.path.join(out, "kernel_stats.csv"))
rows = [r for r in csv.DictReader(open(f1(d_stats, "*kernel_trace.csv"))) if "pmdi_sweep_kernel" in r["Kernel_Name"]]
dur = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows])
timed = dur[-steps:]
def counter(d, name):
    rs = [r for r in csv.DictReader(open(f1(d, "*counter_collection.csv"))) if "pmdi_sweep_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return np.array([float(r["Counter_Value"]) for r in rs])
fetch, write = counter(d_fetch, "FETCH_SIZE"), counter(d_write, "WRITE_SIZE")
fetch_t, write_t = fetch[-steps:].mean(), write[-steps:].mean()
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes
# of WIDE (16 B/lane) streaming reads.  This kernel's reads are 4-16 B gathers, uncalibrated for that correction,
# so both the raw and the doubled figure are recorded; bench.py reports the raw sum.
hbm_raw = (fetch_t + write_t) * 1024.0
hbm_doubled = (2 * fetch_t + write_t) * 1024.0
row0 = rows[-1]
summary = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu   (defaults: 2048 chains, 30 warm-up + 10 timed launches)",
    "kernel": row0["Kernel_Name"], "launches_total": len(dur), "timed_launches": steps,
    "avg_ms_all_launches": float(dur.mean()), "avg_ms_timed_launches": float(timed.mean()),
    "min_ms_timed": float(timed.min()), "max_ms_timed": float(timed.max()),
    "grid": row0.get("Grid_Size_X"), "workgroup": row0.get("Workgroup_Size_X"), "lds_bytes": row0.get("LDS_Block_Size"),
    "vgpr": row0.get("VGPR_Count"), "sgpr": row0.get("SGPR_Count"), "scratch": row0.get("Scratch_Size"),
    "FETCH_SIZE_KB_per_timed_launch": float(fetch_t), "WRITE_SIZE_KB_per_timed_launch": float(write_t),
    "hbm_bytes_per_launch_raw": hbm_raw, "hbm_bytes_per_launch_fetch_doubled": hbm_doubled,
    "per_launch_ms": [round(float(x), 2) for x in dur],
}
json.dump(summary, open(os.path.join(out, "sweep_kernel_summary.json"), "w"), indent=1)
json.dump({"workload": "cfg2", "chains_per_gpu": int(int(row0["Grid_Size_X"]) / int(row0["Workgroup_Size_X"])),
           "hbm_bytes_per_launch": hbm_raw, "hbm_bytes_per_launch_fetch_doubled": hbm_doubled,
           "source": f"profiles/{tag}/sweep_kernel_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, timed launches)"},
          open(os.path.join("profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_launch_ms"}, indent=1))
