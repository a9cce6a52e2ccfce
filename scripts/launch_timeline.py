#!/usr/bin/env python3
"""Launch timeline of the sweep kernels from a rocprofv3 --kernel-trace CSV: start, end and duration (ms) of every pmdi_* launch
of the last `--iters` Gibbs iterations of the widest run in the trace.  Shows what bounds the makespan of a sweep (which launch
group ends last, how long the re-run of given-back chains takes).  Usage: launch_timeline.py <dir or csv> [--iters 2]"""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 2
    f = src if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "pmdi" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    wide = max(int(r["Grid_Size_X"]) for r in rows if "sweep2" in r["Kernel_Name"]) if any("sweep2" in r["Kernel_Name"] for r in rows) else None
    # iterations of the widest run: split at the hypers kernel
    idx = [i for i, r in enumerate(rows) if "sweep2" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == wide] if wide else []
    if not idx:
        idx = [i for i, r in enumerate(rows) if "sweep_kernel" in r["Kernel_Name"]]
    first = idx[-iters] if len(idx) >= iters else idx[0]
    lo = max(0, first - 6)
    hi = min(len(rows), idx[-1] + 8)
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        name = name[name.find("pmdi"):][:64]
        print("%10.3f %10.3f %9.3f  wgs %6d x %4s  lds %6s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6,
              int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""), name))


if __name__ == "__main__":
    main()
