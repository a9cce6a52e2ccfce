"""Instruction mix and issue / wait shares of the settled-chain kernel from two rocprofv3 --pmc passes of scripts/s2_probe.py (GPU box).

    python scripts/pmc_mix.py WORKLOAD CHAINS OUT.json

Pass A: instruction counts; pass B: pipe activity.  The pmdi_sweep2_kernel dispatch of the LAST sweep is taken; counts are per wave
and per swept observation.  This script only starts child processes (rocprofv3 ... -- python3 scripts/s2_probe.py ...)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, chains, out = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
A = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
B = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
prof = os.path.join(ROOT, "gpurun_out", "prof")
os.makedirs(prof, exist_ok=True)
raw = {}
for tag, ctrs in (("A", A), ("B", B)):
    d = os.path.join(prof, f"mix_{name}_{tag}")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--",
                                                              "python3", os.path.join(ROOT, "scripts", "s2_probe.py"), name, chains, "12", "2"]
    print("+", " ".join(cmd), flush=True)
    with open(os.path.join(prof, f"mix_{name}_{tag}.log"), "w") as lg:
        rc = subprocess.call(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=lg, stderr=subprocess.STDOUT)
    if rc:
        sys.exit(open(os.path.join(prof, f"mix_{name}_{tag}.log")).read()[-1500:])
    f = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "pmdi_sweep2_kernel" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    acc = {}
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    raw[tag] = acc
    kname = rows[-1]["Kernel_Name"]
import __graft_entry__ as G  # noqa: E402
G.load_package()
from particlemdi_jl_amd import workloads  # noqa: E402
w = workloads.make(name)
n_s = w["n"] - int(w["rho"] * w["n"]) + 1
waves = raw["A"]["SQ_WAVES"]
res = {"note": f"rocprofv3 --pmc on `python3 scripts/s2_probe.py {name} {chains} 12 2`, the {kname} dispatch of the last sweep; two separate passes "
               f"(instruction counts / pipe activity); per wave and per swept observation ({n_s} observations, all {w['K']} datasets of an observation)",
       "waves": waves,
       "instructions_per_wave_and_observation": {k: round(v / waves / n_s, 1) for k, v in raw["A"].items() if k != "SQ_WAVES"},
       "cycles_per_wave_and_observation": {k: round(v / waves / n_s, 1) for k, v in raw["B"].items() if k != "SQ_BUSY_CYCLES"},
       "fractions_of_wave_cycles": {k: round(v / raw["B"]["SQ_WAVE_CYCLES"], 3) for k, v in raw["B"].items() if k not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")},
       "raw": raw}
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "raw"}, indent=1))
