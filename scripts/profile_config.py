"""Profiling recipe of one bench.py operating point, for the GPU box: three rocprofv3 passes of the same command line -- kernel trace
+ stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes: the two counters do not fit one on gfx950, MI355X_MICROARCH.md) -- then
scripts/make_profile_summary.py, which writes profiles/<TAG>/{kernel_stats.csv, sweep_kernel_summary.json} and the keyed entry of
profiles/hbm_traffic.json that bench.py reads for roofline.traffic.  Raw output goes to gpurun_out/prof/ (scratch).

    python scripts/profile_config.py WORKLOAD TAG [--steps K] [--warmup W] [--no-pmc]

This script only starts child processes (rocprofv3 ... -- python3 bench.py ...); it never touches the GPU itself.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("tag")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--no-pmc", action="store_true")
a = ap.parse_args()
prof = os.path.join(ROOT, "gpurun_out", "prof")
os.makedirs(prof, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
bench = ["python3", os.path.join(ROOT, "bench.py"), "--workload", a.workload, "--steps", str(a.steps), "--warmup", str(a.warmup),
         "--no-cpu", "--no-parity", "--no-latency-form"]


def run(name, extra):
    d = os.path.join(prof, f"{a.workload}_{name}")
    cmd = ["rocprofv3", "--kernel-trace"] + extra + ["--output-format", "csv", "-d", d, "--"] + bench
    print("+", " ".join(cmd), flush=True)
    with open(os.path.join(prof, f"{a.workload}_{name}.json"), "w") as out, open(os.path.join(prof, f"{a.workload}_{name}.err"), "w") as err:
        rc = subprocess.call(cmd, cwd="/tmp", env=env, stdout=out, stderr=err)
    if rc != 0:
        print(open(os.path.join(prof, f"{a.workload}_{name}.err")).read()[-2000:])
        sys.exit(f"{name} pass failed ({rc})")
    return d


d_trace = run("trace", ["--stats"])
line = json.loads(open(os.path.join(prof, f"{a.workload}_trace.json")).read().strip().splitlines()[-1])
cfg = line["config"]
key = f"{a.workload}|chains={cfg['chains_per_gpu']}|burnin={cfg['burnin_iterations']}|warmup={a.warmup}|steps={a.steps}|scale=1.0"
passes = []
if not a.no_pmc:
    passes = [f"FETCH_SIZE={run('fetch', ['--pmc', 'FETCH_SIZE'])}", f"WRITE_SIZE={run('write', ['--pmc', 'WRITE_SIZE'])}"]
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_profile_summary.py"), a.tag, key, str(a.steps), d_trace] + passes, cwd=ROOT)
os.makedirs(os.path.join(ROOT, "profiles", a.tag), exist_ok=True)
json.dump(line, open(os.path.join(ROOT, "profiles", a.tag, "bench_under_trace.json"), "w"), indent=1)
print(f"{a.workload}: {line['value']:.1f} it/s under the kernel trace; key {key}")
# what came out of this run travels back through gpurun_out/ (the only directory a GPU box returns); the raw counter CSVs stay behind
import shutil
back = os.path.join(ROOT, "gpurun_out", "profiles_out")
os.makedirs(back, exist_ok=True)
shutil.copytree(os.path.join(ROOT, "profiles", a.tag), os.path.join(back, a.tag), dirs_exist_ok=True)
shutil.copy(os.path.join(ROOT, "profiles", "hbm_traffic.json"), os.path.join(back, "hbm_traffic.json"))
for name in ("trace", "fetch", "write"):
    shutil.rmtree(os.path.join(prof, f"{a.workload}_{name}"), ignore_errors=True)
