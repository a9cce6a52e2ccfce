"""Re-derive the `roofline.traffic` fields of a saved bench.py line from profiles/hbm_traffic.json.

The PMC passes that measure HBM traffic run AFTER the plain bench run of the same build (they are separate rocprofv3
passes), so a bench line saved before them carries the traffic entry of the previous build.  bench.py itself reads
profiles/hbm_traffic.json at run time; this script applies the same lookup to a saved line.  Nothing else is touched.

    python scripts/refresh_bench_line.py < raw_line.json > profiles/rNN/bench_default.json
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
c = d["config"]
key = f"HL|chains={c['chains_per_gpu']}|burnin={c['burnin_iterations']}|warmup={d['warmup']}|steps={d['steps']}|scale=1.0"
assert c["workload"].startswith("HL"), "only the default workload has a committed traffic entry"
t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))[key]
r = d["roofline"]
r["traffic"] = t["hbm_bytes_per_sweep"]
r["traffic_over_algorithmic"] = t["hbm_bytes_per_sweep"] / r["algorithmic_bytes_per_sweep"]
r["traffic_note"] = "traffic re-derived from profiles/hbm_traffic.json after this build's PMC passes (scripts/refresh_bench_line.py)"
print(json.dumps(d))
