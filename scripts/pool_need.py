"""Profiling aid: how many cluster ids per dataset does a chain really use?  The reference sizes its pool for N*P+1 clusters
(src/pmdi.jl:140) and so does pmdi_create by default; the largest id a sweep touches (pmdi_sweep_stats.max_id) says what a
smaller pool_cap would have to hold.

    python scripts/pool_need.py WORKLOAD [chains] [iterations]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "HL"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
G.build()
pkg = G.load_package()
from particlemdi_jl_amd import workloads  # noqa: E402
w = workloads.make(name)
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=4242)
g = pkg.Gibbs(sw, rho=w["rho"], feature_select=(name == "cfg5"))
cap = w["N"] * w["P"] + 1
print(f"{name}: N*P+1 = {cap} ids per dataset, {C} chains", flush=True)
for it in range(iters):
    g.iterate(1)
    mx = g.results()["stats"][:, 3]
    print(f"iteration {it + 1}: max id over the sweep: median {int(np.median(mx))}, p99 {int(np.percentile(mx, 99))}, max {int(mx.max())} = {mx.max() / cap:.3f} of N*P+1", flush=True)
g.close(); sw.close()
