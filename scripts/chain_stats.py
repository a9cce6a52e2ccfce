"""Per-iteration sweep counters of a few chains of a BASELINE workload on the device-resident driver.
usage: python scripts/chain_stats.py WORKLOAD CHAINS ITERS"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
G.build(); pkg = G.load_package()
from particlemdi_jl_amd import workloads
name, C, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = workloads.make(name)
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=int(os.environ.get("SEED", 41)))
g = pkg.Gibbs(sw, rho=0.25, feature_select=(name == "cfg5"))
ns = (w["n"] - g.n1 + 1) * w["K"]
for it in range(1, iters + 1):
    t0 = time.perf_counter(); g.iterate(1); st = g.results()["stats"]; dt = time.perf_counter() - t0
    costs = sw.chain_costs() / sw.clock_hz
    print(f"it {it}: {dt*1e3:8.1f} ms  ids/step {np.round(st[:,0]/ns,1).tolist()} resamples {st[:,1].tolist()} clones {st[:,2].tolist()} "
          f"chain s {np.round(costs,3).tolist()} labels/dataset {[int(len(np.unique(g.get(c)['s'][:,0]))) for c in range(min(C,8))]}", flush=True)
