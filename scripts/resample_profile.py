"""Where a resampling event spends its cycles (A/B build with -DPMDI_RESAMPLE_TIMERS; see profiles/README.md).
usage: PMDI_LIB_PATH=build_ab/libpmdi_rs.so PMDI_EXTRA_HIPCC_FLAGS=-DPMDI_RESAMPLE_TIMERS python scripts/resample_profile.py WORKLOAD CHAINS ITERS"""
import os, sys
import numpy as np
os.environ.setdefault("PMDI_PHASE_TIMERS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); pkg.build()
from particlemdi_jl_amd import workloads
name, C, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = workloads.make(name)
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=41)
g = pkg.Gibbs(sw, rho=0.25, feature_select=(name == "cfg5"))
g.iterate(iters); st = g.results()["stats"]
names = ["weights", "cumsum + u sequence", "search + ancestors", "(dataset loop top)", "gather + occupancy", "id scan + counts", "relabel", "table reset", "statistics moves", "class rebuild"]
tot = np.zeros(10); ev = 0
for c in range(C):
    if st[c, 1] > 0:
        tot += sw.phase_timers(c)[:10]; ev += st[c, 1]
print(f"{name}: {C} chains, iteration {iters}: {ev} resampling events in all (split={sw.split}); shader cycles per event (all datasets of the workgroup):")
for i in range(10):
    print(f"   {names[i]:24s} {tot[i]/max(ev,1):10.0f}")
print(f"   {'total':24s} {tot.sum()/max(ev,1):10.0f}")
