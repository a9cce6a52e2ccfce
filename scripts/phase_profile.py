"""Per-phase shader-clock totals (lane 0 of the chain's workgroup 0) of the last sweep of a few chains.
usage: PMDI_PHASE_TIMERS=1 python scripts/phase_profile.py WORKLOAD CHAINS ITERS [SCALE]"""
import os, sys
import numpy as np
os.environ.setdefault("PMDI_PHASE_TIMERS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
G.build(); pkg = G.load_package()
from particlemdi_jl_amd import workloads
name, C, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = workloads.make(name, float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=41)
g = pkg.Gibbs(sw, rho=0.25, feature_select=(name == "cfg5"))
g.iterate(iters); st = g.results()["stats"]
ns = (w["n"] - g.n1 + 1)
names = ["0 prefix", "1 looptop/needlist", "2 terms", "3 ordered sum", "4 cdf", "5 draw+vote", "6 keys/firsts", "7 ranks", "8 apply+stats", "9 phi/handoff/ess",
         "10 resample", "11 final", "12 slow:front", "13 stats update", "14 total cycles", "15 wall ticks"]
for c in range(min(C, 4)):
    ph = sw.phase_timers(c)
    print(f"chain {c}: resamples {st[c,1]} ids/step {st[c,0]/(ns*w['K']):.1f} total {ph[14]/1e6:.1f} Mcycles = {ph[14]/sw.clock_hz*1e3:.1f} ms (wall {ph[15]/1e5:.1f} ms)")
    for i in range(14):
        if ph[i]:
            print(f"   {names[i]:22s} {ph[i]/1e6:9.2f} Mcyc  {100.0*ph[i]/ph[14]:5.1f} %   {ph[i]/ns:9.0f} cyc/obs")
