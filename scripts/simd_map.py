"""Where does wave 0 of every chain run?  (HW_ID / XCC_ID recorded by the sweep kernel in phase mode.)
Counts co-resident chain pairs (same XCC/SE/CU) whose wave 0 shares a SIMD."""
import os, sys
os.environ["PMDI_PHASE_TIMERS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.batched import DeviceGibbsK1
chains = int(sys.argv[1]); warm = int(sys.argv[2])
w = workloads.make("cfg2")
g = DeviceGibbsK1(w["data"][0], "gaussian", w["N"], w["P"], chains, seed=1000)
for it in range(warm + 1):
    g.iteration(); g.check()
hw = np.array([g.sw.phase_timers(c)[12:14] for c in range(chains)]).astype(np.int64)
hid, xcc = hw[:, 0], hw[:, 1] & 0xF
wave_id, simd, cu, sh_, se = hid & 15, (hid >> 4) & 3, (hid >> 8) & 15, (hid >> 12) & 1, (hid >> 13) & 7
loc = xcc * 4096 + se * 256 + sh_ * 16 + cu
print("distinct CUs used:", len(np.unique(loc)), "chains:", chains)
print("wave-0 simd histogram:", np.bincount(simd, minlength=4))
t = g.sw.chain_costs() / 2.4e6
print("per-chain ms p50 by simd:", [round(float(np.median(t[simd == q])), 1) if (simd == q).any() else None for q in range(4)])
for b in range(min(chains, 12)):
    print(f"block {b}: xcc {xcc[b]} se {se[b]} sh {sh_[b]} cu {cu[b]} simd {simd[b]} wave_id {wave_id[b]}")
ids = g.check()[:, 0] / (g.n - g.n1 + 1)
from collections import defaultdict
byloc = defaultdict(list)
for c in range(chains): byloc[int(loc[c])].append(c)
same, diff = [], []
for l, cs in byloc.items():
    if len(cs) == 2:
        a_, b_ = cs
        if ids[a_] < 10 and ids[b_] < 10:
            (same if simd[a_] == simd[b_] else diff).append((t[a_] + t[b_]) / 2)
print(f"co-resident converged pairs: same wave-0 simd {len(same)} (mean {np.mean(same) if same else 0:.1f} ms), different {len(diff)} (mean {np.mean(diff) if diff else 0:.1f} ms)")
