"""Development aid: what do the steps of the slowest chains look like?  Runs the device-resident chains of a workload for some
iterations, picks the costliest chains of the last sweep (and a median one), sweeps once more, and re-runs exactly those chains
through the oracle (the checker) with its per-step record: classes, reachable clusters, chosen clusters, clones and distinct
columns per step and dataset, next to the kernel's phase timers of the same sweep.

    PMDI_PHASE_TIMERS=1 python scripts/slow_chain_steps.py [WORKLOAD] [chains] [iterations] [how many slow chains]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "HL"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
top = int(sys.argv[4]) if len(sys.argv) > 4 else 3
G.build()
pkg = G.load_package()
O = G.load_oracle()
from particlemdi_jl_amd import workloads  # noqa: E402
w = workloads.make(name)
n, N, K, P = w["n"], w["N"], w["K"], w["P"]
seed = 1000
sw = pkg.Sweeper(w["data"], w["kinds"], N, P, n_chains=C, seed=seed)
g = pkg.Gibbs(sw, rho=w["rho"])
g.iterate(iters)
import torch
torch.cuda.synchronize()
cs = sw.chain_costs()
order = np.argsort(cs)
pick = list(order[-top:][::-1]) + [order[C // 2]]
g.step(pkg.STEP_BEGIN); g.step(pkg.STEP_HYPERS)
st = {int(c): g.get(int(c)) for c in pick}
g.step(pkg.STEP_SWEEP)
res = g.results()
cs2 = sw.chain_costs() / float(sw.clock_hz)
g.step(pkg.STEP_ALIGN)
it = g.iterations
n_s = n - g.n1 + 1
names = ["prefix", "cluster", "wait1", "particle", "wait2", "ess+book", "wait3", "follow", "resample", "finish"]
for c in pick:
    c = int(c)
    s1 = st[c]
    Pi = s1["gamma"] / s1["gamma"].sum(axis=0, keepdims=True)
    flags = [s1["flags"][sum(w["D"][:k]):sum(w["D"][:k + 1])] for k in range(K)]
    orc = O.Oracle(w["data"], w["kinds"], N, P, seed=seed + c)
    rec = orc.debug_steps(n_s)
    ro = orc.sweep(it, s1["s"], s1["order"], g.n1, Pi, s1["Phi"], flags, lw_init=1.0, trace=True)
    orc.close()
    print(f"chain {c}: {cs2[c]:.3f} s this sweep; {int(ro['stats']['n_resamples'])} resampling events; allocations equal the oracle's: {bool((g.get(c)['s'] == ro['s']).all()) if False else '-'}")
    if os.environ.get("PMDI_PHASE_TIMERS"):
        ph = sw.phase_timers(c).astype(np.float64)
        print("   per observation: " + " ".join(f"{nm} {ph[i] / n_s / 1e3:.1f}k" for i, nm in enumerate(names)))
    for k in range(K):
        r = rec[:, k, :]
        def q(v):
            return f"mean {v.mean():6.2f} p50 {np.median(v):4.0f} p90 {np.percentile(v, 90):4.0f} p99 {np.percentile(v, 99):4.0f} max {v.max():4.0f}"
        gen = (r[:, 2] > 1) | (r[:, 0] > 1) | (r[:, 3] > 0)
        print(f"   dataset {k}: steps off the one-class / one-cluster / no-clone path {gen.mean() * 100:5.1f} %")
        print(f"      classes   {q(r[:, 0])}\n      reachable {q(r[:, 1])}\n      chosen    {q(r[:, 2])}\n      cloned    {q(r[:, 3])}\n      columns   {q(r[:, 4])}\n      max id    {q(r[:, 6])}")
