"""Per-chain sweep time distribution of a C-chain batch after ITERS iterations: light/heavy split, slot efficiency.
usage: python scripts/chain_dist.py WORKLOAD CHAINS ITERS"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
G.build(); pkg = G.load_package()
import torch
from particlemdi_jl_amd import workloads
name, C, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = workloads.make(name)
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=1000)
g = pkg.Gibbs(sw, rho=0.25, feature_select=(name == "cfg5"))
ns = (w["n"] - g.n1 + 1) * w["K"]
g.iterate(iters - 1); g.results()
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.step(pkg.STEP_BEGIN); g.step(pkg.STEP_HYPERS); e0.record(st); g.step(pkg.STEP_SWEEP); e1.record(st); g.step(pkg.STEP_ALIGN)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
stats = g.results()["stats"]; cost = sw.chain_costs() / sw.clock_hz * 1e3
ids = stats[:, 0] / ns
heavy = ids > 40
print(f"{name}: {C} chains, sweep {iters}: {ms:.1f} ms; split={sw.split} T={sw.block_threads} lds={sw.lds_bytes}")
print(f"  light chains {int((~heavy).sum())}: ms p10/p50/p90/max {np.percentile(cost[~heavy],[10,50,90,100]).round(1).tolist()}  sum {cost[~heavy].sum()/1e3:.1f} s")
if heavy.any():
    print(f"  heavy chains {int(heavy.sum())}: ms p10/p50/p90/max {np.percentile(cost[heavy],[10,50,90,100]).round(1).tolist()}  sum {cost[heavy].sum()/1e3:.1f} s; ids/step p50/max {np.percentile(ids[heavy],[50,100]).round(0).tolist()}")
print(f"  sum of chain times {cost.sum()/1e3:.1f} s = {cost.sum()/ms/512:.2f} of 512 slots x sweep; steps fast/converted/fallback {stats[:,5].sum()}/{stats[:,6].sum()}/{stats[:,7].sum()}")
