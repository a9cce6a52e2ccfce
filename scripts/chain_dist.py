"""Per-chain sweep time (lane-0 shader clock) for C chains after W warm-up iterations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.batched import DeviceGibbsK1
chains = int(sys.argv[1]); warm = int(sys.argv[2])
w = workloads.make("cfg2")
g = DeviceGibbsK1(w["data"][0], "gaussian", w["N"], w["P"], chains, seed=1000, pool_cap=int(os.environ.get("PMDI_POOL_CAP", "0")), block_threads=int(os.environ.get("PMDI_BLOCK", "0")))
print("block", g.sw.block_threads, "lds bytes", g.sw.lds_bytes)
n_s = g.n - g.n1 + 1
for it in range(warm + 2):
    g.iteration(time_kernel=True); g.finish_timing(); st = g.check()
t = g.sw.chain_costs() / 2.4e6
ids = st[:, 0] / n_s
conv = ids < 10
print(f"two_per_cu={os.environ.get('PMDI_TWO_PER_CU','1')} chains={chains}: kernel {g.kernel_ms[-1]:.0f} ms | converged chains ({conv.sum()}): per-chain ms p10/p50/p90 "
      f"{np.percentile(t[conv],10):.0f}/{np.percentile(t[conv],50):.0f}/{np.percentile(t[conv],90):.0f} | all: mean {t.mean():.0f} max {t.max():.0f} | sum/512 slots {t.sum()/512:.0f} ms sum/256 {t.sum()/256:.0f} ms; p99 {np.percentile(t,99):.0f} heavy(>2x median) {np.mean(t > 2*np.median(t)):.1%} of chains = {t[t > 2*np.median(t)].sum()/t.sum():.0%} of work", flush=True)
heavy = ids > int(os.environ.get("PMDI_LIGHT_IDS", "40"))
print(f"groups (by this sweep's ids/step): heavy {heavy.sum()} chains, sum {t[heavy].sum()/1e3:.1f} s, mean {t[heavy].mean() if heavy.any() else 0:.0f} ms, max {t[heavy].max() if heavy.any() else 0:.0f}; light {(~heavy).sum()} chains, sum {t[~heavy].sum()/1e3:.1f} s, mean {t[~heavy].mean():.0f} ms; slot-seconds available {g.kernel_ms[-1]*512/1e3:.1f}", flush=True)
