"""Distribution of per-chain sweep time (lane-0 shader clock) vs regime, over C chains."""
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
n_s = g.n - g.n1 + 1
for it in range(warm + 2):
    g.iteration(); st = g.check()
    if it in (warm - 1, warm + 1) or it % 10 == 9:
        t = np.array([g.sw.phase_timers(c)[14] for c in range(chains)]) / 2.4e6     # ms at 2.4 GHz
        ids = st[:, 0] / n_s
        o = np.argsort(t)
        q = lambda a, f: a[o][int(f * (chains - 1))]
        print(f"it {it}: time ms p10/p50/p90/p99/max {q(t,.1):.0f}/{q(t,.5):.0f}/{q(t,.9):.0f}/{q(t,.99):.0f}/{t.max():.0f} mean {t.mean():.0f} | "
              f"ids/step at those {q(ids,.1):.0f}/{q(ids,.5):.0f}/{q(ids,.9):.0f}/{q(ids,.99):.0f}/{ids[o][-1]:.0f} | clones p50/p90/max {np.percentile(st[:,2],50):.0f}/{np.percentile(st[:,2],90):.0f}/{st[:,2].max()} "
              f"| resamples mean {st[:,1].mean():.2f} max {st[:,1].max()} | slow-steps mean {st[:,7].mean():.0f} conv {st[:,6].mean():.0f} | nclust p50 {np.median([len(np.unique(r)) for r in g.s.cpu().numpy()[:, :]]):.0f}", flush=True)
        share = np.cumsum(np.sort(t)[::-1]) / t.sum()
        print(f"      slowest 10% of chains = {share[chains // 10]:.0%} of total time; corr(time, clones) = {np.corrcoef(t, st[:,2])[0,1]:.2f}", flush=True)
