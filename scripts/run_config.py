"""Full-size smoke/timing of a BASELINE config on the GPU: C chains, host hypers per chain (numpy)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.hypers import HyperState
cfg, chains, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
w = workloads.make(cfg, scale)
n, K, N, P = w["n"], w["K"], w["N"], w["P"]
t0 = time.time()
sw = pkg.Sweeper(w["data"], w["kinds"], N, P, n_chains=chains, seed=100)
print(f"{cfg}: n={n} K={K} N={N} P={P} D={w['D']} chains={chains} T={sw.block_threads} LDS={sw.lds_bytes} create {time.time()-t0:.1f}s", flush=True)
rng = np.random.default_rng(0)
hys = [HyperState(n, N, K, np.random.default_rng(100 + c)) for c in range(chains)]
order = np.stack([np.arange(1, n + 1)] * chains)
n1 = int(np.floor(0.25 * n))
for it in range(1, iters + 1):
    th = time.time()
    for c in range(chains):
        rng.shuffle(order[c])
    Pis = np.stack([h.step_pmdi_order() for h in hys]); Phis = np.stack([h.Phi for h in hys]); S = np.stack([h.s for h in hys])
    th = time.time() - th
    t0 = time.time()
    r = sw.sweep(it, S, order, n1, Pis, Phis)
    dt = time.time() - t0
    ta = time.time()
    for c, h in enumerate(hys):
        h.s[:] = r["s"][c]; h.align_labels()
    ta = time.time() - ta
    st = r["stats"]
    ns = n - n1 + 1
    print(f"it {it}: sweep {dt*1e3:9.1f} ms (host hypers {th*1e3:.0f} ms, align {ta*1e3:.0f} ms) ids/step {np.mean([s['n_operations'] for s in st])/(ns*K):8.1f} cls/step {np.mean([s['sum_classes'] for s in st])/(ns*K):6.2f} "
          f"resamples {np.mean([s['n_resamples'] for s in st]):6.1f} fast/conv/slow {np.mean([s['steps_fast'] for s in st]):.0f}/{np.mean([s['steps_converted'] for s in st]):.0f}/{np.mean([s['steps_fallback'] for s in st]):.0f} nclust {[len(np.unique(hys[0].s[:,k])) for k in range(K)]}", flush=True)
    if os.environ.get("PMDI_PHASE_TIMERS"):
        if os.environ.get("PMDI_RESAMPLE_SUBPHASES"):    # library built with -DPMDI_RESAMPLE_TIMERS: the slots hold resampling sub-phases
            rn = ["weights", "cumsum+u", "search+ancestors", "dataset top", "gather", "id scan", "relabel+recount", "counts copy+reset", "moves", "classes"]
            ph = sw.phase_timers(0).astype(np.float64)[:10]
            nres = max(1, st[0]["n_resamples"])
            print("   resampling, cycles per event: " + " ".join(f"{nm}={v/nres:.0f}" for nm, v in zip(rn, ph)) + f" | total {ph.sum()/nres:.0f}", flush=True)
            continue
        names = ["setup+prefix", "stage+needlist", "terms", "sums", "cdf", "C:draw+vote+census", "D1:keys+firsts", "D2:ranks", "E:apply+stats", "phi+ess+looptop", "resample", "final", "slow:draw/census", "slow:stats|unanimous"]
        ph = sw.phase_timers(0).astype(np.float64)[:14]
        print("   chain 0: " + " ".join(f"{nm}={100*v/ph.sum():.1f}%" for nm, v in zip(names, ph) if v > 0.004 * ph.sum()) + f" | cycles/step {ph.sum()/(ns*K):.0f}", flush=True)
if cfg == "cfg5":
    t0 = time.time(); fl, pr = sw.feature_select(iters, np.stack([h.s for h in hys])); print(f"feature_select {1e3*(time.time()-t0):.1f} ms; flags on per dataset {[int(fl[0][k*200:(k+1)*200].sum()) for k in range(3)]}")
