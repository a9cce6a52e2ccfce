"""Development aid: what does a sweep of settled chains cost, and how many chains does the settled-chain kernel hand back?

    python scripts/s2_probe.py [WORKLOAD] [chains] [burn-in] [iterations]      (PMDI_PHASE_TIMERS=1: phase shares of a few chains)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "HL"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
burn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
G.build()
pkg = G.load_package()
from particlemdi_jl_amd import workloads  # noqa: E402
w = workloads.make(name)
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=1000)
g = pkg.Gibbs(sw, rho=w["rho"])
print(f"{name}: {C} chains, settled kernel {sw.settled}, split form {sw.split}, lds {sw.lds_bytes}", flush=True)
stream = torch.cuda.current_stream()
last_gb = np.zeros(4, dtype=np.int64)
for it in range(burn + iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.step(pkg.STEP_BEGIN, stream.cuda_stream); g.step(pkg.STEP_HYPERS, stream.cuda_stream)
    e0.record(stream); g.step(pkg.STEP_SWEEP, stream.cuda_stream); e1.record(stream)
    g.step(pkg.STEP_ALIGN, stream.cuda_stream)
    torch.cuda.synchronize()
    st = g.results()["stats"]
    gb = sw.given_back()
    cs = sw.chain_costs() / float(sw.clock_hz)
    n_s = w["n"] - g.n1 + 1
    print(f"it {it + 1:3d} sweep {e0.elapsed_time(e1):8.1f} ms  handed back {(gb - last_gb).tolist()}  chain s p50 {np.median(cs):.3f} p90 {np.percentile(cs, 90):.3f} max {cs.max():.3f}"
          f"  ids/step {st[:, 0].mean() / (n_s * w['K']):6.1f} classes/step {st[:, 4].mean() / (n_s * w['K']):5.2f} resamples {st[:, 1].mean():7.1f}", flush=True)
    last_gb = gb
if os.environ.get("PMDI_PHASE_TIMERS"):
    names = ["prefix", "cluster", "wait1", "particle", "wait2", "ess+book", "wait3", "follow", "resample", "finish"]
    if os.environ.get("PM2_DETAIL"):      # a library built with -DPM2_DETAIL_TIMERS (PMDI_LIB_PATH)
        names = ["needset", "terms", "sums", "uncached", "cdf", "rows", "draws", "chosen+hist", "census", "phi+max", "ess", "lists", "book", "stats", None, "other"]
    cs = sw.chain_costs()
    for c in np.argsort(cs)[[0, C // 4, C // 2, 3 * C // 4, C - 1]]:
        ph = sw.phase_timers(int(c)).astype(np.float64)
        tot = ph[14] if ph[14] > 0 else ph[:10].sum()
        nres = g.results()["stats"][c, 1]
        print(f"chain {c}: {tot / 1e6:8.1f} Mcycles, {nres} resamples ({ph[8] / max(nres, 1) / 1e3:.1f} kcycles each); per observation "
              + " ".join(f"{nm} {ph[i] / n_s / 1e3:.2f}k" for i, nm in enumerate(names) if nm))
