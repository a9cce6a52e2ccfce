"""Development aid: how well do the chains of a sweep pack onto the workgroup slots?  Per-chain start / end stamps of the settled-chain
kernel (PMDI_PHASE_TIMERS=1), the cost of the previous sweep (what the launch order is built from) and the launch rank, saved for
offline analysis (gpurun_out/packing_<WORKLOAD>.npz).

    PMDI_PHASE_TIMERS=1 python scripts/packing_probe.py [WORKLOAD] [chains] [burn-in] [iterations] [pool_frac]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "HL"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
burn = int(sys.argv[3]) if len(sys.argv) > 3 else 24
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
frac = float(sys.argv[5]) if len(sys.argv) > 5 else 0.4
G.build()
pkg = G.load_package()
from particlemdi_jl_amd import workloads  # noqa: E402
w = workloads.make(name)
cap = 0 if frac >= 1.0 else int(frac * (w["N"] * w["P"] + 1))
sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=C, seed=1000, pool_cap=cap)
g = pkg.Gibbs(sw, rho=w["rho"])
stream = torch.cuda.current_stream()
out = {}
prev = None
for it in range(burn + iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.step(pkg.STEP_BEGIN, stream.cuda_stream); g.step(pkg.STEP_HYPERS, stream.cuda_stream)
    e0.record(stream); g.step(pkg.STEP_SWEEP, stream.cuda_stream); e1.record(stream)
    g.step(pkg.STEP_ALIGN, stream.cuda_stream)
    torch.cuda.synchronize()
    cs = sw.chain_costs().astype(np.float64) / float(sw.clock_hz)
    ms = e0.elapsed_time(e1)
    if it >= burn - 1:
        g.results()
        ph = np.stack([sw.phase_timers(int(c)).astype(np.float64) for c in range(C)])
        j = it - burn + 1
        out[f"t0_{j}"], out[f"t1_{j}"], out[f"cost_{j}"], out[f"by_{j}"], out[f"ms_{j}"] = ph[:, 12], ph[:, 13], cs, sw.swept_by(), np.float64(ms)
    print(f"it {it + 1:3d} sweep {ms:8.1f} ms  busy {cs.sum() / (512 * ms / 1e3):.3f}  chain s p50 {np.median(cs):.3f} max {cs.max():.3f}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"packing_{name}.npz"), **out)
