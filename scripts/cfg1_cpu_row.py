"""the reference's own CPU-runnable case (README.md:35-40: iris-shaped 150 x 4, K = 1, N = 10,
32 particles, rho = 0.25, 1000 iterations after a warm-up call) on the oracle -- the single-threaded C restatement of
src/pmdi.jl + update_hypers.jl + align_labels!, reference-cost bookkeeping kept -- on one host core of the box it is run on.
(The iris file itself is not in this image: a 3-component Gaussian mixture of the same shape stands in, particlemdi.jl_amd/workloads.py.)
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

G.load_package()
from particlemdi_jl_amd import workloads  # noqa: E402
O = G.load_oracle()
w = workloads.make("cfg1")
n, N, K, P = w["n"], w["N"], w["K"], w["P"]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hy = O.Hypers(n, N, K, seed=7)
orc = O.Oracle(w["data"], w["kinds"], N, P, seed=7, faithful_cost=1)
n1 = int(np.floor(w["rho"] * n))
flags = [np.ones(d, dtype=np.uint8) for d in w["D"]]
tot, sweep = [], []
for it in range(1, iters + 2):          # (iteration 1 = the README's warm-up call: not timed)
    t0 = time.perf_counter()
    Pi = hy.step(it)
    r = orc.sweep(it, np.array(hy.s), np.array(hy.order), n1, Pi, hy.Phi, flags, lw_init=1.0)
    hy.s[:] = r["s"]
    hy.align_labels(it)
    if it > 1:
        tot.append(time.perf_counter() - t0); sweep.append(r["stats"]["seconds"])
print(json.dumps({"workload": "cfg1: 150 x 4 Gaussian (iris-shaped stand-in), K=1, N=10, P=32, rho=0.25", "iterations": iters,
                  "cpu_iters_per_sec_one_core": len(tot) / sum(tot), "sweep_only_iters_per_sec": len(sweep) / sum(sweep),
                  "swept_obs_per_iter": n - n1 + 1, "kind": "port (oracle)", "cores": 1,
                  "note": "python driver around the C oracle: the per-iteration ctypes overhead is inside the end-to-end figure"}))
