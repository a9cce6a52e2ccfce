"""Analysis (CPU, oracle): what does a step of a chain look like?  Per (swept observation, dataset): particle classes, clusters the
class leaders read, distinct chosen clusters, clones, distinct columns; per sweep: their maxima and the resampling events.
Sizes the tables of the settled-chain kernel.

    python scripts/step_stats.py [HL] [iterations] [scale] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "HL"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    G.load_package()
    from particlemdi_jl_amd import workloads
    O = G.load_oracle(); O.build()
    w = workloads.make(name, scale)
    n, N, K, P = w["n"], w["N"], w["K"], w["P"]
    n1 = int(np.floor(w["rho"] * n))
    hy = O.Hypers(n, N, K, seed=seed)
    orc = O.Oracle(w["data"], w["kinds"], N, P, seed=seed)
    buf = orc.debug_steps(n - n1 + 1)
    q = lambda a: " ".join(f"{np.percentile(a, x):6.0f}" for x in (50, 90, 99, 99.9, 100))
    for it in range(1, iters + 1):
        Pi = hy.step(it)
        r = orc.sweep(it, np.array(hy.s), np.array(hy.order), n1, Pi, hy.Phi, trace=True)
        hy.s[:] = r["s"]
        hy.align_labels(it)
        tr = r["trace"]
        res = tr[:, 1] > 0
        b = buf
        print(f"it {it:2d} resamples {int(res.sum()):5d} | p50/90/99/99.9/max: classes {q(b[:, :, 0])} | need {q(b[:, :, 1])} | chosen {q(b[:, :, 2])} | "
              f"clones/step mean {b[:, :, 3].mean():.3f} | cols {q(b[:, :, 4])} | maxid {q(b[:, :, 6])} | unanimous {b[:, :, 7].mean():.3f} "
              f"all-K-unanimous obs {np.all(b[:, :, 7] == 1, axis=1).mean():.3f}", flush=True)


if __name__ == "__main__":
    main()
