"""Analysis (CPU, oracle): how many DISTINCT columns particle[:, p, k] does a chain hold per step?

The sweep's particle -> cluster table is N x P entries per dataset and a resampling event gathers all of it
(src/pmdi.jl:322).  If the particles share only a few distinct columns, a column-indexed table (P column ids +
C x N entries) would make that gather O(P + C N).  This script runs the oracle from the random start and prints
the distribution of distinct columns per (step, dataset), next to the reference's own class counts.

    python scripts/column_stats.py [HL] [iterations] [scale]
"""
import ctypes as C
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
    G.load_package()
    from particlemdi_jl_amd import workloads
    O = G.load_oracle(); O.build()
    w = workloads.make(name, scale)
    n, N, K, P = w["n"], w["N"], w["K"], w["P"]
    n1 = int(np.floor(w["rho"] * n))
    hy = O.Hypers(n, N, K, seed=3)
    orc = O.Oracle(w["data"], w["kinds"], N, P, seed=3)
    buf = np.zeros((n - n1 + 1, K), dtype=np.int64)
    orc.L.pmdi_oracle_debug_columns.argtypes = [C.c_void_p, C.c_void_p]
    orc.L.pmdi_oracle_debug_columns(orc.h, buf.ctypes.data)
    for it in range(1, iters + 1):
        Pi = hy.step(it)
        r = orc.sweep(it, np.array(hy.s), np.array(hy.order), n1, Pi, hy.Phi, trace=True)
        hy.s[:] = r["s"]
        hy.align_labels(it)
        tr = r["trace"]
        ncls = tr[:, 2 + K:2 + 2 * K]
        q = lambda a: " ".join(f"{np.percentile(a, x):7.0f}" for x in (10, 50, 90, 99, 100))
        res = tr[:, 1] > 0
        print(f"it {it:2d} resamples {int(res.sum()):5d}  columns p10/50/90/99/max {q(buf)}   at resampling steps {q(buf[res]) if res.any() else '-'}"
              f"   classes {q(ncls)}  mean cols {buf.mean():.1f} (at resampling {buf[res].mean() if res.any() else 0:.1f})", flush=True)


if __name__ == "__main__":
    main()
