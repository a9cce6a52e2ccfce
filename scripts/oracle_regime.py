"""CPU study: what regime does a real chain run in (ids, classes, resamples per sweep)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); O = G.load_oracle()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.hypers import HyperState
cfg, scale, P, iters = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
w = workloads.make(cfg, scale)
P = P or w["P"]
n, K, N = w["n"], w["K"], w["N"]
rng = np.random.default_rng(0)
hy = HyperState(n, N, K, rng)
orc = O.Oracle(w["data"], w["kinds"], N, P, seed=3, faithful_cost=1)
order = np.arange(1, n + 1)
n1 = int(np.floor(0.25 * n)); ns = n - n1 + 1
for it in range(1, iters + 1):
    rng.shuffle(order)
    Pi = hy.step_pmdi_order()
    t0 = time.time()
    r = orc.sweep(it, hy.s, order, n1, Pi, hy.Phi)
    dt = time.time() - t0
    hy.s[:] = r["s"]; hy.align_labels()
    st = r["stats"]
    print(f"it {it}: {dt*1e3:8.1f} ms  ids/step {st['n_operations']/(ns*K):8.1f}  classes/step {st['sum_classes']/(ns*K):7.2f}  resamples {st['n_resamples']:5d} ({st['n_resamples']/ns:.3f}/obs) clones/step {st['n_clones']/(ns*K):6.2f} maxid {st['max_id']} nclust {[len(np.unique(hy.s[:,k])) for k in range(K)]}", flush=True)
