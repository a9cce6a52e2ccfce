"""Small driver for rocprofv3: a few full-size cfg2 iterations on C chains (no timers)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.batched import DeviceGibbsK1
chains = int(sys.argv[1]); iters = int(sys.argv[2]); block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
w = workloads.make("cfg2", scale)
g = DeviceGibbsK1(w["data"][0], "gaussian", w["N"], w["P"], chains, seed=1000, block_threads=block)
for it in range(iters):
    g.iteration()
st = g.check()
print("ids/step", st[:, 0].mean() / (g.n - g.n1 + 1))
