"""Shader clock (s_memtime vs wall clock) of the sweep kernel alone and with the GPU full: is it throttled under load?"""
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
for it in range(warm):
    g.iteration(time_kernel=True); g.finish_timing(); g.check()
ph = np.array([g.sw.phase_timers(c) for c in range(0, chains, max(1, chains // 64))]).astype(float)
clk = ph[:, 14] / ph[:, 15] * 100
print(f"chains={chains}: kernel {g.kernel_ms[-1]:.0f} ms; in-kernel shader clock over {len(clk)} sampled chains: min {clk.min():.0f} median {np.median(clk):.0f} max {clk.max():.0f} MHz; wall per chain (realtime) median {np.median(ph[:,15])/1e5:.0f} ms")
