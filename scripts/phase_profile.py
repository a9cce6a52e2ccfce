"""Per-phase shader-clock distribution of the sweep kernel (lane 0 of chain 0)."""
import os, sys, time
os.environ["PMDI_PHASE_TIMERS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
G.build(); pkg = G.load_package()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.batched import DeviceGibbsK1
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 8
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 8
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
w = workloads.make("cfg2", scale)
g = DeviceGibbsK1(w["data"][0], "gaussian", w["N"], w["P"], chains, seed=1000, block_threads=block)
names = ["setup+prefix", "stage+needlist", "terms", "sums", "cdf", "C:draw+census", "D1:keys+firsts", "D2:ranks", "E:apply+stats", "phi+ess+looptop", "resample", "final", "slow:draw/census", "slow:stats|unanimous"]
for it in range(warm + 3):
    t0 = time.perf_counter(); g.iteration(); st = g.check(); dt = time.perf_counter() - t0
    ids = st[:, 0] / (g.n - g.n1 + 1)
    sel = 0
    if len(sys.argv) > 5: sel = int(np.argmax(ids)) if sys.argv[5] == "max" else int(np.argsort(ids)[len(ids) // 2])
    ph = g.sw.phase_timers(sel).astype(np.float64)
    clk_total, rt_total = ph[14], ph[15]
    ph = ph[:14]
    tot = ph.sum()
    n_s = g.n - g.n1 + 1
    print(f"it {it} chain {sel} ids {ids[sel]:.0f} fast/conv/slow {st[sel,5]}/{st[sel,6]}/{st[sel,7]}: wall {dt*1e3:8.1f} ms  ids/step {st[:,0].mean()/n_s:7.1f} cls/step {st[:,4].mean()/n_s:5.2f} resamp {st[:,1].mean():5.1f} clones {st[:,2].mean():7.1f} | "
          + " ".join(f"{nm}={100*v/tot:.1f}%" for nm, v in zip(names, ph) if v > 0.004 * tot) + f" | cycles/step {tot/n_s:.0f} | shader clk {clk_total/max(rt_total,1)*100:.0f} MHz (memtime {clk_total:.3g}, realtime ticks {rt_total:.3g})", flush=True)
