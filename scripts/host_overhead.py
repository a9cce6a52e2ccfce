"""Where does the non-kernel time of a DeviceGibbsK1 iteration go?  (2048 chains, cfg2)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package()
from particlemdi_jl_amd import workloads
from particlemdi_jl_amd.batched import DeviceGibbsK1
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = workloads.make("cfg2")
g = DeviceGibbsK1(w["data"][0], "gaussian", w["N"], w["P"], chains, seed=1000)
for _ in range(3):
    g.iteration()
g.check()
def tm(f, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
t_rand, _ = tm(lambda: torch.rand((g.C, g.n), device=g.dev, generator=g.gen))
r = torch.rand((g.C, g.n), device=g.dev, generator=g.gen)
t_sort, _ = tm(lambda: torch.argsort(r, dim=1))
def cnt():
    c = torch.zeros((g.C, g.N), dtype=torch.float64, device=g.dev); c.scatter_add_(1, g.s.long(), g._ones); return c.cpu().numpy()
t_cnt, c = tm(cnt)
t_hy, Pi = tm(lambda: g.hy.step(c))
t_up, _ = tm(lambda: g.Pi.copy_(torch.from_numpy(np.ascontiguousarray(Pi))))
print(f"rand {t_rand:.1f} ms  argsort {t_sort:.1f} ms  counts+D2H {t_cnt:.1f} ms  host hypers {t_hy:.1f} ms  Pi upload {t_up:.1f} ms")
