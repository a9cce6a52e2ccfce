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
    if it >= burn:
        ids = st[:, 0] / (n_s * w["K"])
        hv = ids > float(os.environ.get("PMDI_LIGHT_IDS", "40"))
        def pc(v):
            return "-" if v.size == 0 else f"n {v.size} p50 {np.median(v):.3f} p90 {np.percentile(v, 90):.3f} max {v.max():.3f} sum {v.sum():.1f}"
        print(f"      next sweep's groups by ids/step: light [{pc(cs[~hv])}]  heavy [{pc(cs[hv])}]   ids/step p50 {np.median(ids):.1f} p90 {np.percentile(ids, 90):.1f} p99 {np.percentile(ids, 99):.1f} max {ids.max():.1f}", flush=True)
if os.environ.get("PMDI_PHASE_TIMERS"):
    names = ["prefix", "cluster", "wait1", "particle", "wait2", "ess+book", "wait3", "follow", "resample", "finish"]
    if os.environ.get("PM2_DETAIL") == "2":   # -DPM2_DETAIL_TIMERS=2: inside the resampling events
        names = ["exp+uniforms", "cumsum", "utab", "slotcounts", "search", "ancestors", "scatter+gather", "percolumn+leaders", "colranks", "idocc",
                 "idranks", "countsmove", "statsmove", "cache+cols+classes", None, "outside"]
    elif os.environ.get("PM2_DETAIL") == "3":   # -DPM2_DETAIL_TIMERS=3: inside the bookkeeping phase of dataset 0
        names = ["lists", "fasttest", "clone-or-inplace", "classids", "classreps", "colsplits", "classlist", "chosenstats", "cleanup", "barrier3",
                 None, None, None, None, None, "outside"]
    elif os.environ.get("PM2_DETAIL"):      # a library built with -DPM2_DETAIL_TIMERS (PMDI_LIB_PATH)
        names = ["needset", "terms", "sums", "uncached", "cdf", "rows", "draws", "chosen+hist", "census", "phi+max", "ess", "lists", "book", "stats", None, "other"]
    cs = sw.chain_costs()
    allph = np.stack([sw.phase_timers(int(c)).astype(np.float64) for c in range(C)])
    if not os.environ.get("PM2_DETAIL"):
        t0w, t1w = allph[:, 12], allph[:, 13]
        ran = (t0w > t0w.max() - 1e9) & (t1w > t0w)          # (chains the general kernel swept hold other things in these slots)
        base = t0w[ran].min()
        odd = ~ran
        if odd.any():
            ids_ = st[:, 0] / (n_s * w["K"])
            rank = np.argsort(np.argsort(-cs))
            print(f"   {odd.sum()} chains swept by the general kernel: ids/step p50 {np.median(ids_[odd]):.1f} min {ids_[odd].min():.1f} max {ids_[odd].max():.1f}; "
                  f"chain s p50 {np.median(cs[odd]) / float(sw.clock_hz):.3f} max {cs[odd].max() / float(sw.clock_hz):.3f}; cost rank p50 {np.median(rank[odd]):.0f} max {rank[odd].max()}; "
                  f"classes/step p50 {np.median(st[odd, 4]) / (n_s * w['K']):.2f}; resamples p50 {np.median(st[odd, 1]):.0f} (all chains {np.median(st[:, 1]):.0f})")
        st_, en_ = (t0w[ran] - base) / 1e5, (t1w[ran] - base) / 1e5        # ms at 100 MHz
        print(f"settled kernel timeline ({ran.sum()} chains): starts p50 {np.median(st_):.0f} ms, p90 {np.percentile(st_, 90):.0f}, max {st_.max():.0f}; "
              f"ends p50 {np.median(en_):.0f} p90 {np.percentile(en_, 90):.0f} max {en_.max():.0f}; first-round chains (start < 5 ms) {int((st_ < 5).sum())}; "
              f"sum of durations {np.sum(en_ - st_) / 1e3:.1f} s")
        import heapq
        dur = np.sort(en_ - st_)[::-1]
        for slots in (512,):
            h = [0.0] * slots
            heapq.heapify(h)
            for d in dur:
                heapq.heappush(h, heapq.heappop(h) + d)
            print(f"   longest-first packing of these durations on {slots} slots: {max(h):.0f} ms")
        late = np.argsort(en_)[-8:]
        print("   last to end: " + ", ".join(f"[start {st_[i]:.0f} dur {en_[i] - st_[i]:.0f}]" for i in late))
    print("mean over chains, per observation: " + " ".join(f"{nm} {allph[:, i].mean() / n_s / 1e3:.2f}k" for i, nm in enumerate(names) if nm)
          + f" | whole sweep {allph[:, 14].mean() / n_s / 1e3:.2f}k")
    for c in np.argsort(cs)[[0, C // 4, C // 2, 3 * C // 4, C - 6, C - 5, C - 4, C - 3, C - 2, C - 1]]:
        ph = sw.phase_timers(int(c)).astype(np.float64)
        tot = ph[14] if ph[14] > 0 else ph[:10].sum()
        nres = g.results()["stats"][c, 1]
        print(f"chain {c}: {tot / 1e6:8.1f} Mcycles, {nres} resamples ({ph[8] / max(nres, 1) / 1e3:.1f} kcycles each); per observation "
              + " ".join(f"{nm} {ph[i] / n_s / 1e3:.2f}k" for i, nm in enumerate(names) if nm))
