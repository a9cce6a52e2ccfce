#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of ParticleMDI's conditional-SMC sweep on MI355X.

Workload at N=1 (default): the north-star headline "HL" -- 4 Gaussian datasets of 10 000 x 50 sharing a 3-cluster
structure, K = 4, N = 20 clusters, 1 024 particles, rho = 0.25 (BASELINE.json north_star; D and N are SURVEY 8d's
stated assumptions).  `--workload cfg2|cfg3|cfg4|cfg5` runs the other BASELINE configs.

One "step" = one Gibbs iteration (src/pmdi.jl:164-384) of every chain on this rank, entirely on the device through
the C ABI (pmdi_gibbs_step): shuffle + hyper-parameter updates (one kernel), the whole conditional-SMC sweep (one
persistent kernel per launch group, one workgroup per chain), feature selection when the config has it, label
alignment (one kernel).  Inputs are resident: nothing crosses PCIe inside the timed region.

Two forms of the sweep for K > 1 (include/pmdi_hip.h, PMDI_KSPLIT): one workgroup per chain (throughput form: what `value`
is measured with, the GPU filled with chains) and K cooperating workgroups per chain (latency form: a chain's sweep is ~2.2x
shorter; measured in the same run on a subset of the settled chains and reported as `latency_form`).

Protocol (state-independent by construction -- the cost of a sweep depends on how settled the chains are):
  1. `--burnin B` iterations from the random start of src/pmdi.jl:59-66, timed separately (`burnin_iters_per_sec`:
     what a user sees for the first B iterations);  B is a property of the workload, NOT of --warmup;
  2. `--warmup W` untimed iterations;  3. exactly `--steps K` timed iterations -> `value` (settled chains).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload HL] [--chains C] [--burnin B] [--no-cpu]

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU; chains are independent (no
data-path collective, scaling: weak) and the retained allocation samples of the timed iterations are all-gathered
over RCCL at the end for the posterior-similarity matrix.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"

DESCR = {
    "HL": "HL (north-star headline): 4 x Gaussian 10000x50, K=4, N=20, P=1024, rho=0.25",
    "cfg2": "cfg2: 3-mixture Gaussian 10000x50, K=1, N=20, P=1024, rho=0.25",
    "cfg3": "cfg3: Gaussian 5000x50 + Categorical 5000x20, K=2, N=30, P=1024, rho=0.25",
    "cfg4": "cfg4: 2 x Gaussian 10000x50 + Categorical 10000x20 + NegBinom 10000x30, K=4, N=50, P=2048, rho=0.25",
    "cfg5": "cfg5: 3 x Gaussian 20000x200, K=3, N=50, P=4096, rho=0.25, featureSelect on",
}
# chains per GPU (one workgroup each) and burn-in iterations per workload.  HL: 2 048 chains since round 3 (227 GB of the 288 GB: the
# arena of a chain is sized for N * P + 1 cluster ids) -- 512 workgroup slots, so four rounds of chains instead of two pack under the
# slowest chain: 1 146 it/s against 1 001-1 009 with 1 024 chains (profiles/README.md r03 keeps both operating points)
DEFAULTS = {"HL": (3072, 20), "cfg2": (2048, 30), "cfg3": (3072, 12), "cfg4": (1024, 6), "cfg5": (80, 3)}
# Cluster ids per dataset a chain's pool holds, as a fraction of the reference's N*P+1 (src/pmdi.jl:140).  Measured (scripts/pool_need.py,
# profiles/r04/pool_need.txt): the largest id a chain touches is 0.18-0.33 of N*P+1 in its FIRST sweep from the random start and below
# 0.05-0.19 afterwards, so the statistics pool -- 60-85 % of a chain's memory -- is mostly never touched.  A smaller pool is what lets a GPU
# hold 2-4x more chains than workgroup slots (better packing under the slowest chain).  Margin >= 1.5x over the measured first-sweep
# peak; a chain that needs more reports PMDI_E_POOL and the bench starts over with the full pool (FULLPOOL_CHAINS).
POOL_FRAC = {"HL": 0.4, "cfg2": 1.0, "cfg3": 1.0, "cfg4": 0.3, "cfg5": 1.0}
FULLPOOL_CHAINS = {"HL": 2048, "cfg2": 2048, "cfg3": 3072, "cfg4": 512, "cfg5": 80}
# (round 4: HL 3 072 chains (six rounds on the 512 workgroup slots: 1 218 it/s against 1 180 with 2 048), cfg3 2 048, cfg4 1 024 on 256
# slots -- a sweep lasts as long as its slowest chain once the rest packs under it.  cfg5: 80 chains = 240 workgroups of the
# K-workgroups-per-chain form, the only form whose LDS tables hold P = 4 096, N = 50: one per CU; round 3 ran 16)


def algorithmic_bytes(w, P, n, n1, work, stats, layout="column"):
    """De-duplication-aware algorithmic HBM bytes of one sweep of every chain, from the kernel's own work counters
    (pmdi_work_counters; DESIGN.md section 6 has the derivation).  work: (C, K, 8); stats: (C, 8)."""
    N, K = w["N"], w["K"]
    n_s = n - n1 + 1
    C = work.shape[0]
    tot = 0.0
    for k, (kind, D) in enumerate(zip(w["kinds"], w["D"])):
        ev, up, cl, moved, mev, cols, splits = (work[:, k, j].astype(np.float64).sum() for j in range(7))
        if kind == "gaussian":                   # pool entry = (Sigma, beta) per feature (16 B) + n (4 B)
            b_eval, b_upd, b_clone_extra, b_x = 16 * D + 4, 2 * 16 * D + 8, 0, 8 * D
        elif kind == "categorical":              # one count per feature at the observed level; a clone copies L levels
            L = int(w["data"][k].max())
            b_eval, b_upd, b_clone_extra, b_x = 4 * D + 4, 2 * 4 * D + 8, 2 * 4 * D * (L - 1), 4 * D
        else:                                    # NegBinom: Sigma per feature (8 B)
            b_eval, b_upd, b_clone_extra, b_x = 8 * D + 4, 2 * 8 * D + 8, 0, 4 * D
        per_step = b_x + P * (4 + 4 + 1)                               # obs row; per particle: column index, its label->cluster entry, sstar byte
        prefix = (n1 - 1) * (b_x + 8) + N * P * (4 + 8) + P * 4        # known-prefix rows + s_in/order; new_id / split-scratch reset, column indices
        final = n * 4 + n_s * 1                                        # s_out write, sstar[p_star] reads
        # a resampling event: class ids and column indices of the P particles gathered (read + write each), the live columns of
        # the particle -> cluster table read and written once (compacted + relabelled); a copy-on-write split copies a column
        resample = stats[:, 1].astype(np.float64).sum() * (4 * P * 4) + cols * N * 4 * 2 + splits * N * 4 * 2
        if layout == "particle":                 # the N x P table per particle of the earlier builds: same work, more bytes
            per_step = b_x + P * (4 + 1)
            prefix = (n1 - 1) * (b_x + 8) + N * P * 4 * 2
            resample = stats[:, 1].astype(np.float64).sum() * (3 * N * P * 4 + P * 4) + mev * N * P * 4
        tot += ev * b_eval + up * b_upd + cl * b_clone_extra + moved * (b_upd - 8) \
            + resample + C * (n_s * per_step + prefix + final)
    return tot


def oracle_chain(w, state, n_iter, seed, progress=False):
    """The CPU path beside it: the oracle (single-threaded restatement of src/pmdi.jl + update_hypers.jl + align_labels!,
    reference-cost bookkeeping kept) continuing one chain from `state` (a settled chain of the GPU run).
    Returns the seconds of every iteration: (whole iteration, sweep only)."""
    O = G.load_oracle()
    n, N, K, P = w["n"], w["N"], w["K"], w["P"]
    hy = O.Hypers(n, N, K, seed=seed)
    hy.M, hy.gamma, hy.gamma0, hy.Phi = state["M"], state["gamma"], state["gamma0"], state["Phi"]
    hy.v, hy.Z = float(state["v"]), float(state["Z"])
    hy.s[:] = state["s"]
    hy.order[:] = state["order"]
    orc = O.Oracle(w["data"], w["kinds"], N, P, seed=seed, faithful_cost=1)
    n1 = int(np.floor(w["rho"] * n))
    flags = [np.ones(d, dtype=np.uint8) for d in w["D"]]
    out = []
    for it in range(1, n_iter + 1):
        t0 = time.perf_counter()
        Pi = hy.step(1000 + it)
        r = orc.sweep(1000 + it, np.array(hy.s), np.array(hy.order), n1, Pi, hy.Phi, flags, lw_init=1.0)
        if w["name"] == "cfg5":
            flags, _ = orc.feature_select(1000 + it, r["s"])
        hy.s[:] = r["s"]
        hy.align_labels(1000 + it)
        out.append((time.perf_counter() - t0, r["stats"]["seconds"]))
        if progress:
            print(f"[bench] cpu baseline iteration {it}/{n_iter}: {out[-1][0]:.2f} s (sweep {out[-1][1]:.2f} s)", file=sys.stderr, flush=True)
    orc.close(); hy.close()
    return out


KERNEL_NAMES = {0: "general", 1: "settled-chain", 2: "general after a hand-back"}


def one_chain_alone(pkg, w, g, chain, seed, device, fsel, torch):
    """ONE chain alone on the GPU -- the reference's use (README.md:31-46 runs one chain) -- continuing a settled chain of the run:
    sweep-only iterations/s in every form a single-chain handle can take: the settled-chain kernel (one workgroup), the general
    kernel in one workgroup, and for K > 1 the general kernel with K cooperating workgroups (what pmdi_create picks by itself for a
    small batch is reported as `default_form`)."""
    K = w["K"]
    forms = {"settled_chain_kernel": {"ksplit": 0, "settled": 2, "sticky": 0}, "general_kernel_one_workgroup": {"ksplit": 0, "settled": 0}}
    if K > 1:
        forms["general_kernel_K_workgroups"] = {"ksplit": 1}
    forms["default_form"] = {}
    st = g.get(chain)
    stream = torch.cuda.current_stream(torch.device("cuda", device))
    out = {}
    for name, tun in forms.items():
        try:
            sw1 = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=1, seed=seed + (2 << 20) + chain, device=device, tuning=tun)
        except pkg.PmdiError as e:           # (a form this configuration cannot take: e.g. P = 4 096, N = 50 in one workgroup's LDS)
            out[name] = {"not_available": str(e)}
            continue
        g1 = pkg.Gibbs(sw1, rho=w["rho"], feature_select=fsel)
        g1.set(0, M=st["M"], gamma=st["gamma"], gamma0=st["gamma0"], Phi=st["Phi"], v=st["v"], Z=st["Z"], s=st["s"], order=st["order"], flags=st["flags"])
        ms = []
        for it in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g1.step(pkg.STEP_BEGIN, stream.cuda_stream); g1.step(pkg.STEP_HYPERS, stream.cuda_stream)
            e0.record(stream); g1.step(pkg.STEP_SWEEP, stream.cuda_stream); e1.record(stream)
            if fsel:
                g1.step(pkg.STEP_FEATSEL, stream.cuda_stream)
            g1.step(pkg.STEP_ALIGN, stream.cuda_stream)
            torch.cuda.synchronize()
            if it >= 2:
                ms.append(e0.elapsed_time(e1))
        g1.results()
        out[name] = {"sweep_only_iters_per_sec": 1e3 / float(np.mean(ms)), "sweep_ms": float(np.mean(ms)),
                     "finished_by": KERNEL_NAMES.get(int(sw1.swept_by()[0]), "?"), "workgroups": (K if sw1.split else 1),
                     "settled_chain_kernel_enabled": bool(sw1.settled)}
        g1.close(); sw1.close()
    out["note"] = "one settled chain of the run (median cost) alone on the GPU, 4 timed sweeps after 2; the CPU figure beside it is cpu_baseline.one_core_alone_iters_per_sec"
    return out


def parity_check(pkg, w, g, sw, chains, seed, fsel):
    """Outside the timed region: ONE more iteration of the device-resident chains, and every chain of `chains` (label, index) of it
    re-run through the oracle (the checker) from the same post-hyper-update state, same Philox key, same iteration number:
    allocations, the picked particle and the five counters must be equal, the log-weights within the north-star tolerance.  Which
    kernel swept each compared chain comes back with it (pmdi_chain_swept_by).  Returns the JSON fragment."""
    O = G.load_oracle()
    n, N, K, P = w["n"], w["N"], w["K"], w["P"]
    g.step(pkg.STEP_BEGIN); g.step(pkg.STEP_HYPERS)
    before = {c: g.get(c) for _, c in chains}
    g.step(pkg.STEP_SWEEP)
    res = g.results()
    swept = sw.swept_by()
    s_dev = {c: g.get(c)["s"] for _, c in chains}      # before the label alignment: s = sstar[p_star, :, :]
    if fsel:
        g.step(pkg.STEP_FEATSEL)
    g.step(pkg.STEP_ALIGN)
    it = g.iterations
    detail, all_ok = [], True
    for label, chain in chains:
        st1 = before[chain]
        Pi = st1["gamma"] / st1["gamma"].sum(axis=0, keepdims=True)
        flags = [st1["flags"][sum(w["D"][:k]):sum(w["D"][:k + 1])] for k in range(K)]
        orc = O.Oracle(w["data"], w["kinds"], N, P, seed=seed + chain)
        ro = orc.sweep(it, st1["s"], st1["order"], g.n1, Pi, st1["Phi"], flags, lw_init=1.0)
        orc.close()
        same_s = bool((s_dev[chain] == ro["s"]).all())
        same_p = int(res["p_star"][chain]) == int(ro["p_star"])
        same_counters = all(int(res["stats"][chain, j]) == int(ro["stats"][key])
                            for j, key in enumerate(("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes")))
        lw_ok = bool(np.allclose(res["logweight"][chain], ro["logweight"], rtol=1e-6, atol=1e-6))
        all_ok = all_ok and same_s and same_p and same_counters and lw_ok
        detail.append({"which": label, "chain": int(chain), "kernel": KERNEL_NAMES.get(int(swept[chain]), "?"), "allocations_equal": same_s,
                       "p_star_equal": same_p, "counters_equal": same_counters, "logweights_within_1e-6": lw_ok,
                       "oracle_sweep_seconds": float(ro["stats"]["seconds"])})
    return {"parity_check": "ok" if all_ok else "FAILED",
            "parity_check_detail": {"iteration": int(it), "chains": detail,
                                    "note": "timed-state chains (median cost, 99th percentile of cost, a chain of the last launch round) re-run through the "
                                            "CPU oracle (checker only) outside the timed region; `kernel` = which device kernel swept the compared chain"}}


def host_core_share():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each
    tenant a share of a large host: 256 hardware threads visible, 16 cores' worth of quota)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, q // pr))
        except Exception:
            pass
    return cores


def _cpu_worker(args):
    name, scale, path, chain, n_iter, seed = args
    sys.path.insert(0, ROOT)
    G.load_package()
    from particlemdi_jl_amd import workloads
    w = workloads.make(name, scale)
    z = np.load(path)
    state = {k[len(f"c{chain}_"):]: z[k] for k in z.files if k.startswith(f"c{chain}_")}
    return oracle_chain(w, state, n_iter, seed)


def cpu_baseline(w, scale, states, budget_s):
    """One oracle chain per host core, all at once (what a user of the single-threaded reference can do with
    `julia -p`), each continuing a different settled chain of the GPU run; the iteration count is sized from one
    probe iteration so that the leg stays near `budget_s` seconds."""
    import multiprocessing as mp
    host_cores = host_core_share()
    t0 = time.perf_counter()
    probe = oracle_chain(w, states[0], 1, 5, progress=True)[0]          # also: the 1-core figure with the box otherwise idle
    n_iter = int(max(1, min(8, (budget_s - probe[0]) // max(probe[0] * 1.3, 1e-3))))
    # one oracle pool is (N*P+1) cluster objects per dataset (src/pmdi.jl:140): bound the workers by host memory
    pool_bytes = sum((w["N"] * w["P"] + 1) * d * (32 if kind == "gaussian" else 8 * (int(x.max()) if kind == "categorical" else 1))
                     for kind, d, x in zip(w["kinds"], w["D"], w["data"]))
    workers = int(max(1, min(host_cores, 24e9 // max(pool_bytes, 1))))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "states.npz")
        np.savez(path, **{f"c{c}_{k}": v for c, st in enumerate(states) for k, v in st.items()})
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_cpu_worker, [(w["name"], scale, path, c % len(states), n_iter, 100 + c) for c in range(workers)])
    per_core = [np.mean([x[0] for x in r]) for r in res]
    sweep_frac = float(np.mean([x[1] / x[0] for r in res for x in r]))
    return {"value": float(sum(1.0 / x for x in per_core)), "unit": "Gibbs iters/s (whole iterations, one chain per core, aggregate)",
            "cores": workers, "host_cores": host_cores, "kind": "port",
            "one_core_alone_iters_per_sec": 1.0 / probe[0], "per_core_iters_per_sec_all_busy": float(np.mean([1.0 / x for x in per_core])),
            "sweep_fraction_of_iteration": sweep_frac,
            "sample": f"oracle (C restatement of pmdi.jl sweep + update_hypers.jl + align_labels!, single-threaded like the reference), "
                      f"same workload, {workers} chains at once (one per host core{'' if workers == host_cores else ', capped by host memory'}) x {n_iter} iteration(s), each continuing a settled "
                      f"chain of the GPU run; 1 probe iteration alone first ({time.perf_counter() - t0:.0f} s of wall time in total)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="HL", choices=sorted(DESCR))
    ap.add_argument("--chains", type=int, default=0, help="independent chains per GPU (one workgroup each); 0 = the workload's default")
    ap.add_argument("--burnin", type=int, default=-1, help="iterations from the random start before --warmup; -1 = the workload's default")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink n of the workload (debug only)")
    ap.add_argument("--pool-frac", type=float, default=0.0, help="cluster ids per dataset a chain's pool holds, as a fraction of the reference's N*P+1 "
                    "(src/pmdi.jl:140); 0 = the workload's default (POOL_FRAC).  A sweep that needs more stops with PMDI_E_POOL: the bench then starts "
                    "over with the full pool and the chain count that fits it")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of one timed-state chain (outside the timed region)")
    ap.add_argument("--no-latency-form", action="store_true", help="skip the split-form leg (K > 1)")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="wall-time budget of the CPU baseline leg")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    saved_stdout = None
    if world > 1 or os.environ.get("PMDI_BENCH_FORCE_DIST"):     # the env var rehearses the RCCL path on one GPU
        # RCCL prints a version banner on stdout when a communicator is created: stdout carries ONE JSON line, so everything
        # up to that line goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    G.build()
    pkg = G.load_package()
    comm = None
    if dist is not None:
        # the exchange step goes through the library's own RCCL communicator (pmdi_allgather_samples); torch.distributed only
        # carries the 128-byte unique id to the other ranks and the barriers / max-over-ranks of the timing contract
        uid = torch.from_numpy(pkg.Comm.unique_id() if rank == 0 else np.zeros(128, dtype=np.uint8)).to(dev)
        dist.broadcast(uid, 0)
        comm = pkg.Comm(local_rank, world, rank, uid.cpu().numpy())
    from particlemdi_jl_amd import workloads

    w = workloads.make(args.workload, args.scale)
    n, N, P, K = w["n"], w["N"], w["P"], w["K"]
    C = args.chains or DEFAULTS[args.workload][0]
    burnin = args.burnin if args.burnin >= 0 else DEFAULTS[args.workload][1]
    # chain c of rank r draws from Philox key base + r * 2^32 + c: no two chains of a job share a stream
    seed = 1000 + (rank << 32)
    chains_requested = C
    state = {}

    def build_chains(C_, pool_frac):
        """The handle (all chains of this rank) and its device-resident Gibbs state; fewer chains if the device's free memory says so."""
        pool_cap = 0 if pool_frac >= 1.0 else max(N + 2, int(pool_frac * (N * P + 1)))
        while True:
            try:
                sw_ = pkg.Sweeper(w["data"], w["kinds"], N, P, n_chains=C_, seed=seed, device=local_rank, block_threads=args.block, pool_cap=pool_cap)
                break
            except pkg.PmdiError as e:
                if e.code != -3 or C_ <= 8:          # PMDI_E_MEMORY
                    raise
                C_ = max(8, (C_ * 3 // 4) // 8 * 8)
                print(f"[bench] rank {rank}: not enough device memory for {chains_requested} chains; trying {C_}", file=sys.stderr, flush=True)
        if dist is not None:                         # every rank runs the same number of chains
            tC = torch.tensor([C_], dtype=torch.int64, device=dev)
            dist.all_reduce(tC, op=dist.ReduceOp.MIN)
            if int(tC.item()) != C_:
                C_ = int(tC.item())
                sw_.close()
                sw_ = pkg.Sweeper(w["data"], w["kinds"], N, P, n_chains=C_, seed=seed, device=local_rank, block_threads=args.block, pool_cap=pool_cap)
        state["sw"], state["g"], state["C"] = sw_, pkg.Gibbs(sw_, rho=w["rho"], feature_select=(args.workload == "cfg5")), C_

    pool_frac = args.pool_frac if args.pool_frac > 0 else POOL_FRAC.get(args.workload, 1.0)
    build_chains(C, pool_frac)
    sw, g, C = state["sw"], state["g"], state["C"]
    n1 = g.n1
    n_s = n - n1 + 1
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream

    def iteration(events=None):
        g.step(pkg.STEP_BEGIN, sp)
        g.step(pkg.STEP_HYPERS, sp)
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        g.step(pkg.STEP_SWEEP, sp)
        if events is not None:
            e1.record(stream)
            events.append((e0, e1))
        if args.workload == "cfg5":
            g.step(pkg.STEP_FEATSEL, sp)
        g.step(pkg.STEP_ALIGN, sp)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- 1. burn-in from the random start (timed on its own) ----
    barrier()
    tb = time.perf_counter()
    for b in range(burnin):
        if b == 1 and pool_frac < 1.0:
            # the first sweep from the random start is the one that needs the most cluster ids: did every chain's pool hold them?
            torch.cuda.synchronize()
            bad = 0
            try:
                g.results()
            except pkg.PmdiError as e:
                if e.code != -4:
                    raise
                bad = 1
            if dist is not None:
                tb_ = torch.tensor([bad], dtype=torch.int64, device=dev)
                dist.all_reduce(tb_, op=dist.ReduceOp.MAX)
                bad = int(tb_.item())
            if bad:
                print(f"[bench] a chain's first sweep needed more than {pool_frac:.2f} x (N*P+1) cluster ids: starting over with the full pool", file=sys.stderr, flush=True)
                g.close(); sw.close()
                pool_frac = 1.0
                build_chains(min(chains_requested, FULLPOOL_CHAINS.get(args.workload, chains_requested)), 1.0)
                sw, g, C = state["sw"], state["g"], state["C"]
                iteration([])
                tb = time.perf_counter() - 0.0
        ev = []
        iteration(ev)
        if args.verbose and rank == 0:
            torch.cuda.synchronize()
            st = g.results()["stats"]
            print(f"[bench] burn-in {b + 1}: sweep {ev[0][0].elapsed_time(ev[0][1]):8.1f} ms  ids/step {st[:, 0].mean() / (n_s * K):7.1f} "
                  f"classes/step {st[:, 4].mean() / (n_s * K):6.2f} resamples {st[:, 1].mean():7.1f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    burnin_s = time.perf_counter() - tb
    g.results()          # raises if a chain reported an error
    # ---- 2. warm-up ----
    for _ in range(args.warmup):
        iteration()
    # ---- 3. the timed region: exactly --steps iterations ----
    samples = torch.empty((args.steps, C, K, n), dtype=torch.uint8, device=dev)
    per = C * K * n
    events = []
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        iteration(events)
        g.pack_samples(samples.data_ptr() + k * per, sp)       # retained allocation sample of this iteration (uint8): the PSM's input
    torch.cuda.synchronize()
    allgather_ms = None
    if comm is not None:
        # every retained sample of every rank reaches every rank (the PSM's input), a few iterations' worth at a time: the receive
        # buffer is reused, so that 8 ranks x 20 iterations x 3 072 chains (19.7 GB per rank in one piece) fit beside the chains' arenas
        chunk = max(1, min(args.steps, int(4e9 // max(1, world * per))))
        gathered = torch.empty((world, chunk) + tuple(samples.shape[1:]), dtype=torch.uint8, device=dev)
        ag0, ag1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ag0.record(stream)
        for k0 in range(0, args.steps, chunk):
            kn = min(chunk, args.steps - k0)
            comm.allgather(samples.data_ptr() + k0 * per, gathered.data_ptr(), kn * per, sp)      # RCCL over xGMI
        ag1.record(stream)
        torch.cuda.synchronize()
        allgather_ms = float(ag0.elapsed_time(ag1))       # inside the timed region; reported on its own so that sampling and exchange separate
    barrier()
    dt = time.perf_counter() - t0
    res = g.results()
    stats = res["stats"]
    work = sw.work_counters()
    costs = sw.chain_costs().astype(np.float64)               # shader cycles of every chain's last sweep
    swept = sw.swept_by()                                     # ... and which kernel finished it
    given_back = sw.given_back()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    out = None
    if rank == 0:
        total_iters = args.steps * C * world
        clock_hz = float(sw.clock_hz)
        chain_s = costs / clock_hz
        alg = algorithmic_bytes(w, P, n, n1, work, stats)                       # bytes of the last timed sweep, all chains
        alg_pp = algorithmic_bytes(w, P, n, n1, work, stats, layout="particle")   # same work on the per-particle N x P table of the earlier builds
        dense = float(workloads.algorithmic_bytes_per_obs_particle(w["kinds"], w["D"], N)) * n_s * P * C
        achieved = alg / (kernel_ms * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        key = f"{args.workload}|chains={C}|burnin={burnin}|warmup={args.warmup}|steps={args.steps}|scale={args.scale}"
        if os.path.exists(tpath):       # PMC-measured HBM bytes per sweep of this exact command line (profiles/README.md), else null
            tj = json.load(open(tpath))
            traffic = tj.get(key, {}).get("hbm_bytes_per_sweep")
            if traffic is None:           # the bytes of a sweep do not depend on how many sweeps were timed: same workload, chains, burn-in
                pre = f"{args.workload}|chains={C}|burnin={burnin}|"
                traffic = next((v.get("hbm_bytes_per_sweep") for k_, v in tj.items() if k_.startswith(pre) and k_.endswith(f"|scale={args.scale}")), None)
        # second ceiling: how busy the issue slots of the dominant kernel are (PMC, profiles/r04/pmc_instruction_mix.json): HBM is at a few
        # per cent of its peak here and says little about this kernel; its waves wait on dependent LDS / L2 round trips, logs and barriers
        issue = None
        mpath = os.path.join(ROOT, "profiles", "r04", "pmc_instruction_mix.json")
        if args.workload == "HL" and os.path.exists(mpath):
            mj = json.load(open(mpath))
            fr = mj["fractions_of_wave_cycles"]
            issue = {"issue_slots_busy_frac": fr["SQ_ACTIVE_INST_ANY"], "valu_frac": fr["SQ_ACTIVE_INST_VALU"], "scalar_frac": fr["SQ_ACTIVE_INST_SCA"],
                     "lds_frac": fr["SQ_ACTIVE_INST_LDS"], "wait_frac": fr["SQ_WAIT_ANY"], "waves_per_simd": 2,
                     "instructions_per_wave_and_observation": mj["instructions_per_wave_and_observation"],
                     "source": "profiles/r04/pmc_instruction_mix.json (rocprofv3 --pmc, two passes, the settled-chain kernel's dispatch of a 512-chain sweep of the final tree): "
                               "fractions of SQ_WAVE_CYCLES"}
        out = {
            "metric": "Gibbs iters/sec (and obs·particles/sec) at 1/2/4/8 GPUs vs CPU ref",
            "value": total_iters / dt,
            "unit": "Gibbs iters/s (aggregate over chains, settled chains)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": DESCR[args.workload] if args.scale == 1.0 else f"{args.workload} scaled n={n}",
                       "chains_per_gpu": C, "chains_requested": chains_requested, "burnin_iterations": burnin,
                       "pool_ids_per_dataset": int(sw.cap), "pool_frac_of_reference": float(sw.cap) / float(N * P + 1),
                       "block_threads": sw.block_threads, "lds_bytes_per_workgroup": sw.lds_bytes,
                       "workgroups_per_chain": (K if sw.split else 1),
                       "swept_obs_per_iter": n_s, "parallelism": f"chains x{world}",
                       "host_side": "none inside an iteration: hypers, shuffle, sweep, label alignment are device kernels (pmdi_gibbs_step)"},
            "obs_particles_per_sec": total_iters * n_s * P / dt,
            "burnin_iters_per_sec": (burnin * C * world / burnin_s) if burnin else None,
            "sweep_kernel_ms": kernel_ms,
            "sweep_kernel_ms_each": [round(float(a.elapsed_time(b)), 1) for a, b in events],
            "allgather_ms": allgather_ms, "allgather_bytes_per_rank": (int(samples.numel()) if allgather_ms is not None else None),
            "sweep_only_iters_per_sec": C * world / (kernel_ms * 1e-3),
            "chain_slot_busy_frac": float(chain_s.sum() / (min(C * (K if sw.split else 1), (2 if sw.lds_bytes <= 80 * 1024 - 256 else 1) * 256) * kernel_ms * 1e-3))
            if not sw.split else None,
            "per_chain_iters_per_sec": {"p50": float(1.0 / np.median(chain_s)), "slowest": float(1.0 / chain_s.max()),
                                        "fastest": float(1.0 / chain_s.min()),
                                        "note": "1 / (shader cycles of the chain's own sweep / shader clock), all chains co-resident"},
            "kernels": {"settled_chain_kernel": bool(sw.settled),
                        "chains_by_kernel_last_sweep": {KERNEL_NAMES[j]: int((swept == j).sum()) for j in (0, 1, 2)},
                        "chain_seconds_p50_by_kernel": {KERNEL_NAMES[j]: (float(np.median(chain_s[swept == j])) if (swept == j).any() else None) for j in (0, 1, 2)},
                        "handed_back_so_far": {"more_than_32_classes": int(given_back[1]), "more_than_16_classes_or_wide_ids": int(given_back[2]), "total": int(given_back[3])},
                        "note": "pmdi_chain_swept_by / pmdi_settled_kernel: which device kernel finished each chain's last timed sweep"},
            "sweep_stats_last": {"ids_per_step": float(stats[:, 0].mean()) / (n_s * K),
                                 "classes_per_step": float(stats[:, 4].mean()) / (n_s * K),
                                 "resamples": float(stats[:, 1].mean()), "clones": float(stats[:, 2].mean()),
                                 "clusters_evaluated_per_step": float(work[:, :, 0].sum()) / (C * n_s * K),
                                 "clusters_updated_per_step": float(work[:, :, 1].sum()) / (C * n_s * K),
                                 "steps_fast_frac": float(stats[:, 5].sum()) / float(stats[:, 5:8].sum())},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_over_algorithmic": (traffic / alg) if traffic else None,
                         "algorithmic_bytes_per_sweep": alg, "dense_model_ratio": dense / alg,
                         "issue_rate": issue,
                         "per_particle_table_model": {"algorithmic_bytes_per_sweep": alg_pp, "frac": alg_pp / (kernel_ms * 1e-3) / HBM_PEAK,
                                                      "note": "the same sweep priced with the byte model of the builds that kept particle[:, :, k] "
                                                              "as an N x P table per particle (rounds 1-2 before the column table): what frac "
                                                              "compares with across rounds; the column table needs ~4x fewer bytes for the same work"},
                         "note": "achieved = de-duplication-aware algorithmic bytes of one sweep (built from the kernel's work counters: "
                                 "clusters evaluated / updated / cloned / moved, resampling events, per-step and per-sweep streams; "
                                 "DESIGN.md section 6) / sweep time. traffic = PMC-measured HBM bytes per sweep of this operating point (profiles/hbm_traffic.json; "
                                 "null when that operating point has no counter pass); where it is BELOW the algorithmic bytes (HL since round 3) the settled-chain kernel keeps the tables the model "
                                 "prices as memory traffic in LDS and registers (profiles/README.md). dense_model_ratio = SURVEY 8d's dense-model bytes / these: the work "
                                 "the reference's de-duplication (kept here) avoids. A sweep is bound by dependent latency, not by HBM."},
        }
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and not args.no_parity and args.workload != "cfg5":
        # ---- parity of the timed kernel instantiation, outside the timed region (cfg5: the oracle needs minutes per sweep)
        by_cost = np.argsort(costs)
        last_round = int(sw.launch_order()[-1]) if hasattr(sw, "launch_order") else int(by_cost[0])
        picks = [("median cost", int(by_cost[C // 2])), ("99th percentile of cost", int(by_cost[min(C - 1, int(0.99 * C))])), ("last launch round", last_round)]
        out.update(parity_check(pkg, w, g, sw, picks, seed, False))
        if out["parity_check"] != "ok":
            print(json.dumps(out))
            raise SystemExit("bench: the timed kernel's results differ from the oracle's: " + json.dumps(out["parity_check_detail"]))
    if rank == 0 and K > 1 and not args.no_latency_form and not sw.split:
        # ---- the latency form on settled chains of this run: K cooperating workgroups per chain, as many chains as are resident at once
        order_c = np.argsort(costs)
        C2 = max(1, min(C, 512 // K))
        picks = [int(order_c[int(q * (C - 1) / max(C2 - 1, 1))]) for q in range(C2)]          # across the cost range
        os.environ["PMDI_KSPLIT"] = "1"
        sw2 = pkg.Sweeper(w["data"], w["kinds"], N, P, n_chains=C2, seed=seed + (1 << 20), device=local_rank, block_threads=args.block)
        del os.environ["PMDI_KSPLIT"]
        g2 = pkg.Gibbs(sw2, rho=w["rho"], feature_select=(args.workload == "cfg5"))
        for j, c in enumerate(picks):
            st = g.get(c)
            g2.set(j, M=st["M"], gamma=st["gamma"], gamma0=st["gamma0"], Phi=st["Phi"], v=st["v"], Z=st["Z"], s=st["s"], order=st["order"], flags=st["flags"])
        lat_ms = []
        for it in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g2.step(pkg.STEP_BEGIN, sp); g2.step(pkg.STEP_HYPERS, sp)
            e0.record(stream); g2.step(pkg.STEP_SWEEP, sp); e1.record(stream)
            g2.step(pkg.STEP_ALIGN, sp)
            torch.cuda.synchronize()
            if it >= 2:
                lat_ms.append(e0.elapsed_time(e1))
        g2.results()
        c2 = sw2.chain_costs().astype(np.float64) / float(sw2.clock_hz)
        out["latency_form"] = {"split": bool(sw2.split), "chains": C2, "workgroups_per_chain": K, "sweep_kernel_ms": float(np.mean(lat_ms)),
                               "per_chain_iters_per_sec": {"p50": float(1.0 / np.median(c2)), "slowest": float(1.0 / c2.max()), "fastest": float(1.0 / c2.min())},
                               "aggregate_sweep_only_iters_per_sec": C2 / (float(np.mean(lat_ms)) * 1e-3),
                               "handoffs_per_iteration": n_s,
                               "note": "K cooperating workgroups per chain (one per dataset), one sc1 hand-off per swept observation; "
                                       "same settled chains as the throughput run, 3 timed iterations after 2"}
        g2.close(); sw2.close()
    if rank == 0 and not args.no_latency_form:
        out["one_chain_alone"] = one_chain_alone(pkg, w, g, int(np.argsort(costs)[C // 2]), seed, local_rank, args.workload == "cfg5", torch)
    if rank == 0:
        print(f"[bench] GPU part done: {out['value']:.1f} iters/s", file=sys.stderr, flush=True)
        if not args.no_cpu:
            order = np.argsort(costs)
            pick = [int(order[int(q * (C - 1))]) for q in (0.5, 0.1, 0.9, 0.3, 0.7, 0.2, 0.8, 0.4)]   # settled chains across the cost range
            states = [g.get(c) for c in pick]
            out["cpu_baseline"] = cpu_baseline(w, args.scale, states, args.cpu_seconds)
            out["gpu_over_all_host_cores"] = out["value"] / out["cpu_baseline"]["value"]
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    g.close(); sw.close()


if __name__ == "__main__":
    main()
