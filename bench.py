#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the conditional-SMC sweep on MI355X.

One "step" = one Gibbs iteration of every chain on this rank: shuffle, host hyper-updates
(M, gamma, Pi), then the whole sweep (src/pmdi.jl:165-171, 188-350, 373) as ONE persistent HIP
kernel launch that advances all `chains` chains (one workgroup per chain).  Workload at N=1:
BASELINE.json configs[1] ("cfg2": synthetic 3-mixture Gaussian 10k x 50, K=1, N=20, 1024
particles, rho=0.25).  `value` is the whole-job aggregate over all chains and ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--chains C] [--scale S] [--no-cpu]

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU;
chains are independent (no data-path collective) and the retained allocation samples are
all-gathered over RCCL at the end for the posterior-similarity matrix (scaling: weak).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def cpu_baseline(w, warm, timed, seed=5):
    """The oracle (a single-threaded port of the reference's loop, reference-cost bookkeeping
    kept) timed on this box's host cores for the same workload, one chain, 1 core."""
    O = G.load_oracle()
    from particlemdi_jl_amd.hypers import HyperState
    n, N, P = w["n"], w["N"], w["P"]
    rng = np.random.default_rng(seed - 4)
    hy = HyperState(n, N, 1, rng)
    orc = O.Oracle(w["data"], w["kinds"], N, P, seed=seed, faithful_cost=1)
    order = np.arange(1, n + 1)
    n1 = int(np.floor(w["rho"] * n))
    secs = []
    for it in range(1, warm + timed + 1):
        rng.shuffle(order)
        Pi = hy.step_pmdi_order()
        r = orc.sweep(it, hy.s, order, n1, Pi, hy.Phi)
        hy.s[:] = r["s"]
        if it > warm:
            secs.append(r["stats"]["seconds"])
        if seed == 5:
            print(f"[bench] cpu baseline iteration {it}/{warm + timed}: {r['stats']['seconds']:.2f} s", file=sys.stderr, flush=True)
    orc.close()
    return float(np.mean(secs)), secs


def _cpu_worker(args):
    scale, warm, timed, seed = args
    sys.path.insert(0, ROOT)
    G.load_package()
    from particlemdi_jl_amd import workloads
    w = workloads.make("cfg2", scale)
    sec, _ = cpu_baseline(w, warm, timed, seed)
    return sec


def cpu_baseline_all_cores(scale, warm, timed):
    """the all-cores figure -- one independent chain per host core, what a user of the
    single-threaded reference could do with `julia -p`.  Returns (aggregate iters/s, cores)."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with mp.get_context("spawn").Pool(cores) as pool:
        secs = pool.map(_cpu_worker, [(scale, warm, timed, 5 + c) for c in range(cores)])
    return float(sum(1.0 / x for x in secs)), cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--chains", type=int, default=2048, help="independent chains per GPU (one workgroup each)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink n of the workload (debug only)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--groups", type=int, default=1,
                    help="drive the chains of a GPU as this many independent groups, each on its own stream "
                         "(measured: 2 groups 15%% slower, 4 groups 40%% slower than 1 -- a launch is bounded by its "
                         "slowest chain and concurrent launches slow each other's slow chains down)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=30)
    ap.add_argument("--cpu-all", action="store_true", help="also time one oracle chain per host core (SURVEY 8d all-cores figure; minutes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("PMDI_BENCH_FORCE_DIST"):     # the env var rehearses the RCCL path on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    G.build()
    pkg = G.load_package()
    from particlemdi_jl_amd import workloads
    from particlemdi_jl_amd.batched import DeviceGibbsK1

    w = workloads.make("cfg2", args.scale)
    n, N, P, C = w["n"], w["N"], w["P"], args.chains
    NG = max(1, min(args.groups, C))
    Cg = [C // NG + (1 if gi < C % NG else 0) for gi in range(NG)]
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(NG)]
    groups = []
    for gi in range(NG):
        with torch.cuda.stream(streams[gi]):
            groups.append(DeviceGibbsK1(w["data"][0], "gaussian", N, P, Cg[gi], seed=1000 + 7919 * rank + 104729 * gi,
                                        device=local_rank, block_threads=args.block, rho=w["rho"]))
    g = groups[0]
    n_s = n - g.n1 + 1

    def all_stats():
        return np.concatenate([gr.check() for gr in groups], axis=0)

    for w_it in range(args.warmup):
        for gi, gr in enumerate(groups):
            with torch.cuda.stream(streams[gi]):
                gr.iteration()
        if os.environ.get("PMDI_BENCH_VERBOSE") and rank == 0:
            st = all_stats()
            print(f"warmup {w_it}: ids/step {st[:, 0].mean() / n_s:.1f} classes/step {st[:, 4].mean() / n_s:.2f} "
                  f"resamples {st[:, 1].mean():.1f}", file=sys.stderr, flush=True)
    all_stats()

    samples = [torch.empty((args.steps, Cg[gi], n), dtype=torch.uint8, device=g.dev) for gi in range(NG)]
    base = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    base.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        for gi, gr in enumerate(groups):
            with torch.cuda.stream(streams[gi]):
                gr.iteration(time_kernel=True)          # waits for this group's previous sweep only
                samples[gi][k].copy_(gr.s)              # retained allocation sample of this iteration
    torch.cuda.synchronize()
    if dist is not None:
        allsamp = torch.cat(samples, dim=1)
        gathered = torch.empty((world,) + tuple(allsamp.shape), dtype=torch.uint8, device=g.dev)
        dist.all_gather_into_tensor(gathered, allsamp)      # RCCL over xGMI: PSM input
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = all_stats()
    # per-launch durations and the time at least one sweep launch was running (HIP events on the launch streams)
    ivals = sorted((base.elapsed_time(e0), base.elapsed_time(e1)) for gr in groups for (e0, e1) in gr.events)
    launch_ms = [b - a for a, b in ivals]
    busy_ms, cur_a, cur_b = 0.0, None, None
    for a, b in ivals:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                busy_ms += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy_ms += (cur_b - cur_a) if cur_b is not None else 0.0

    t = torch.tensor([dt], dtype=torch.float64, device=g.dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    out = None
    if rank == 0:
        total_iters = args.steps * C * world
        kernel_ms = float(np.mean(launch_ms))                     # one launch = the chains of one group
        bytes_unit = workloads.algorithmic_bytes_per_obs_particle(w["kinds"], w["D"], N)
        alg_bytes_launch = float(bytes_unit) * n_s * P * (C / NG)
        achieved = alg_bytes_launch / (kernel_ms * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):       # PMC-measured HBM bytes per launch of this same command (profiles/README.md)
            tj = json.load(open(tpath))
            if tj.get("chains_per_gpu") == C and tj.get("groups", 1) == NG and tj.get("workload") == "cfg2" and args.scale == 1.0:
                traffic = tj["hbm_bytes_per_launch"]
        out = {
            "metric": "Gibbs iters/sec (and obs·particles/sec) at 1/2/4/8 GPUs vs CPU ref",
            "value": total_iters / dt,
            "unit": "Gibbs iters/s (aggregate over chains)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2: 3-mixture Gaussian 10000x50, K=1, N=20, P=1024, rho=0.25"
                       if args.scale == 1.0 else f"cfg2 scaled n={n}",
                       "chains_per_gpu": C, "chain_groups": NG, "chains_per_launch": C // NG,
                       "block_threads": g.sw.block_threads, "lds_bytes_per_chain": g.sw.lds_bytes,
                       "swept_obs_per_iter": n_s, "parallelism": f"chains x{world}"},
            "obs_particles_per_sec": total_iters * n_s * P / dt,
            "sweep_kernel_ms": kernel_ms,
            "sweep_busy_ms_per_step": busy_ms / args.steps,
            "launch_concurrency": float(np.sum(launch_ms)) / busy_ms if busy_ms > 0 else None,
            "sweep_only_iters_per_sec": args.steps * C * world / (busy_ms * 1e-3),
            "median_chain_iters_per_sec_hint": "see profiles/README.md (per-chain latency distribution)",
            "sweep_stats_last": {"ids_per_step": float(stats[:, 0].mean()) / n_s,
                                 "classes_per_step": float(stats[:, 4].mean()) / n_s,
                                 "resamples": float(stats[:, 1].mean()), "clones": float(stats[:, 2].mean()),
                                 "steps_fast_frac": float(stats[:, 5].sum()) / float(stats[:, 5:8].sum())},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_frac": (traffic / (kernel_ms * 1e-3) / HBM_PEAK) if traffic else None,
                         "note": "achieved = dense-model algorithmic bytes (SURVEY 8d: 19392 B per obs*particle) / "
                                 "kernel time. The kernel de-duplicates clusters and particle classes like the "
                                 "reference does, so it moves far fewer bytes than the dense model and frac > 1 "
                                 "is expected; `traffic` is the PMC-measured HBM bytes per sweep and `traffic_frac` that traffic / "
                                 "sweep time / peak. A sweep ends with its slowest chain; converged chains are bound by dependent "
                                 "latency, chains with many private clusters by HBM round trips (DESIGN.md section 6)."},
        }
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(f"[bench] GPU part done: {out['value']:.1f} iters/s", file=sys.stderr, flush=True)
        if not args.no_cpu:
            sec, secs = cpu_baseline(w, 4, args.cpu_iters)
            out["cpu_baseline"] = {"value": 1.0 / sec, "unit": "Gibbs iters/s (sweep only, one chain)",
                                   "cores": 1, "kind": "port",
                                   "sample": f"oracle sweep, same workload, {args.cpu_iters} iterations after 4 warm-up "
                                             f"iterations ({sum(secs):.1f} s of CPU work)"}
            if args.cpu_all:
                agg, cores = cpu_baseline_all_cores(args.scale, 4, max(4, args.cpu_iters // 3))
                out["cpu_baseline_all_cores"] = {"value": agg, "unit": "Gibbs iters/s (sweep only, one chain per core, aggregate)",
                                                 "cores": cores, "kind": "port",
                                                 "sample": f"{cores} oracle processes at once, {max(4, args.cpu_iters // 3)} iterations each after 4 warm-up"}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
