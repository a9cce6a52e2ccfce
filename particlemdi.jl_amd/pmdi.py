"""pmdi(): host driver with the reference's signature, argument checks and output files
(src/pmdi.jl:36-390).  Everything per iteration runs on the MI355X through the C ABI: the sweep
(Sweeper), the hyper-parameter updates, label alignment and shuffle (Gibbs: pmdi_gibbs_*), feature
selection; the CSV rows are written by the library's byte-compatible writer (pmdi_csv_*).  This is the
Python twin of the Julia glue in `particlemdi.jl_amd/julia/ParticleMDIHIP.jl`.
"""
import time

import numpy as np

from ._lib import KIND_BY_NAME, CsvWriter, Gibbs, Sweeper, format_float64


def jl_float(x):
    """A Float64 the way Julia's print/writedlm shows it (the library's formatter, pmdi_format_float64)."""
    return format_float64(x)


def gaussian_normalise(x):
    """gaussian_normalise! (gaussian_cluster.jl:85-94): median / half (median - q05) scaling.
    Julia's quantile() default is type 7, numpy's default."""
    x = np.array(x, dtype=np.float64)
    for d in range(x.shape[1]):
        mu = np.median(x[:, d])
        sigma = 0.5 * (mu - np.quantile(x[:, d], 0.05)) + np.finfo(np.float64).eps
        x[:, d] = (x[:, d] - mu) / sigma
    return x


def coerce_categorical(data):
    """coerce_categorical (categorical_cluster.jl:81-92): levels -> 1..n_unique per column,
    numbered by first appearance."""
    data = np.asarray(data)
    out = np.empty(data.shape, dtype=np.int64)
    for j in range(data.shape[1]):
        seen = {}
        for i, v in enumerate(data[:, j].tolist()):
            out[i, j] = seen.setdefault(v, len(seen) + 1)
    return out


def pmdi(dataFiles, dataTypes, N, particles, rho, iter, outputFile, thin=1, featureSelect=None,
         dataNames=None, seed=0, device=0, q1_mode=0, q2_mode=0, return_state=False):
    """Runs particleMDI on the given datasets (signature of src/pmdi.jl:36-40 plus the
    seed/device keywords this implementation needs).  dataTypes entries are
    "GaussianCluster" / "CategoricalCluster" / "NegBinomCluster" (or the short names)."""
    K = len(dataFiles)
    n_obs = int(dataFiles[0].shape[0])
    if dataNames is None:
        dataNames = [f"K{i}" for i in range(1, K + 1)]
    # src/pmdi.jl:50-55 (@assert in the reference; ValueError here so that `python -O` keeps them)
    def need(cond, msg):
        if not cond:
            raise ValueError(msg)
    need(len(dataTypes) == K, "Number of datatypes not equal to number of datasets")
    need(len(dataNames) == K, "Number of data names not equal to number of datasets")
    need(all(d.shape[0] == n_obs for d in dataFiles),
         "Datasets don't have same number of observations. Each row must correspond to the same underlying observational unit across datasets.")
    need(0 < rho < 1, "ρ must be between 0 and 1")
    need(1 < N <= n_obs, "Number of clusters must be greater than 1 and not greater than the number of observations")
    need(particles > 1, "Conditional particle filter requires 2 or more particles")
    for t in dataTypes:
        if t not in KIND_BY_NAME:
            raise TypeError(f"{t!r} has no device kernel; user-defined cluster types run on the "
                            "reference's own CPU loop (see INTEGRATION.md), not here")
    need(int(np.floor(rho * n_obs)) >= 1, "floor(ρ·n) must be >= 1 (the reference indexes order_obs[0] otherwise)")

    sweeper = Sweeper(dataFiles, dataTypes, N, particles, n_chains=1, seed=seed, device=device,
                      q1_mode=q1_mode, q2_mode=q2_mode)
    g = Gibbs(sweeper, rho=rho, feature_select=featureSelect is not None)      # src/pmdi.jl:59-66,95-96,106-110
    D = [int(d.shape[1]) for d in dataFiles]
    ffile = None
    if featureSelect is not None:                                               # :111-116
        ffile = CsvWriter(featureSelect, K, n_obs, dataNames, feature_D=D)
        ffile.flags(g.get(0)["flags"])
    out = CsvWriter(outputFile, K, n_obs, dataNames)                            # :147-156
    t_start = time.perf_counter()
    out.gibbs_row(g, 0, 0.0)                                                    # :158
    sweep_seconds = 0.0
    for it in range(1, iter + 1):
        t0 = time.perf_counter()
        g.iterate(1)                                                            # :165-375
        res = g.results()                                                       # synchronises; raises on a kernel-side error
        sweep_seconds += time.perf_counter() - t0
        ll = time.perf_counter() - t_start                                      # :377
        if it % thin == 0:
            out.gibbs_row(g, 0, ll)                                             # :379
            if ffile is not None:
                ffile.flags(g.get(0)["flags"])                                  # :381
    out.close()
    if ffile is not None:
        ffile.close()
    state = None
    if return_state:
        st = g.get(0)
        state = {"s": st["s"], "M": st["M"], "Phi": st["Phi"], "gamma": st["gamma"], "flags": st["flags"],
                 "sweep_seconds": sweep_seconds, "total_seconds": time.perf_counter() - t_start,
                 "last": {"stats": [dict(zip(("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes",
                                              "steps_fast", "steps_converted", "steps_fallback"), map(int, res["stats"][0])))] if iter > 0 else []}}
    g.close()
    sweeper.close()
    return state
