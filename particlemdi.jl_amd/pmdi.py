"""pmdi(): host driver with the reference's signature, asserts and CSV output
(src/pmdi.jl:36-390).  The per-iteration sweep runs on the MI355X through the
C ABI (Sweeper); hyper-parameter updates and label alignment stay on the host
(hypers.py).  This is the Python twin of the Julia glue in
`particlemdi.jl_amd/julia/ParticleMDIHIP.jl`.
"""
import time
from decimal import Decimal

import numpy as np

from ._lib import KIND_BY_NAME, Sweeper
from .hypers import HyperState, phi_lab


def jl_float(x):
    """Format a Float64 the way Julia's print/writedlm does: shortest round-trip digits,
    fixed notation for 1e-5 <= |x| < 1e6, otherwise d.ddde±x."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if np.signbit(x) else "0.0"
    sign, dig, exp = Decimal(repr(x)).as_tuple()
    digits = "".join(map(str, dig)).lstrip("0")
    stripped = digits.rstrip("0")
    exp += len(digits) - len(stripped)
    digits = stripped or "0"
    e10 = len(digits) - 1 + exp
    if -5 < e10 < 6:
        if e10 >= 0:
            out = digits[:e10 + 1].ljust(e10 + 1, "0") + "." + (digits[e10 + 1:] or "0")
        else:
            out = "0." + "0" * (-e10 - 1) + digits
    else:
        out = f"{digits[0]}.{digits[1:] or '0'}e{e10}"
    return ("-" if sign else "") + out


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
         dataNames=None, seed=0, device=0, q1_mode=0, return_state=False):
    """Runs particleMDI on the given datasets (signature of src/pmdi.jl:36-40 plus the
    seed/device keywords this implementation needs).  dataTypes entries are
    "GaussianCluster" / "CategoricalCluster" / "NegBinomCluster" (or the short names)."""
    K = len(dataFiles)
    n_obs = int(dataFiles[0].shape[0])
    if dataNames is None:
        dataNames = [f"K{i}" for i in range(1, K + 1)]
    # src/pmdi.jl:50-55
    assert len(dataTypes) == K, "Number of datatypes not equal to number of datasets"
    assert len(dataNames) == K, "Number of data names not equal to number of datasets"
    assert all(d.shape[0] == n_obs for d in dataFiles), \
        "Datasets don't have same number of observations. Each row must correspond to the same underlying observational unit across datasets."
    assert 0 < rho < 1, "ρ must be between 0 and 1"
    assert 1 < N <= n_obs, "Number of clusters must be greater than 1 and not greater than the number of observations"
    assert particles > 1, "Conditional particle filter requires 2 or more particles"
    for t in dataTypes:
        if t not in KIND_BY_NAME:
            raise TypeError(f"{t!r} has no device kernel; user-defined cluster types run on the "
                            "reference's own CPU loop (see INTEGRATION.md), not here")
    n1 = int(np.floor(rho * n_obs))
    assert n1 >= 1, "floor(ρ·n) must be >= 1 (the reference indexes order_obs[0] otherwise)"

    rng = np.random.default_rng(seed)
    hy = HyperState(n_obs, N, K, rng)
    sweeper = Sweeper(dataFiles, dataTypes, N, particles, n_chains=1, seed=seed, device=device,
                      q1_mode=q1_mode)
    D = [int(d.shape[1]) for d in dataFiles]
    # feature flags (src/pmdi.jl:106-116)
    if featureSelect is None:
        flags = np.ones(sum(D), dtype=np.uint8)
        ffile = None
    else:
        flags = (rng.random(sum(D)) < 0.5).astype(np.uint8)
        ffile = open(featureSelect, "w")
        ffile.write(",".join(f"{dataNames[k]}_d{d}" for k in range(K) for d in range(1, D[k] + 1)) + "\n")
        ffile.write(",".join("true" if f else "false" for f in flags) + "\n")

    pairs = phi_lab(K)
    header = ([f"MassParameter_{k}" for k in range(1, K + 1)]
              + [f"phi_{a + 1}_{b + 1}" for a, b in pairs]
              + ["ll"]
              + [f"{dataNames[k]}_n{i}" for k in range(K) for i in range(1, n_obs + 1)])
    out = open(outputFile, "w")
    out.write(",".join(header) + "\n")

    def write_row(ll):
        row = [jl_float(v) for v in hy.M] + [jl_float(v) for v in hy.Phi] + [jl_float(ll)]
        row += [jl_float(v) for v in hy.s.T.reshape(-1)]       # s[1:(n*K)], column-major
        out.write(",".join(row) + "\n")

    t_start = time.perf_counter()
    write_row(0.0)                                               # src/pmdi.jl:158
    order_obs = np.arange(1, n_obs + 1)
    sweep_seconds = 0.0
    last = None
    for it in range(1, iter + 1):
        rng.shuffle(order_obs)                                   # :172
        Pi = hy.step_pmdi_order()                                # :176-185
        t0 = time.perf_counter()
        last = sweeper.sweep(it, hy.s[None], order_obs[None], n1, Pi[None], hy.Phi[None],
                             flags=flags[None], lw_init=0.0 if it == 1 else 1.0)   # :165-171,188-350
        if featureSelect is not None:
            fl, _ = sweeper.feature_select(it, last["s"])        # :354-370
            flags = fl[0].copy()
        sweep_seconds += time.perf_counter() - t0
        hy.s[:] = last["s"][0]                                   # :373
        hy.align_labels()                                        # :375
        ll = time.perf_counter() - t_start                       # :377
        if it % thin == 0:
            write_row(ll)
            if ffile is not None:
                ffile.write(",".join("true" if f else "false" for f in flags) + "\n")
    out.close()
    if ffile is not None:
        ffile.close()
    sweeper.close()
    if return_state:
        return {"s": hy.s.copy(), "M": hy.M.copy(), "Phi": hy.Phi.copy(), "gamma": hy.gamma.copy(),
                "sweep_seconds": sweep_seconds, "total_seconds": time.perf_counter() - t_start,
                "flags": flags, "last": last}
    return None
