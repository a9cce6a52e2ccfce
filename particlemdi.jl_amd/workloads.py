"""Synthetic workloads of BASELINE.json's configs.  Every
parameter BASELINE.json leaves open (D, L, N of the headline, rho) is the
survey's stated assumption and is repeated in the returned dict."""
import numpy as np

from .pmdi import gaussian_normalise


def _gauss(rng, z, D, n_comp, informative=None):
    means = rng.choice([-2.0, 0.0, 2.0], size=(n_comp, D))
    if informative is not None:
        means[:, informative:] = 0.0
    x = rng.normal(size=(z.size, D)) + means[z]
    return gaussian_normalise(x)


def _cat(rng, z, D, L, n_comp):
    probs = rng.dirichlet(np.full(L, 0.5), size=(n_comp, D))
    u = rng.random((z.size, D))
    cdf = np.cumsum(probs[z], axis=2)
    x = 1 + (u[:, :, None] > cdf).sum(axis=2)
    x = np.minimum(x, L)
    x[0, :] = L  # every column reaches level L, so nlevels is the same for every feature
    return x.astype(np.int64)


def _negbin(rng, z, D, n_comp):
    p = rng.beta(2.0, 2.0, size=(n_comp, D)) * 0.8 + 0.1
    return (rng.geometric(p[z]) - 1).astype(np.int64)


def make(cfg, scale=1.0, seed=None):
    """cfg in {"cfg1".."cfg5", "HL"}; scale < 1 shrinks n (and only n) for CPU-sized runs."""
    spec = {
        "cfg1": dict(n=150, K=1, kinds=["gaussian"], D=[4], N=10, P=32, comps=3, seed=1),
        "cfg2": dict(n=10000, K=1, kinds=["gaussian"], D=[50], N=20, P=1024, comps=3, seed=1),
        "cfg3": dict(n=5000, K=2, kinds=["gaussian", "categorical"], D=[50, 20], N=30, P=1024, comps=3, seed=2),
        "cfg4": dict(n=10000, K=4, kinds=["gaussian", "gaussian", "categorical", "negbinom"],
                     D=[50, 50, 20, 30], N=50, P=2048, comps=4, seed=100),
        "cfg5": dict(n=20000, K=3, kinds=["gaussian"] * 3, D=[200] * 3, N=50, P=4096, comps=5, seed=5,
                     informative=50),
        "HL": dict(n=10000, K=4, kinds=["gaussian"] * 4, D=[50] * 4, N=20, P=1024, comps=3, seed=9),
    }[cfg]
    n = max(int(round(spec["n"] * scale)), 4 * spec["N"])
    rng = np.random.default_rng(spec["seed"] if seed is None else seed)
    z = rng.integers(0, spec["comps"], n)
    data = []
    for kind, D in zip(spec["kinds"], spec["D"]):
        if kind == "gaussian":
            data.append(_gauss(rng, z, D, spec["comps"], spec.get("informative")))
        elif kind == "categorical":
            data.append(_cat(rng, z, D, 4, spec["comps"]))
        else:
            data.append(_negbin(rng, z, D, spec["comps"]))
    return dict(name=cfg, data=data, kinds=spec["kinds"], N=spec["N"], P=spec["P"], rho=0.25, n=n,
                K=spec["K"], D=spec["D"], truth=z)


def algorithmic_bytes_per_obs_particle(kinds, D, N):
    """Dense-model HBM bytes per swept observation x particle."""
    tot = 0
    for kind, d in zip(kinds, D):
        if kind == "gaussian":
            tot += 16 * d * (N + 4) + 8 * N + 32
        else:
            tot += 8 * d * (N + 2) + 8 * N + 32
    return tot
