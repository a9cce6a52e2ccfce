"""Generates the golden sweep vectors under tests/golden/.

The reference (Julia) cannot run in this container or on the GPU box and its own tests hold no
vectors or seeds, so these fixtures come from the oracle (oracle/pmdi_oracle.c,
itself pinned by test/runtests.jl T1/T2/T5 and scipy known answers).  They freeze the oracle's
behaviour -- any later change to the oracle or to the HIP path that alters results shows up here --
and they give the GPU tests inputs/outputs that do not depend on the oracle being rebuilt.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as G  # noqa: E402

O = G.load_oracle()


def t5_data(rng):
    # the reference's integration problem: test/runtests.jl:138-144
    return [np.vstack([rng.normal(2, 1, (50, 16)), rng.normal(-2, 1, (50, 16))]) for _ in range(3)]


def mixed_data(rng, n=160):
    z = rng.integers(0, 3, n)
    g = rng.normal(size=(n, 6)) + 2.5 * (z[:, None] - 1)
    c = 1 + (rng.random((n, 5)) < (0.15 + 0.35 * z[:, None])).astype(np.int64) + (z[:, None] == 2) * rng.integers(0, 2, (n, 5))
    nb = rng.geometric(0.2 + 0.25 * z[:, None], size=(n, 4)) - 1
    return [g, c, nb]


def run_case(name, data, kinds, N, P, iters, seed, q1=0, flags=None):
    rng = np.random.default_rng(seed)
    n, K = data[0].shape[0], len(data)
    orc = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1)
    s = rng.integers(1, N + 1, size=(n, K))
    n1 = n // 4
    rec = {"N": N, "P": P, "seed": seed, "q1": q1, "n1": n1, "iters": iters, "kinds": np.array(kinds)}
    for k, d in enumerate(data):
        rec[f"data{k}"] = d
    if flags is not None:
        rec["flags"] = flags
    fl_list = None if flags is None else np.split(flags, np.cumsum([d.shape[1] for d in data])[:-1])
    for it in range(1, iters + 1):
        order = rng.permutation(n) + 1
        Pi = rng.gamma(1.0 / N, 1.0, size=(N, K)) + 1e-12
        Pi /= Pi.sum(0)
        Phi = rng.gamma(1.0, 0.2, size=max(1, K * (K - 1) // 2))
        r = orc.sweep(it, s, order, n1, Pi, Phi, flags=fl_list)
        rec[f"s_in{it}"] = s.copy(); rec[f"order{it}"] = order; rec[f"Pi{it}"] = Pi; rec[f"Phi{it}"] = Phi
        rec[f"s_out{it}"] = r["s"]; rec[f"p_star{it}"] = r["p_star"]; rec[f"lw{it}"] = r["logweight"]
        rec[f"stats{it}"] = np.array([r["stats"][k] for k in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes")])
        s = r["s"]
    ff, fp = orc.feature_select(iters, s)
    rec["featsel_flags"] = np.concatenate(ff); rec["featsel_prob"] = np.concatenate(fp)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, "written")


if __name__ == "__main__":
    rng = np.random.default_rng(20261003)
    d3 = t5_data(rng)
    run_case("t5_P2", d3, ["gaussian"] * 3, 10, 2, 2, 101)
    run_case("t5_P64", d3, ["gaussian"] * 3, 10, 64, 2, 102)
    dm = mixed_data(rng)
    run_case("mixed_P128", dm, ["gaussian", "categorical", "negbinom"], 9, 128, 2, 103)
    run_case("mixed_P128_q1", dm, ["gaussian", "categorical", "negbinom"], 9, 128, 1, 104, q1=1)
    fl = (np.random.default_rng(5).random(6 + 5 + 4) < 0.6).astype(np.uint8)
    run_case("mixed_P96_flags", dm, ["gaussian", "categorical", "negbinom"], 9, 96, 2, 105, flags=fl)
