"""Shared helpers for the sweep tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kinds = [str(k) for k in z["kinds"]]
    data = [z[f"data{k}"] for k in range(len(kinds))]
    return z, data, kinds


def replay(z, data, kinds, sweeper_factory, check_featsel=True):
    """Run every recorded iteration through `sweeper_factory(data, kinds, N, P, seed, q1)` and
    compare with the recorded outputs.  The factory returns an object with .sweep(...) returning
    a dict with 's' (n, K), 'p_star', 'logweight', 'stats' dict, and .feature_select(it, s)."""
    N, P, seed, q1, n1 = int(z["N"]), int(z["P"]), int(z["seed"]), int(z["q1"]), int(z["n1"])
    flags = z["flags"] if "flags" in z.files else None
    sw = sweeper_factory(data, kinds, N, P, seed, q1)
    for it in range(1, int(z["iters"]) + 1):
        r = sw.sweep(it, z[f"s_in{it}"], z[f"order{it}"], n1, z[f"Pi{it}"], z[f"Phi{it}"], flags)
        assert (np.asarray(r["s"]) == z[f"s_out{it}"]).all(), f"allocations differ at iteration {it}"
        assert int(r["p_star"]) == int(z[f"p_star{it}"])
        assert np.allclose(r["logweight"], z[f"lw{it}"], rtol=1e-9, atol=1e-9)
        st = r["stats"]
        got = np.array([st[k] for k in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes")])
        assert (got == z[f"stats{it}"]).all(), f"counters differ at iteration {it}: {got} vs {z[f'stats{it}']}"
    if check_featsel:
        ff, fp = sw.feature_select(int(z["iters"]), z[f"s_out{int(z['iters'])}"])
        assert (np.asarray(ff) == z["featsel_flags"]).all()
        assert np.allclose(fp, z["featsel_prob"], rtol=1e-9, atol=1e-9)
    return sw


def t5_invariants(state, N, P, K, n_obs):
    """test/runtests.jl:147-162 on an exported state (particle (K,P,N), counts/cluster_n (K,cap))."""
    cap = N * P + 1
    for k in range(K):
        part = state["particle"][k]
        for j in range(P):
            assert state["cluster_n"][k, part[j] - 1].sum() == n_obs        # :147,:156
        cnt = np.bincount(part.ravel(), minlength=cap + 1)[1:cap + 1]
        assert (cnt == state["counts"][k][:cap]).all()                    # :149-153,:158-162


def check_work_counters(wk, rec, trace, N, kernel):
    """The device's work counters of one chain-sweep (wk: (K, 8)) against the oracle's per-step record.  Clusters evaluated
    (column 0) depends on which kernel swept the chain (pmdi_chain_swept_by): the settled-chain kernel (1) evaluates the reachable
    clusters in every step; the general kernel (0) every live id in a step whose (class, label) items outgrow its LDS tables; a
    chain handed over mid-sweep (2) lies between the two.  Columns met by resampling events and copy-on-write splits are the same
    everywhere."""
    ev_need, cols, splits = expected_work_counters(rec, trace, N, item_cap=None)
    ev_gen, _, _ = expected_work_counters(rec, trace, N)
    if kernel == 1:
        assert (wk[:, 0] == ev_need).all(), (wk[:, 0], ev_need)
    elif kernel == 0:
        assert (wk[:, 0] == ev_gen).all(), (wk[:, 0], ev_gen)
    else:
        assert (wk[:, 0] >= ev_need).all() and (wk[:, 0] <= ev_gen).all(), (wk[:, 0], ev_need, ev_gen)
    assert (wk[:, 5] == cols).all() and (wk[:, 6] == splits).all(), (wk, cols, splits)


def expected_work_counters(rec, trace, N, item_cap="auto"):
    """What the device's work counters [WK_EVAL, WK_COLS, WK_SPLITS] must equal, per dataset, from the oracle's per-step record
    (pmdi_oracle_debug_steps) and trace of the same sweep.  Clusters evaluated: the clusters the class leaders read at
    src/pmdi.jl:232 -- or every live id in a step whose (class, label) items outgrow the LDS tables (item_cap; None = never).
    Columns met by resampling events: the distinct columns of particle[:, :, k] before each event.  Copy-on-write splits: the
    growth of the distinct-column count inside the steps (a step never merges columns; a resampling only drops them)."""
    if item_cap == "auto":
        item_cap = 384 if N > 32 else 256
    ncls, need, cols_pre, cols_post, maxid = rec[:, :, 0], rec[:, :, 1], rec[:, :, 4], rec[:, :, 5], rec[:, :, 6]
    ev = need if item_cap is None else np.where(ncls * N <= item_cap, need, maxid)
    res = trace[:, 1] > 0
    prev = np.vstack([np.ones((1, rec.shape[1]), dtype=rec.dtype), cols_post[:-1]])
    return ev.sum(axis=0), cols_pre[res].sum(axis=0), (cols_pre - prev).sum(axis=0)
