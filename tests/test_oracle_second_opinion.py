"""The C oracle against a second restatement of the sweep written separately, in the reference's own style
(tests/_py_sweep.py: mutable cluster objects, deepcopy, 1-based arrays, literal loops of src/pmdi.jl:165-350).
Same counter-based uniforms at the same draw sites => the two must agree in every allocation, the picked particle,
the particle -> cluster tables, the reference counts and (to rounding of nothing: same libm, same order) the weights."""
import numpy as np
import pytest

import _py_sweep as PS
from conftest import make_mixed, random_hypers


def _run_both(O, data, kinds, N, P, seed, iters, n1, q1=0, q2=0, flags=None, rng=None):
    K, n = len(data), data[0].shape[0]
    o = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1, q2_mode=q2)
    uni = lambda it, pos, k, p, site: O.uniform(seed, it, pos, k, p, site)
    pdata = [np.asarray(d) if kd == "gaussian" else np.asarray(d, dtype=np.int64) for d, kd in zip(data, kinds)]
    pflags = [np.ones(d.shape[1], dtype=bool) if flags is None else np.asarray(f, dtype=bool) for d, f in zip(data, flags or data)]
    s = rng.integers(1, N + 1, size=(n, K))
    total = {"n_resamples": 0, "n_clones": 0}
    for it in range(1, iters + 1):
        Pi, Phi = random_hypers(rng, N, K)
        order = rng.permutation(n) + 1
        r = o.sweep(it, s, order, n1, Pi, Phi, flags=None if flags is None else [np.asarray(f, dtype=np.uint8) for f in flags])
        s1 = np.zeros((n + 1, K + 1), dtype=np.int64); s1[1:, 1:] = s
        Pi1 = np.zeros((N + 1, K + 1)); Pi1[1:, 1:] = Pi
        s_py, p_star, lw, cnt, state = PS.sweep(pdata, kinds, N, P, s1.tolist(), order.tolist(), n1, Pi1.tolist(), list(np.atleast_1d(Phi)),
                                                pflags, uni, it, 0.0 if it == 1 else 1.0, q1_mode=q1, q2_mode=q2)
        assert np.array_equal(s_py, r["s"]), f"iteration {it}: allocations differ"
        assert p_star == r["p_star"]
        assert np.array_equal(np.array(lw), r["logweight"]), np.abs(np.array(lw) - r["logweight"]).max()
        st = r["stats"]
        assert cnt["n_resamples"] == st["n_resamples"] and cnt["sum_classes"] == st["sum_classes"]
        assert cnt["n_operations"] == st["n_operations"]
        ex = o.export()
        for k in range(K):
            part = np.array([[state["particle"][k + 1][nn][p] for nn in range(1, N + 1)] for p in range(1, P + 1)])
            assert np.array_equal(part, ex["particle"][k])
            mx = int(part.max())
            assert mx == ex["max_id"][k]
            assert np.array_equal(np.array(state["counts"][k + 1][1:mx + 1]), ex["counts"][k][:mx])
            assert np.array_equal(np.array([state["clusters"][k + 1][c].n for c in range(1, mx + 1)]), ex["cluster_n"][k][:mx])
        total["n_resamples"] += cnt["n_resamples"]; total["n_clones"] += cnt["n_clones"]
        s = r["s"]
    o.close()
    return total


@pytest.mark.parametrize("q1,q2", [(0, 0), (1, 0), (0, 1)])
def test_mixed_types_three_datasets(O, q1, q2):
    rng = np.random.default_rng(100 + 10 * q1 + q2)
    data, kinds = make_mixed(rng, n=48)
    tot = _run_both(O, data, kinds, N=5, P=16, seed=77 + q1, iters=3, n1=12, q1=q1, q2=q2, rng=rng)
    assert tot["n_resamples"] > 0 and tot["n_clones"] > 0


def test_gaussian_pair_with_feature_flags(O):
    rng = np.random.default_rng(31)
    z = rng.integers(0, 2, 60)
    data = [rng.normal(size=(60, 6)) + 3.0 * z[:, None], rng.normal(size=(60, 4)) - 2.0 * z[:, None]]
    flags = [np.array([1, 0, 1, 1, 0, 1]), np.array([1, 1, 0, 1])]
    tot = _run_both(O, data, ["gaussian", "gaussian"], N=6, P=32, seed=5, iters=2, n1=15, flags=flags, rng=rng)
    assert tot["n_resamples"] > 0


def test_single_dataset_no_phi(O):
    rng = np.random.default_rng(32)
    data = [rng.poisson(3.0, size=(40, 5)).astype(np.int64)]
    _run_both(O, data, ["negbinom"], N=4, P=24, seed=11, iters=2, n1=2, rng=rng)


def test_more_particles_than_a_cumsum_block(O):
    """P > 128 puts draw_partstar's cumsum (src/misc.jl:29) on Base's pairwise path."""
    rng = np.random.default_rng(33)
    z = rng.integers(0, 3, 30)
    data = [rng.normal(size=(30, 3)) + 2.0 * z[:, None]]
    tot = _run_both(O, data, ["gaussian"], N=3, P=160, seed=2, iters=1, n1=8, rng=rng)
    assert tot["n_resamples"] > 0


def test_more_labels_than_a_cumsum_block(O):
    """N > 128 puts the mutation CDF's cumsum (src/pmdi.jl:240) on Base's pairwise path as well: c[1] = e1, the other N - 1 elements
    split once into two leaves, the right leaf carried by e1 + total(left)."""
    rng = np.random.default_rng(34)
    z = rng.integers(0, 3, 170)
    data = [rng.normal(size=(170, 3)) + 2.0 * z[:, None]]
    _run_both(O, data, ["gaussian"], N=150, P=8, seed=3, iters=1, n1=20, rng=rng)


@pytest.mark.parametrize("seed", range(12))
def test_random_small_configurations(O, seed):
    rng = np.random.default_rng(9000 + seed)
    K, N, P = int(rng.integers(1, 4)), int(rng.integers(2, 8)), int(rng.integers(2, 41))
    n = int(rng.integers(12, 70))
    z = rng.integers(0, 3, n)
    data, kinds = [], []
    for _ in range(K):
        kd = ["gaussian", "categorical", "negbinom"][int(rng.integers(0, 3))]
        D = int(rng.integers(1, 7))
        if kd == "gaussian":
            data.append(rng.normal(size=(n, D)) + 2.0 * z[:, None])
        elif kd == "categorical":
            data.append(1 + (z[:, None] + rng.integers(0, 2, (n, D))) % int(rng.integers(2, 5)))
        else:
            data.append(rng.poisson(1.0 + 3.0 * z[:, None], size=(n, D)).astype(np.int64))
        kinds.append(kd)
    flags = [(rng.random(d.shape[1]) < 0.8).astype(np.uint8) for d in data] if seed % 2 else None
    _run_both(O, data, kinds, N=N, P=P, seed=seed, iters=2, n1=max(2, int(rng.integers(2, n // 2))), q1=seed % 2, q2=(seed // 2) % 2,
              flags=flags, rng=rng)


@pytest.mark.parametrize("seed", range(8))
def test_column_table_shadows_the_literal_table(O, seed):
    """The device keeps particle[:, :, k] by distinct column.  tests/_py_columns.py restates that scheme;
    here it shadows the literal N x P table of the Python sweep through every step and resampling event -- equal tables,
    equal occupancies, no empty column, never more than P -- with the arbitrary "which group keeps the column" choice shuffled."""
    import _py_columns as PC
    rng = np.random.default_rng(700 + seed)
    K, N, P, n = int(rng.integers(1, 4)), int(rng.integers(2, 7)), int(rng.integers(2, 33)), int(rng.integers(20, 60))
    z = rng.integers(0, 3, n)
    data = [rng.normal(size=(n, 3)) + 2.0 * z[:, None] for _ in range(K)]
    kinds = ["gaussian"] * K
    uni = lambda it, pos, k, p, site: O.uniform(seed, it, pos, k, p, site)
    flags = [np.ones(3, dtype=bool)] * K
    s = rng.integers(1, N + 1, size=(n, K))
    seen_cols = 0
    for it in range(1, 3):
        Pi, Phi = random_hypers(rng, N, K)
        order = rng.permutation(n) + 1
        s1 = np.zeros((n + 1, K + 1), dtype=np.int64); s1[1:, 1:] = s
        Pi1 = np.zeros((N + 1, K + 1)); Pi1[1:, 1:] = Pi
        sh = PC.ColumnShadow(K, N, P, order_seed=seed)
        s, _, _, cnt, _ = PS.sweep(data, kinds, N, P, s1.tolist(), order.tolist(), max(2, n // 4), Pi1.tolist(), list(np.atleast_1d(Phi)),
                                    flags, uni, it, 0.0 if it == 1 else 1.0, q1_mode=seed % 2, shadow=sh)
        seen_cols = max(seen_cols, sh.max_cols)
    assert 1 <= seen_cols <= P
