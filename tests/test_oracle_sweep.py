"""The restated sweep: the reference's T5 invariants (test/runtests.jl:136-162), golden vectors,
determinism, and equivalence of the oracle's own switches."""
import numpy as np
import pytest

from _cases import golden_cases, load_golden, replay, t5_invariants
from conftest import make_mixed, random_hypers


class OracleRunner:
    def __init__(self, O, data, kinds, N, P, seed, q1, faithful=0):
        self.o = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1, faithful_cost=faithful)
        self.D = [d.shape[1] for d in data]

    def sweep(self, it, s, order, n1, Pi, Phi, flags):
        fl = None if flags is None else np.split(flags, np.cumsum(self.D)[:-1])
        return self.o.sweep(it, s, order, n1, Pi, Phi, flags=fl)

    def feature_select(self, it, s):
        ff, fp = self.o.feature_select(it, s)
        return np.concatenate(ff), np.concatenate(fp)


@pytest.mark.parametrize("case", golden_cases())
def test_golden(O, case):
    z, data, kinds = load_golden(case)
    replay(z, data, kinds, lambda d, k, N, P, seed, q1: OracleRunner(O, d, k, N, P, seed, q1))


@pytest.mark.parametrize("P,iters", [(2, 1), (1024, 6)])
def test_T5_invariants(O, P, iters):
    # test/runtests.jl:136-162: 3 Gaussian datasets 100x16, N=10, rho=0.25
    rng = np.random.default_rng(5)
    data = [np.vstack([rng.normal(2, 1, (50, 16)), rng.normal(-2, 1, (50, 16))]) for _ in range(3)]
    N, K, n = 10, 3, 100
    o = O.Oracle(data, ["gaussian"] * 3, N, P, seed=9, faithful_cost=1)
    s = rng.integers(1, N + 1, size=(n, K))
    for it in range(1, iters + 1):
        Pi, Phi = random_hypers(rng, N, K)
        r = o.sweep(it, s, rng.permutation(n) + 1, 25, Pi, Phi)
        s = r["s"]
        assert s.min() >= 1 and s.max() <= N
    t5_invariants(o.export(), N, P, K, n)


def test_T5_invariants_in___pmdi_mode(O):
    """test/runtests.jl:155 runs __pmdi (history permuted on resample: src/__pmdi.jl:285 = q2_mode 1) with
    P = 1024 for 100 iterations on 3 x (100 x 16); the same shape here, hyper-parameters from the restated updates."""
    rng = np.random.default_rng(15)
    data = [np.vstack([rng.normal(2, 1, (50, 16)), rng.normal(-2, 1, (50, 16))]) for _ in range(3)]
    N, K, n, P = 10, 3, 100, 1024
    o = O.Oracle(data, ["gaussian"] * 3, N, P, seed=9, q2_mode=1)
    hy = O.Hypers(n, N, K, seed=9)
    resamples = 0
    for it in range(1, 101):
        Pi = hy.step(it)
        r = o.sweep(it, np.array(hy.s), np.array(hy.order), 25, Pi, hy.Phi)
        hy.s[:] = r["s"]
        hy.align_labels(it)
        resamples += r["stats"]["n_resamples"]
        if it in (1, 2, 50, 100):
            t5_invariants(o.export(), N, P, K, n)
    assert resamples > 0
    s = np.array(hy.s)
    assert s.min() >= 1 and s.max() <= N
    hy.close()


def test_determinism_and_cost_switch(O):
    rng = np.random.default_rng(6)
    data, kinds = make_mixed(rng, 150)
    N, P, K, n = 8, 64, 3, 150
    s0 = rng.integers(1, N + 1, size=(n, K))
    order = rng.permutation(n) + 1
    Pi, Phi = random_hypers(rng, N, K)
    outs = []
    for faithful in (0, 1, 0):
        o = O.Oracle(data, kinds, N, P, seed=3, faithful_cost=faithful)
        outs.append(o.sweep(1, s0, order, 37, Pi, Phi, trace=True))
    for r in outs[1:]:
        assert (r["s"] == outs[0]["s"]).all() and r["p_star"] == outs[0]["p_star"]
        assert (r["trace"] == outs[0]["trace"]).all()
        assert {k: v for k, v in r["stats"].items() if k != "seconds"} == \
               {k: v for k, v in outs[0]["stats"].items() if k != "seconds"}
    # a different seed gives a different trajectory
    o = O.Oracle(data, kinds, N, P, seed=4)
    assert (o.sweep(1, s0, order, 37, Pi, Phi)["s"] != outs[0]["s"]).any()


def test_known_prefix_is_kept_and_reference_particle(O):
    rng = np.random.default_rng(7)
    data, kinds = make_mixed(rng, 120)
    N, P, K, n, n1 = 6, 16, 3, 120, 30
    s0 = rng.integers(1, N + 1, size=(n, K))
    order = rng.permutation(n) + 1
    Pi, Phi = random_hypers(rng, N, K)
    o = O.Oracle(data, kinds, N, P, seed=1)
    r = o.sweep(1, s0, order, n1, Pi, Phi)
    pre = order[:n1 - 1] - 1
    assert (r["s"][pre] == s0[pre]).all()                 # sstar[:, i, k] .= s[i, k] (src/pmdi.jl:204)


def test_q_modes_run_and_differ(O):
    rng = np.random.default_rng(8)
    data, kinds = make_mixed(rng, 150)
    N, P, K, n = 8, 64, 3, 150
    s0 = rng.integers(1, N + 1, size=(n, K))
    order = rng.permutation(n) + 1
    Pi, Phi = random_hypers(rng, N, K)
    base = O.Oracle(data, kinds, N, P, seed=3).sweep(1, s0, order, 37, Pi, Phi)
    q1 = O.Oracle(data, kinds, N, P, seed=3, q1_mode=1).sweep(1, s0, order, 37, Pi, Phi)
    q2 = O.Oracle(data, kinds, N, P, seed=3, q2_mode=1).sweep(1, s0, order, 37, Pi, Phi)
    assert q1["stats"]["sum_classes"] >= base["stats"]["sum_classes"]    # no class aliasing -> more CDFs
    assert q2["stats"]["n_resamples"] == base["stats"]["n_resamples"]    # Q2 only changes the returned history
    for r in (q1, q2):
        assert r["s"].min() >= 1 and r["s"].max() <= N


def test_rejects_bad_inputs(O):
    rng = np.random.default_rng(9)
    x = rng.normal(size=(20, 2))
    with pytest.raises(ValueError):
        O.Oracle([x], ["gaussian"], 1, 8)             # N > 1 (src/pmdi.jl:54)
    with pytest.raises(ValueError):
        O.Oracle([x], ["gaussian"], 4, 1)             # particles > 1 (:55)
    o = O.Oracle([x], ["gaussian"], 4, 8)
    Pi, Phi = random_hypers(rng, 4, 1)
    with pytest.raises(RuntimeError):
        o.sweep(1, rng.integers(1, 5, size=(20, 1)), np.arange(1, 21), 0, Pi, Phi)   # n1 >= 1 (Q8)


def test_distinct_column_counter_matches_the_exported_table(O):
    """The analysis hook behind scripts/column_stats.py (distinct columns of particle[:, :, k] per step) against a direct count
    on the exported table after the last step of a sweep that ends without a resampling at its last observation."""
    import ctypes as C
    rng = np.random.default_rng(21)
    data, kinds = make_mixed(rng, 120)
    N, P, K, n, n1 = 6, 64, 3, 120, 30
    o = O.Oracle(data, kinds, N, P, seed=3)
    buf = np.zeros((n - n1 + 1, K), dtype=np.int64)
    o.L.pmdi_oracle_debug_columns.argtypes = [C.c_void_p, C.c_void_p]
    o.L.pmdi_oracle_debug_columns(o.h, buf.ctypes.data)
    s = rng.integers(1, N + 1, size=(n, K))
    for it in range(1, 4):
        Pi, Phi = random_hypers(rng, N, K)
        r = o.sweep(it, s, rng.permutation(n) + 1, n1, Pi, Phi, trace=True)
        s = r["s"]
        assert buf.min() >= 1 and buf.max() <= P
        if r["trace"][-1, 1] == 0:          # no resampling after the last step: the exported table is the one that was counted
            part = o.export()["particle"]
            assert [len({tuple(col) for col in part[k]}) for k in range(K)] == list(buf[-1])
    o.L.pmdi_oracle_debug_columns(o.h, None)


def test_per_step_record_adds_up_to_the_sweep_counters(O):
    """pmdi_oracle_debug_steps (what the device's work counters are checked against): its per-step columns add up to the sweep's
    own counters, its column counts equal the older per-observation hook's, and a unanimous step has one chosen cluster."""
    import ctypes as C
    rng = np.random.default_rng(22)
    data, kinds = make_mixed(rng, 160)
    N, P, K, n, n1 = 7, 128, 3, 160, 40
    o = O.Oracle(data, kinds, N, P, seed=5)
    rec = o.debug_steps(n - n1 + 1)
    cols = np.zeros((n - n1 + 1, K), dtype=np.int64)
    o.L.pmdi_oracle_debug_columns(o.h, cols.ctypes.data)
    s = rng.integers(1, N + 1, size=(n, K))
    for it in range(1, 4):
        Pi, Phi = random_hypers(rng, N, K)
        r = o.sweep(it, s, rng.permutation(n) + 1, n1, Pi, Phi, trace=True)
        s = r["s"]
        st, tr = r["stats"], r["trace"]
        up, _ = o.work()
        assert rec[:, :, 0].sum() == st["sum_classes"] and rec[:, :, 6].sum() == st["n_operations"]
        assert (rec[:, :, 2].sum(axis=0) == up).all() and rec[:, :, 3].sum() == st["n_clones"]
        assert (rec[:, :, 4] == cols).all()
        assert (rec[:, :, 1] >= 1).all() and (rec[:, :, 1] <= rec[:, :, 6]).all()          # reachable clusters: some, never more than live ids
        assert (rec[:, :, 1] <= rec[:, :, 0] * N).all()
        assert (rec[rec[:, :, 7] == 1][:, 2] == 1).all()                                      # unanimous: one chosen cluster
        res = tr[:, 1] > 0
        assert (rec[~res][:, :, 5] == rec[~res][:, :, 4]).all() and (rec[res][:, :, 5] <= rec[res][:, :, 4]).all()
        # the classes column is the number of classes at the START of a step = the trace's count after the previous one
        assert (rec[1:, :, 0] == tr[:-1, 2 + K:2 + 2 * K]).all()
    o.L.pmdi_oracle_debug_columns(o.h, None)
    o.debug_steps(0)
