"""The settled-chain kernel's device code (particlemdi.jl_amd/csrc/pmdi_sweep2_body.h) run on the CPU in the lock-step workgroup
emulator of tests/emu/ against the oracle: allocations, picked particle, log-weights, counters, per-observation trace, work
counters and the exported state, on seeded problems that exercise unanimous and split steps, clones, resampling, several
particle classes, the arena fallback of the LDS tables (tiny cols_l / idcap) and feature flags.  Logic only: memory ordering and
performance are the GPU tests' business."""
import numpy as np
import pytest

from _cases import expected_work_counters, t5_invariants
from conftest import random_hypers


def _gauss(rng, n, K, D=12, sep=3.0):
    z = rng.integers(0, 3, n)
    return [rng.normal(size=(n, D + k)) + sep * (z[:, None] - 1) for k in range(K)], z


def _mixed(rng, n, kinds, sep=3.0):
    """Datasets of the given cluster types sharing one planted 3-cluster structure."""
    z = rng.integers(0, 3, n)
    data = []
    for j, kind in enumerate(kinds):
        if kind == "gaussian":
            data.append(rng.normal(size=(n, 9 + j)) + sep * (z[:, None] - 1))
        elif kind == "categorical":      # levels 1..4, per-cluster level probabilities
            pr = rng.dirichlet(0.5 * np.ones(4), size=(3, 7 + j))
            x = np.empty((n, 7 + j), dtype=np.int64)
            for q in range(7 + j):
                for c in range(3):
                    m = z == c
                    x[m, q] = 1 + rng.choice(4, size=int(m.sum()), p=pr[c, q])
            data.append(x)
        else:                            # geometric counts with a per-cluster rate
            data.append(rng.geometric(0.15 + 0.3 * z[:, None], size=(n, 6 + j)) - 1)
    return data, z


def _compare(O, data, N, P, iters, seed, n1, q1=0, flags=None, cols_l=64, idcap=128, settle=False, truth=None, scramble=0.05, allow_requeue=0, variant="", kinds=None, cls=16, cdfl=0):
    from _emu import EmuSweeper
    rng = np.random.default_rng(seed)
    n, K = data[0].shape[0], len(data)
    kinds = kinds or ["gaussian"] * K
    e = EmuSweeper(data, N, P, seed=seed, q1_mode=q1, cols_l=cols_l, idcap=idcap, variant=variant, kinds=kinds, cls=cls, cdfl=cdfl)
    o = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1)
    rec = o.debug_steps(n - n1 + 1)
    requeued = []
    s = rng.integers(1, N + 1, size=(n, K))
    if truth is not None:                       # a mid-chain state: the planted clustering with a few labels scrambled
        s = np.repeat((truth + 1)[:, None], K, axis=1)
        idx = rng.random((n, K)) < scramble
        s[idx] = rng.integers(1, N + 1, size=int(idx.sum()))
    for it in range(1, iters + 1):
        order = rng.permutation(n) + 1
        Pi, Phi = random_hypers(rng, N, K)
        if settle:
            Pi[:3] += float(settle); Pi /= Pi.sum(0)
        ro = o.sweep(it, s, order, n1, Pi, Phi, flags=flags, trace=True)
        re = e.sweep(it, s, order, n1, Pi, Phi, flags=flags, trace=True)
        if re["err"] == 1 and allow_requeue:       # the chain does not fit the settled-chain kernel at some step: the general kernel's job
            requeued.append((it, re["why"]))
            s = ro["s"]
            continue
        assert re["err"] == 0, f"iteration {it}: kernel stopped with err {re['err']} (reason {re['why']}; oracle classes/step {ro['stats']['sum_classes'] / (K * (n - n1 + 1)):.2f})"
        bad = np.where(~np.isclose(re["trace"], ro["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
        assert bad.size == 0, f"iteration {it}: first diverging swept observation {bad[0]}: emu={re['trace'][bad[0]]} oracle={ro['trace'][bad[0]]}"
        assert (re["s"] == ro["s"]).all() and re["p_star"] == ro["p_star"]
        assert np.allclose(re["logweight"], ro["logweight"], rtol=1e-12, atol=1e-12)
        for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
            assert re["stats"][key] == ro["stats"][key], key
        up, mv = o.work()
        ev, cols, splits = expected_work_counters(rec, ro["trace"], N, item_cap=None)
        wk = re["work"]
        assert (wk[:, 1] == up).all() and (wk[:, 3] == mv).all() and wk[:, 2].sum() == ro["stats"]["n_clones"]
        assert (wk[:, 0] == ev).all() and (wk[:, 5] == cols).all() and (wk[:, 6] == splits).all(), (wk, ev, cols, splits)
        eo = o.export()
        for key in ("particle", "max_id"):
            assert (re["state"][key] == eo[key]).all(), key
        for k in range(K):
            m = int(eo["max_id"][k])
            assert (re["state"]["counts"][k][:m] == eo["counts"][k][:m]).all() and (re["state"]["cluster_n"][k][:m] == eo["cluster_n"][k][:m]).all()
        s = ro["s"]
    e.close()
    assert len(requeued) <= allow_requeue, requeued
    return rec


def _chain(O, data, N, P, seed, burn, iters, q1=0, cols_l=64, idcap=128, allow_requeue=0, flags=None, variant="", kinds=None):
    """A real Gibbs chain (the oracle's hyper-parameter updates): `burn` iterations on the oracle alone from the random start of
    src/pmdi.jl:63-66, then `iters` iterations swept by both."""
    from _emu import EmuSweeper
    n, K = data[0].shape[0], len(data)
    n1 = max(1, n // 4)
    hy = O.Hypers(n, N, K, seed=seed)
    kinds = kinds or ["gaussian"] * K
    o = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1)
    e = EmuSweeper(data, N, P, seed=seed, q1_mode=q1, cols_l=cols_l, idcap=idcap, variant=variant, kinds=kinds)
    rec = o.debug_steps(n - n1 + 1)
    requeued, compared = [], 0
    for it in range(1, burn + iters + 1):
        Pi = hy.step(it)
        s, order = np.array(hy.s), np.array(hy.order)
        ro = o.sweep(it, s, order, n1, Pi, hy.Phi, flags=flags, trace=True)
        if it > burn:
            re = e.sweep(it, s, order, n1, Pi, hy.Phi, flags=flags, trace=True)
            if re["err"] == 1:
                requeued.append((it, re["why"]))
            else:
                assert re["err"] == 0
                bad = np.where(~np.isclose(re["trace"], ro["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
                assert bad.size == 0, f"iteration {it}: first diverging swept observation {bad[0]}: emu={re['trace'][bad[0]]} oracle={ro['trace'][bad[0]]}"
                assert (re["s"] == ro["s"]).all() and re["p_star"] == ro["p_star"]
                assert np.allclose(re["logweight"], ro["logweight"], rtol=1e-12, atol=1e-12)
                for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
                    assert re["stats"][key] == ro["stats"][key], key
                up, mv = o.work()
                ev, cols, splits = expected_work_counters(rec, ro["trace"], N, item_cap=None)
                wk = re["work"]
                assert (wk[:, 1] == up).all() and (wk[:, 3] == mv).all() and wk[:, 2].sum() == ro["stats"]["n_clones"]
                assert (wk[:, 0] == ev).all() and (wk[:, 5] == cols).all() and (wk[:, 6] == splits).all(), (wk, ev, cols, splits)
                eo = o.export()
                assert (re["state"]["particle"] == eo["particle"]).all() and (re["state"]["max_id"] == eo["max_id"]).all()
                t5_invariants({"particle": re["state"]["particle"], "counts": re["state"]["counts"], "cluster_n": re["state"]["cluster_n"]}, N, P, K, n)
                compared += 1
        hy.s[:] = ro["s"]
        hy.align_labels(it)
    e.close(); o.close(); hy.close()
    assert len(requeued) <= allow_requeue, requeued
    return compared


@pytest.mark.parametrize("K,P,sep", [(1, 256, 3.0), (2, 256, 3.0), (4, 256, 3.0)])
def test_settled_chain_equals_oracle(O, K, P, sep):
    rng = np.random.default_rng(3)
    data, z = _gauss(rng, 160, K, sep=sep)
    _compare(O, data, 6, P, 3, 100 + K, 40, settle=True, truth=z, allow_requeue=1)


@pytest.mark.parametrize("K,P,n,N", [(3, 512, 240, 8), (2, 1024, 200, 6), (4, 256, 300, 10)])
def test_gibbs_chain_after_burn_in_equals_oracle(O, K, P, n, N):
    rng = np.random.default_rng(5)
    data, _ = _gauss(rng, n, K, D=10, sep=3.0)
    assert _chain(O, data, N, P, 7 + K, 6, 3, allow_requeue=1) >= 2


MIXED = [("gaussian", "categorical"), ("categorical", "negbinom"), ("gaussian", "gaussian", "categorical", "negbinom"), ("negbinom",), ("categorical",)]


@pytest.mark.parametrize("kinds", MIXED, ids=["+".join(k[:3] for k in ks) for ks in MIXED])
def test_mixed_cluster_types_equal_oracle(O, kinds):
    """Categorical and NegBinom datasets beside Gaussian ones (categorical_cluster.jl:29-51, negbinom_cluster.jl:22-51): integer
    statistics, host-built log / loggamma tables -- allocations, log-weights, counters, exported state equal to the oracle's, from a
    planted start and in a Gibbs chain after burn-in (clones, resampling, renumbering moves of the integer pool rows)."""
    rng = np.random.default_rng(21)
    data, z = _mixed(rng, 180, list(kinds))
    _compare(O, data, 6, 256, 3, 300 + len(kinds), 45, settle=True, truth=z, allow_requeue=1, kinds=list(kinds), scramble=0.15)
    assert _chain(O, data, 8, 256, 17 + len(kinds), 5, 3, allow_requeue=1, kinds=list(kinds)) >= 2


def test_mixed_cluster_types_with_feature_flags_and_arena_tables(O):
    rng = np.random.default_rng(22)
    kinds = ["categorical", "gaussian", "negbinom"]
    data, z = _mixed(rng, 160, kinds)
    fl = [(rng.random(d.shape[1]) < 0.7).astype(np.uint8) for d in data]
    _compare(O, data, 6, 512, 2, 77, 40, settle=True, truth=z, flags=fl, allow_requeue=1, kinds=kinds, cols_l=2, idcap=8, scramble=0.2)


def test_up_to_32_particle_classes_per_dataset(O):
    """The class slots of a lane are 5 bits wide; how many classes the LDS tables hold is a layout parameter (16 .. 32, the host's
    choice for the LDS budget).  A planted start whose steps fan one class out into 17 .. 32: handed back with 16, swept in place
    with 32 -- same results as the oracle, whose class count per step is checked to be in that range."""
    rng = np.random.default_rng(24)
    data, z = _mixed(rng, 150, ["gaussian", "categorical"])
    with pytest.raises(AssertionError, match="reason 4"):
        _compare(O, data, 40, 512, 1, 94, 40, settle=0.5, truth=z, allow_requeue=0, kinds=["gaussian", "categorical"], scramble=0.1, cls=16)
    rec = _compare(O, data, 40, 512, 1, 94, 40, settle=0.5, truth=z, allow_requeue=0, kinds=["gaussian", "categorical"], scramble=0.1, cls=32)
    assert 16 < rec[:, :, 0].max() <= 32, rec[:, :, 0].max()
    # ... and, in the 8-wave build (P = 2 048), with the CDF rows of the class slots beyond the first 16 (or 4) in the chain's arena
    # instead of LDS: how cfg4's N = 50 tables hold 32 classes
    for cdfl in (16, 4):
        rec = _compare(O, data, 40, 2048, 1, 96, 40, settle=1.0, truth=z, allow_requeue=0, kinds=["gaussian", "categorical"], scramble=0.1, cls=32, cdfl=cdfl)
        assert rec[:, :, 0].max() > 16, rec[:, :, 0].max()


def test_eight_wave_workgroup_for_2048_particles(O):
    """P = 2 048: a 512-thread workgroup (eight waves, four particles per lane; the first K waves own a dataset each, all eight share
    the particle phases, the statistics phase and the resampling) -- mixed cluster types, N = 50 labels as in BASELINE config 4."""
    rng = np.random.default_rng(23)
    kinds = ["gaussian", "gaussian", "categorical", "negbinom"]
    data, z = _mixed(rng, 120, kinds)
    _compare(O, data, 50, 2048, 2, 91, 30, settle=200.0, truth=z, allow_requeue=0, kinds=kinds, scramble=0.1)
    data, z = _gauss(rng, 120, 1, D=7)
    _compare(O, data, 5, 2048, 2, 92, 30, settle=True, truth=z, allow_requeue=0)


def test_arena_fallback_of_the_lds_tables(O):
    """cols_l = 2 columns and idcap = 8 ids in LDS: nearly every column and cluster id of the chain lives in the arena (the
    direct-indexed tables continue there), same results."""
    rng = np.random.default_rng(6)
    data, z = _gauss(rng, 160, 2, sep=3.0)
    _compare(O, data, 6, 256, 2, 31, 40, settle=True, truth=z, cols_l=2, idcap=8, allow_requeue=0)


def test_uncached_clusters_beyond_the_lds_list(O):
    """A build with two LDS entries for uncached reachable clusters (PM2_XCAP=2) and eight cacheable ids: the clusters a class leader
    reaches beyond that are listed and evaluated through the arena (four per round), same results, nothing is handed back for it."""
    rng = np.random.default_rng(12)
    data, z = _gauss(rng, 200, 2, sep=2.0)
    _compare(O, data, 10, 256, 2, 61, 50, settle=True, truth=z, cols_l=64, idcap=8, allow_requeue=0, variant="xcap2", scramble=0.3)
    assert _chain(O, data, 10, 256, 13, 3, 2, idcap=8, allow_requeue=0, variant="xcap2") >= 2


def test_feature_flags(O):
    # (q1_mode = 1, the per-step new_id reset, multiplies the particle classes: such handles never get this kernel)
    rng = np.random.default_rng(8)
    data, z = _gauss(rng, 160, 2, D=9, sep=3.5)
    fl = [(rng.random(d.shape[1]) < 0.7).astype(np.uint8) for d in data]
    _compare(O, data, 6, 256, 3, 41, 40, settle=True, truth=z, flags=fl, allow_requeue=1)


def test_lane_order_inside_a_wave_does_not_matter(O, monkeypatch):
    """The emulator runs the lanes of a wave in a random order between collectives: every cross-lane LDS dependency of the kernel
    has to be separated by a wave (or workgroup) barrier for the results to stay equal."""
    monkeypatch.setenv("WAVESIM_SHUFFLE", "12345")
    rng = np.random.default_rng(9)
    data, z = _gauss(rng, 160, 2, sep=3.0)
    _compare(O, data, 6, 256, 2, 51, 40, settle=True, truth=z, allow_requeue=0)
    assert _chain(O, data, 6, 256, 11, 4, 2, allow_requeue=1) >= 1
