"""Pins oracle/pmdi_oracle_hypers.c (the literal restatement of src/update_hypers.jl, align_labels!,
shuffle!, the initialisation of src/pmdi.jl:59-96):
  * what the reference's own tests hold for this code: T3 (update_Z == brute-force sum over all N^K label
    combinations, test/runtests.jl:57-108) and T4 (align_labels! on perfectly permuted datasets,
    test/runtests.jl:111-134), restated;
  * known answers of the samplers (moments, closed forms at K = 1);
  * a second, independently written numpy restatement (tests/_np_hypers.py, (N,)*K arrays instead of
    N^K-row tables, contingency tables instead of recounts) driven by the same Philox variates.
No GPU needed."""
import itertools

import numpy as np
import pytest

import _np_hypers as NP

SITE = dict(shuffle=5, m_normal=6, m_accept=7, gamma=8, phi_alpha=9, phi_gamma=10, v=11, align=12)


class PhiloxDraws:
    def __init__(self, O, seed):
        self.O, self.seed = O, seed

    def uniform(self, it, pos, k, site):
        return self.O.uniform(self.seed, it, pos, k, 0, site)

    def normal(self, it, pos, k, site):
        return self.O.normal(self.seed, it, pos, k, 0, site)

    def gamma(self, shape, it, pos, k, site):
        return self.O.gamma(shape, self.seed, it, pos, k, site)


def test_T3_update_Z_equals_brute_force(O):
    # test/runtests.jl:57-108 (the grid is N = 2..20, K = 1..5 there; the same identity on the sizes the table fits)
    rng = np.random.default_rng(0)
    for N, K in [(2, 1), (7, 1), (20, 1), (2, 2), (9, 2), (20, 2), (3, 3), (11, 3), (20, 3), (4, 4), (13, 4), (2, 5), (7, 5), (3, 6)]:
        hy = O.Hypers(50, N, K, seed=1)
        g0 = rng.gamma(1.0 / N, 1.0, size=(N, K)) + 1e-9
        Phi = rng.gamma(1.0, 5.0, size=hy.npairs) if K > 1 else np.zeros(1)
        hy.gamma0, hy.Phi = g0, Phi
        tab = np.ones((N,) * K)
        for k in range(K):
            tab = tab * g0[:, k].reshape([N if d == k else 1 for d in range(K)])
        pr = 0
        for a in range(K - 1):
            for b in range(a + 1, K):
                ia = np.arange(N).reshape([N if d == a else 1 for d in range(K)])
                ib = np.arange(N).reshape([N if d == b else 1 for d in range(K)])
                tab = tab * (1.0 + Phi[pr] * (ia == ib))
                pr += 1
        assert np.isclose(hy.update_Z(), tab.sum(), rtol=1e-10), (N, K)
        hy.close()


def test_T4_align_labels_on_permuted_datasets(O):
    # test/runtests.jl:111-134
    rng = np.random.default_rng(1)
    K, N, n = 5, 10, 10000
    hy = O.Hypers(n, N, K, seed=2)
    s = rng.integers(1, N + 1, size=(n, K))
    gam = rng.gamma(1.0 / N, 1.0, size=(N, K))
    for k in range(1, K):
        shuf = rng.permutation(N) + 1
        s[:, k] = shuf[s[:, 0] - 1]
        gam[:, k] = gam[np.argsort(shuf), 0]
    hy.s[:] = s
    hy.gamma, hy.Phi = gam, np.full(10, 10.0)
    for it in range(1, 11):
        hy.align_labels(it)
        S, G = np.array(hy.s), hy.gamma
        assert (S[:, 1:] == S[:, :1]).all() == (G[:, 1:] == G[:, :1]).all()
    assert (S[:, 1:] == S[:, :1]).all() and (G[:, 1:] == G[:, :1]).all()
    hy.close()


def test_samplers_known_answers(O):
    z = np.array([O.normal(3, 1, i, 0, 0, 6) for i in range(40000)])
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1.0) < 0.03
    for shape in (0.05, 0.5, 1.0, 3.7, 250.0, 10000.0):
        g = np.array([O.gamma(shape, 3, 1, i, 0, 8) for i in range(40000)])
        assert (g > 0).all()
        assert abs(g.mean() / shape - 1.0) < 0.03 + 0.1 * (shape < 0.1), shape
        assert abs(g.var() / shape - 1.0) < 0.08 + 0.3 * (shape < 0.1), shape
    # counter-based: a draw is a pure function of its key
    assert O.gamma(2.5, 9, 4, 3, 2, 8) == O.gamma(2.5, 9, 4, 3, 2, 8) != O.gamma(2.5, 9, 4, 3, 2, 10)


def test_shuffle_is_a_uniform_permutation(O):
    n = 6
    hy = O.Hypers(n, 2, 1, seed=4)
    seen = {}
    for it in range(1, 7201):
        hy.order[:] = np.arange(1, n + 1)
        hy.shuffle(it)
        p = tuple(np.array(hy.order))
        assert sorted(p) == list(range(1, n + 1))
        seen[p] = seen.get(p, 0) + 1
    assert len(seen) == 720 and min(seen.values()) >= 1
    first = np.bincount([p[0] for p in seen for _ in range(seen[p])], minlength=n + 1)[1:]
    assert (abs(first / 7200.0 - 1.0 / n) < 0.02).all()
    # cumulative like shuffle!(order_obs): the second call permutes the first call's result
    hy.order[:] = np.arange(1, n + 1)
    hy.shuffle(1); a = np.array(hy.order); hy.shuffle(2); b = np.array(hy.order)
    hy.order[:] = a
    hy.shuffle(2)
    assert (np.array(hy.order) == b).all()
    hy.close()


def test_K1_closed_forms(O):
    """K = 1: norm_temp is the initial gamma itself, so update_gamma!'s beta* is 1 + v*gamma0[n]/gamma[n] and Z = sum(gamma0)."""
    n, N = 400, 6
    hy = O.Hypers(n, N, 1, seed=8)
    g0 = hy.gamma0[:, 0].copy()
    assert np.isclose(hy.Z, g0.sum(), rtol=1e-13)
    rng = np.random.default_rng(3)
    gam = rng.gamma(1.0, 1.0, size=(N, 1)) + 0.01
    hy.gamma = gam
    M, v = hy.M[0], hy.v
    cnt = np.bincount(np.array(hy.s)[:, 0], minlength=N + 1)[1:]
    hy.update_gamma(5)
    for m in range(N):
        want = O.gamma(M / N + cnt[m], 8, 5, m, 0, SITE["gamma"]) * (1.0 / (1.0 + v * g0[m] / gam[m, 0])) + np.finfo(float).eps
        assert np.isclose(hy.gamma[m, 0], want, rtol=1e-13)
    hy.close()


def _sync(np_h, hy):
    np_h.M, np_h.gamma, np_h.Phi = hy.M.copy(), hy.gamma.copy(), hy.Phi.copy()
    np_h.v, np_h.Z = hy.v, hy.Z
    np_h.s = np.array(hy.s).copy()
    np_h._sumGamma = np_h._outer_sum(np.log(hy.gamma0))


@pytest.mark.parametrize("n,N,K", [(300, 5, 1), (400, 6, 2), (500, 4, 3), (350, 7, 4), (200, 3, 5)])
def test_C_restatement_equals_numpy_restatement(O, n, N, K):
    """update_M!, update_gamma!, update_Phi!, update_Z, update_v and align_labels! (both the recount form and the
    contingency-table form of the numpy side) from a common mid-chain state with common variates."""
    seed = 17
    rng = np.random.default_rng(5 * K + N)
    hy = O.Hypers(n, N, K, seed=seed)
    base = rng.integers(1, N + 1, n)
    hy.s[:] = np.stack([np.where(rng.random(n) < 0.7, base, rng.integers(1, N + 1, n)) for _ in range(K)], axis=1)
    hy.gamma = rng.gamma(0.5, 1.0, size=(N, K)) + 1e-3
    if K > 1:
        hy.Phi = rng.gamma(1.0, 1.0, size=hy.npairs)
    hy.update_Z(); hy.update_v(0)
    ref = NP.HyperState(n, N, K, np.random.default_rng(0), draws=PhiloxDraws(O, seed))
    for it in range(1, 4):
        _sync(ref, hy)
        ref.it = it
        hy.update_M(it); ref.update_M()
        assert np.allclose(ref.M, hy.M, rtol=1e-12)
        ref.M = hy.M.copy()
        hy.update_gamma(it); ref.update_gamma()
        assert np.allclose(ref.gamma, hy.gamma, rtol=1e-10)
        ref.gamma = hy.gamma.copy()
        hy.update_Phi(it); ref.update_Phi()
        assert np.allclose(ref.Phi, hy.Phi, rtol=1e-10)
        ref.Phi = hy.Phi.copy()
        assert np.isclose(ref.update_Z(), hy.update_Z(), rtol=1e-11)
        hy.update_v(it); ref.update_v()
        assert np.isclose(ref.v, hy.v, rtol=1e-11)
        # align: both numpy forms against the C recount
        ref2 = NP.HyperState(n, N, K, np.random.default_rng(0), draws=PhiloxDraws(O, seed))
        _sync(ref, hy); _sync(ref2, hy)
        ref2.it = ref.it = it
        hy.align_labels(it); ref.align_labels(); ref2._align_labels_by_recount()
        assert (ref.s == np.array(hy.s)).all() and (ref2.s == np.array(hy.s)).all()
        assert (ref.gamma == hy.gamma).all() and (ref2.gamma == hy.gamma).all()
        hy.s[:] = np.stack([rng.permutation(N)[np.array(hy.s)[:, k] - 1] + 1 for k in range(K)], axis=1)
    hy.close()


def test_step_runs_in_pmdi_order_and_keeps_invariants(O):
    hy = O.Hypers(500, 6, 3, seed=12)
    for it in range(1, 6):
        Pi = hy.step(it)
        assert Pi.shape == (6, 3) and np.allclose(Pi.sum(0), 1.0) and (Pi > 0).all()
        assert (hy.Phi >= 0).all() and (hy.M > 0).all() and hy.Z > 0 and hy.v > 0
        hy.align_labels(it)
    assert sorted(np.array(hy.order)) == list(range(1, 501))
    s = np.array(hy.s)
    assert s.min() >= 1 and s.max() <= 6
    hy.close()
