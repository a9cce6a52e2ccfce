"""Host-side mirror: hyper-parameter updates (T3/T4 of test/runtests.jl), CSV number format,
the pmdi() argument checks, workloads."""
import itertools

import numpy as np
import pytest


def test_T3_update_Z_vs_brute_force(pkg):
    # test/runtests.jl:57-108, smaller grid
    from particlemdi_jl_amd.hypers import HyperState
    rng = np.random.default_rng(0)
    for N, K in [(2, 1), (3, 2), (4, 3), (5, 2), (3, 4), (7, 1)]:
        hy = HyperState(50, N, K, rng)
        gam = np.exp(hy._sumGamma)            # product of gammas per combination (initial gamma)
        if K > 1:
            hy.Phi = rng.gamma(1, 5, size=hy.npairs)
        Z = 0.0
        for combo in itertools.product(range(N), repeat=K):
            tmp = np.prod([hy.gamma[combo[k], k] for k in range(K)])
            if K > 1:
                for i, (a, b) in enumerate(hy.pairs):
                    tmp *= 1 + hy.Phi[i] * (combo[a] == combo[b])
            Z += tmp
        assert np.isclose(hy.update_Z(), Z, rtol=1e-10)
        assert gam.shape == (N,) * K


def test_T4_align_labels(pkg):
    # test/runtests.jl:111-134: perfectly permuted datasets, strong Phi -> labels and gammas align together
    from particlemdi_jl_amd.hypers import HyperState
    rng = np.random.default_rng(1)
    K, N, n = 4, 6, 2000
    hy = HyperState(n, N, K, rng)
    s = rng.integers(1, N + 1, size=(n, K))
    gam = rng.gamma(1.0 / N, 1, size=(N, K))
    for k in range(1, K):
        shuf = rng.permutation(N) + 1
        s[:, k] = shuf[s[:, 0] - 1]
        inv = np.argsort(shuf)
        gam[:, k] = gam[inv, 0]
    hy.s, hy.gamma, hy.Phi = s, gam, np.full(hy.npairs, 10.0)
    for _ in range(10):
        hy.align_labels()
        assert (hy.s[:, 1:] == hy.s[:, :1]).all() == (hy.gamma[:, 1:] == hy.gamma[:, :1]).all()
    assert (hy.s[:, 1:] == hy.s[:, :1]).all()
    assert (hy.gamma[:, 1:] == hy.gamma[:, :1]).all()


def test_hyper_step_keeps_shapes_and_positivity(pkg):
    from particlemdi_jl_amd.hypers import HyperState
    rng = np.random.default_rng(2)
    hy = HyperState(200, 5, 3, rng)
    for _ in range(5):
        Pi = hy.step_pmdi_order()
        assert Pi.shape == (5, 3) and np.allclose(Pi.sum(0), 1.0) and (Pi > 0).all()
        assert (hy.Phi >= 0).all() and (hy.M > 0).all() and hy.Z > 0 and hy.v > 0
    assert hy.s.min() >= 1 and hy.s.max() <= 5


def test_batched_hypers_k1(pkg):
    from particlemdi_jl_amd.batched import BatchedHypersK1
    rng = np.random.default_rng(3)
    bh = BatchedHypersK1(500, 6, 7, rng)
    s = bh.initial_s()
    assert s.shape == (7, 500) and s.min() >= 0 and s.max() <= 5
    counts = np.stack([np.bincount(r, minlength=6) for r in s]).astype(float)
    Pi = bh.step(counts)
    assert Pi.shape == (7, 6) and np.allclose(Pi.sum(1), 1.0) and (Pi > 0).all()


def test_jl_float(pkg):
    from particlemdi_jl_amd.pmdi import jl_float
    cases = {1.0: "1.0", 0.001: "0.001", 1e-5: "1.0e-5", 123456.7: "123456.7", 1234567.8: "1.2345678e6",
             100000.0: "100000.0", 1e6: "1.0e6", 3.0: "3.0", 2.5e-7: "2.5e-7", -0.25: "-0.25", 1e-4: "0.0001",
             0.30000000000000004: "0.30000000000000004", 0.0: "0.0", 12.0: "12.0", 1e21: "1.0e21"}
    for v, want in cases.items():
        assert jl_float(v) == want


def test_pmdi_asserts(pkg, tmp_path):
    from particlemdi_jl_amd.pmdi import pmdi
    x = np.random.default_rng(0).normal(size=(30, 3))
    out = str(tmp_path / "o.csv")
    with pytest.raises(AssertionError):
        pmdi([x], ["GaussianCluster", "GaussianCluster"], 5, 8, 0.25, 1, out)      # src/pmdi.jl:50
    with pytest.raises(AssertionError):
        pmdi([x, x[:20]], ["GaussianCluster"] * 2, 5, 8, 0.25, 1, out)             # :52
    with pytest.raises(AssertionError):
        pmdi([x], ["GaussianCluster"], 5, 8, 1.0, 1, out)                          # :53
    with pytest.raises(AssertionError):
        pmdi([x], ["GaussianCluster"], 1, 8, 0.25, 1, out)                         # :54
    with pytest.raises(AssertionError):
        pmdi([x], ["GaussianCluster"], 5, 1, 0.25, 1, out)                         # :55
    with pytest.raises(TypeError):
        pmdi([x], ["SignCluster"], 5, 8, 0.25, 1, out)                             # user types: no device kernel


def test_preprocessing_helpers(pkg):
    from particlemdi_jl_amd.pmdi import coerce_categorical, gaussian_normalise
    rng = np.random.default_rng(4)
    x = rng.normal(3, 2, size=(200, 2))
    y = gaussian_normalise(x)
    assert np.allclose(np.median(y, axis=0), 0.0, atol=1e-12)
    c = coerce_categorical(np.array([["a", "x"], ["b", "x"], ["a", "y"]]))
    assert c.tolist() == [[1, 1], [2, 1], [1, 2]]


def test_workloads_and_algorithmic_bytes(pkg):
    from particlemdi_jl_amd import workloads
    w = workloads.make("cfg3", 0.02)
    assert w["K"] == 2 and w["data"][1].min() >= 1 and w["data"][1].max() == 4
    w4 = workloads.make("cfg4", 0.02)
    assert w4["data"][3].min() >= 0 and len(w4["data"]) == 4
    # SURVEY.md section 8(d) table
    assert workloads.algorithmic_bytes_per_obs_particle(["gaussian"], [50], 20) == 19392
    assert workloads.algorithmic_bytes_per_obs_particle(["gaussian", "categorical"], [50, 20], 30) == 32864
    assert workloads.algorithmic_bytes_per_obs_particle(["gaussian"] * 3, [200] * 3, 50) == 519696


def test_psm_rows_matches_reference_definition(pkg):
    from particlemdi_jl_amd.psm import psm_rows
    rng = np.random.default_rng(5)
    S, K, n = 7, 2, 12
    samples = rng.integers(1, 4, size=(S, K, n)).astype(np.uint8)
    full = psm_rows(samples, 0, n, host=True)
    assert full.shape == (K + 1, n, n)
    for k in range(K):
        for j in range(n - 1):
            for i in range(j + 1, n):
                assert np.isclose(full[k, i, j], (samples[:, k, i] == samples[:, k, j]).mean())   # consensus_map.jl:52
        assert (np.diag(full[k]) == 1).all() and np.triu(full[k], 1).sum() == 0
    want = np.eye(n) + sum(full[k] for k in range(K)) / K
    np.fill_diagonal(want, 1.0)
    assert np.allclose(full[K], want)
    assert np.allclose(np.concatenate([psm_rows(samples, 0, 5, host=True), psm_rows(samples, 5, n, host=True)], axis=1), full)


def test_align_labels_tables_equal_the_recount(pkg):
    # the contingency-table form makes the same decisions as the line-by-line restatement
    import copy
    from particlemdi_jl_amd.hypers import HyperState
    for seed in range(6):
        rng = np.random.default_rng(seed)
        n, N, K = 180 + 17 * seed, 4 + seed, 2 + seed % 3
        hy = HyperState(n, N, K, np.random.default_rng(100 + seed))
        z = rng.integers(1, N + 1, n)
        for k in range(K):       # correlated allocations with permuted labels: swaps do get accepted
            p = rng.permutation(N) + 1
            hy.s[:, k] = np.where(rng.random(n) < 0.8, p[z - 1], rng.integers(1, N + 1, n))
        hy.Phi[:] = rng.gamma(2.0, 2.0, size=hy.Phi.shape)
        a, b = copy.deepcopy(hy), copy.deepcopy(hy)
        a.align_labels()
        b._align_labels_by_recount()
        assert (a.s == b.s).all() and (a.gamma == b.gamma).all()
        assert a.rng.random() == b.rng.random()          # same number of uniforms consumed
        assert not (a.s == hy.s).all()                    # and something did move
