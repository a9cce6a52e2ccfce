"""The oracle against the reference's own analytic tests (test/runtests.jl T1, T2) and
independent scipy known answers for everything the reference leaves unpinned."""
import numpy as np
import pytest
from scipy import integrate, stats
from scipy.special import gammaln


def test_T1_gaussian_closed_form(O):
    # test/runtests.jl:11-36
    rng = np.random.default_rng(10)
    n = 1000
    x = rng.normal(size=(n, 1))
    c = O.Cluster(x, "gaussian")
    assert c.stats()["n"] == 0
    for i in range(n):
        c.add(i)
    st = c.stats()
    assert st["n"] == n
    assert np.isclose(st["Sigma"][0], x.sum())
    assert np.isclose(st["mu"][0], st["Sigma"][0] / (n + 0.001))
    xbar = st["Sigma"][0] / n
    s2 = ((x - xbar) ** 2).sum()
    beta = 0.5 + 0.5 * (s2 + (0.001 * n * xbar ** 2) / (n + 0.001))
    assert np.isclose(st["beta"][0], beta)
    assert np.isclose(st["lambda"][0], ((0.5 + n * 0.5) * (n + 0.001)) / (st["beta"][0] * (n + 1.001)))
    xc = (x[-1, 0] - st["mu"][0]) * np.sqrt(st["lambda"][0])
    true_lp = stats.t.logpdf(xc, n + 1) + 0.5 * np.log(st["lambda"][0])
    assert np.isclose(true_lp, c.logprob(n - 1), rtol=1.5e-8)


def test_T2_categorical(O):
    # test/runtests.jl:38-54
    rng = np.random.default_rng(11)
    x = rng.integers(1, 11, size=(1000, 1))
    x[0, 0] = 10
    c = O.Cluster(x, "categorical")
    assert c.stats()["n"] == 0
    for i in range(1000):
        c.add(i)
    st = c.stats()
    assert st["n"] == 1000
    for lvl in np.unique(x):
        assert (x == lvl).sum() == st["counts"][lvl - 1, 0]
    row1 = int(np.where(x[:, 0] == 1)[0][0])
    assert np.isclose(c.logprob(row1), np.log(((x == 1).sum() + 0.5) / 1005))


def test_gaussian_predictive_integrates_to_one(O):
    rng = np.random.default_rng(12)
    base = rng.normal(1.0, 2.0, size=(30, 1))
    grid = np.linspace(-60, 60, 24001)
    x = np.vstack([base, grid[:, None]])
    c = O.Cluster(x, "gaussian")
    for i in range(30):
        c.add(i)
    lp = np.array([c.logprob(30 + j) for j in range(0, grid.size, 8)])
    area = integrate.trapezoid(np.exp(lp), grid[::8])
    assert abs(area - 1.0) < 1e-6


def test_categorical_predictive_sums_to_one(O):
    rng = np.random.default_rng(13)
    L = 5
    x = rng.integers(1, L + 1, size=(50, 1))
    x = np.vstack([x, np.arange(1, L + 1)[:, None]])
    c = O.Cluster(x, "categorical")
    for i in range(50):
        c.add(i)
    tot = sum(np.exp(c.logprob(50 + l)) for l in range(L))
    assert abs(tot - 1.0) < 1e-12


def test_negbinom_predictive_sums_to_one_and_chain_rule(O):
    rng = np.random.default_rng(14)
    obs = rng.geometric(0.3, size=(40, 1)) - 1
    grid = np.arange(0, 4000)[:, None]
    x = np.vstack([obs, grid])
    c = O.Cluster(x, "negbinom")
    tot_pred = 0.0
    for i in range(40):
        tot_pred += c.logprob(i)
        c.add(i)
    # sum over the support of the posterior predictive
    tot = sum(np.exp(c.logprob(40 + j)) for j in range(grid.size))
    assert abs(tot - 1.0) < 1e-3       # heavy geometric-mixture tail beyond the grid
    # chain rule: product of one-step predictives == marginal likelihood (negbinom_cluster.jl:53-60)
    assert np.isclose(tot_pred, c.logmarginal()[0], rtol=1e-12, atol=1e-10)
    # closed form of the per-feature term (r = 1, Beta(1,1)): B(n+2, S+x+1)/B(n+1, S+1)
    S, n = int(obs.sum()), 40
    xq = 3
    want = (gammaln(n + 2) + gammaln(1 + xq + S) + gammaln(n + 2 + S) - gammaln(n + 3 + xq + S)
            - gammaln(n + 1) - gammaln(1 + S))
    row = 40 + xq
    assert np.isclose(c.logprob(row), want, rtol=1e-13)


def test_gaussian_marginal_matches_predictive_chain_up_to_constant(O):
    rng = np.random.default_rng(15)
    x = rng.normal(size=(30, 3))
    diffs = []
    for n in (1, 2, 7, 30):
        c = O.Cluster(x, "gaussian")
        tot = 0.0
        for i in range(n):
            tot += c.logprob(i)
            c.add(i)
        diffs.append(tot - c.logmarginal().sum())
    assert np.allclose(diffs, diffs[0], rtol=0, atol=1e-9)


def test_categorical_marginal_formula(O):
    # categorical_cluster.jl:53-66 as written (Q9: not the textbook Dirichlet-multinomial constant)
    rng = np.random.default_rng(16)
    x = rng.integers(1, 5, size=(25, 2))
    x[0, :] = 4
    c = O.Cluster(x, "categorical")
    for i in range(25):
        c.add(i)
    cnt = c.stats()["counts"]
    lm = c.logmarginal()
    for q in range(2):
        nl = 0.5 * x[:, q].max()
        want = gammaln(2 * nl) - gammaln(2 * nl + 25) + sum(gammaln(cnt[r, q] + 0.5) for r in range(int(2 * nl)))
        assert np.isclose(lm[q], want, rtol=1e-13)


def test_feature_flags_skip_features(O):
    rng = np.random.default_rng(17)
    x = rng.normal(size=(20, 4))
    flag = np.array([1, 0, 1, 0], dtype=np.uint8)
    c = O.Cluster(x, "gaussian")
    cs = O.Cluster(np.ascontiguousarray(x[:, [0, 2]]), "gaussian")
    for i in range(19):
        c.add(i, flag)
        cs.add(i)
    st = c.stats()
    assert st["n"] == 19 and (st["Sigma"][[1, 3]] == 0).all() and (st["beta"][[1, 3]] == 0.5).all()
    assert np.isclose(c.logprob(19, flag), cs.logprob(19), rtol=1e-13)
