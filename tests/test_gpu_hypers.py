"""GPU parity of the device-resident Gibbs driver (csrc/pmdi_hypers.hip, pmdi_gibbs_* in include/pmdi_hip.h)
against the literal CPU restatement of src/update_hypers.jl / align_labels! / shuffle! (oracle/pmdi_oracle_hypers.c,
N^K tables kept) through the C ABI.

Bar: every discrete outcome (shuffled order, accepted label swaps, allocations, the alpha* draw of update_Phi!,
Metropolis accept of update_M!) equal; floating-point hyper-parameters within rtol 1e-9 (the factorised evaluation
of the normalising-constant sums reorders additions, device log/exp/cos/lgamma may differ from glibc in the last ulp).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _data(rng, n, K):
    z = rng.integers(0, 3, n)
    return [rng.normal(size=(n, 3)) + 2.0 * (z[:, None] - 1) for _ in range(K)], ["gaussian"] * K


def _mk(pkg, O, rng, n, N, K, P=32, seed=5, chains=1):
    data, kinds = _data(rng, n, K)
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=chains, seed=seed)
    g = pkg.Gibbs(sw, rho=0.25)
    return data, kinds, sw, g


def _copy_state(g, hy, chain=0):
    """oracle state -> device chain"""
    g.set(chain, M=hy.M, gamma=hy.gamma, gamma0=hy.gamma0, Phi=hy.Phi, v=hy.v, Z=hy.Z, s=np.array(hy.s), order=np.array(hy.order))


def _assert_state(st, hy, what="", s=True):
    assert np.allclose(st["M"], hy.M, rtol=RTOL, atol=0), f"M differs {what}: {st['M']} vs {hy.M}"
    assert np.allclose(st["gamma"], hy.gamma, rtol=RTOL, atol=1e-300), f"gamma differs {what}"
    assert np.allclose(st["Phi"], hy.Phi, rtol=RTOL, atol=0), f"Phi differs {what}: {st['Phi']} vs {hy.Phi}"
    assert np.isclose(st["Z"], hy.Z, rtol=RTOL), f"Z differs {what}: {st['Z']} vs {hy.Z}"
    assert np.isclose(st["v"], hy.v, rtol=RTOL), f"v differs {what}: {st['v']} vs {hy.v}"
    if s:
        assert (st["s"] == np.array(hy.s)).all(), f"allocations differ {what}"
    assert (st["order"] == np.array(hy.order)).all(), f"order_obs differs {what}"


@pytest.mark.parametrize("n,N,K", [(300, 5, 1), (400, 6, 2), (500, 4, 3), (256, 5, 4), (1000, 20, 2), (200, 3, 6), (700, 7, 3)])
def test_gibbs_init_equals_oracle(pkg, O, n, N, K):
    """src/pmdi.jl:59-66,95-96: M, gamma, Phi, s, Z, v of every chain; chain c is keyed seed + c."""
    rng = np.random.default_rng(1)
    _, _, sw, g = _mk(pkg, O, rng, n, N, K, seed=21, chains=3)
    for c in range(3):
        hy = O.Hypers(n, N, K, seed=21 + c)
        st = g.get(c)
        _assert_state(st, hy, f"(chain {c})")
        assert np.allclose(st["gamma0"], hy.gamma0, rtol=1e-12)
        hy.close()
    g.close(); sw.close()


@pytest.mark.parametrize("n,N,K", [(300, 5, 1), (400, 6, 2), (500, 4, 3), (600, 20, 4), (300, 8, 5), (200, 3, 6), (150, 2, 8),
                                    (5000, 30, 2), (2000, 50, 3), (1500, 100, 2), (900, 128, 1)])
def test_hypers_kernel_equals_oracle(pkg, O, n, N, K):
    """src/pmdi.jl:172-185 from a mid-chain state (gamma != gamma0, uneven allocations) for several iterations."""
    rng = np.random.default_rng(100 * K + N)
    _, _, sw, g = _mk(pkg, O, rng, n, N, K, seed=7)
    hy = O.Hypers(n, N, K, seed=7)
    # a mid-chain state: allocations concentrated on a few labels and partly agreeing across datasets
    base = rng.choice(N, size=n, p=rng.dirichlet(np.full(N, 0.3))) + 1
    s = np.stack([np.where(rng.random(n) < 0.6, base, rng.integers(1, N + 1, n)) for _ in range(K)], axis=1)
    hy.s[:] = s
    hy.gamma = rng.gamma(0.5, 1.0, size=(N, K)) + 1e-3
    hy.M = rng.gamma(2.0, 1.0, size=K) + 0.1
    if K > 1:
        hy.Phi = rng.gamma(1.0, 1.0, size=hy.npairs)
    hy.update_Z()
    hy.update_v(0)
    _copy_state(g, hy)
    for it in range(1, 4):
        Pi = hy.step(it)
        g.step(pkg.STEP_BEGIN); g.step(pkg.STEP_HYPERS)
        st = g.get(0)
        _assert_state(st, hy, f"(iteration {it})")
        # the next iteration starts from the oracle's values so that rounding differences cannot accumulate into a decision
        _copy_state(g, hy)
    g.close(); sw.close(); hy.close()


@pytest.mark.parametrize("n,N,K,phi", [(400, 6, 2, 3.0), (500, 5, 3, 1.0), (600, 10, 4, 10.0), (300, 4, 5, 0.3), (2000, 30, 3, 2.0),
                                        (1000, 50, 4, 5.0), (3000, 100, 2, 4.0), (2000, 128, 3, 6.0), (3000, 255, 2, 4.0)])
def test_align_kernel_equals_oracle(pkg, O, n, N, K, phi):
    """align_labels! (src/misc.jl:61-96): contingency-table form on the device == recount form of the oracle."""
    rng = np.random.default_rng(7 * K + N)
    _, _, sw, g = _mk(pkg, O, rng, n, N, K, seed=3)
    hy = O.Hypers(n, N, K, seed=3)
    base = rng.integers(1, N + 1, n)
    cols = [base]
    for k in range(1, K):
        perm = rng.permutation(N) + 1
        cols.append(np.where(rng.random(n) < 0.8, perm[base - 1], rng.integers(1, N + 1, n)))
    hy.s[:] = np.stack(cols, axis=1)
    hy.Phi = np.full(hy.npairs, phi) * rng.uniform(0.5, 1.5, hy.npairs)
    swaps = 0
    for it in range(1, 4):
        _copy_state(g, hy)
        g.step(pkg.STEP_BEGIN)
        before = np.array(hy.s)
        hy.align_labels(it)
        swaps += int((before != np.array(hy.s)).any())
        g.step(pkg.STEP_ALIGN)
        st = g.get(0)
        assert (st["s"] == np.array(hy.s)).all(), f"aligned labels differ at call {it}"
        assert (st["gamma"] == hy.gamma).all(), "gamma rows must be exchanged with the labels"
        hy.s[:] = np.stack([rng.permutation(N)[np.array(hy.s)[:, k] - 1] + 1 if k else np.array(hy.s)[:, 0] for k in range(K)], axis=1)
    assert swaps > 0, "the test never exercised an accepted swap"
    g.close(); sw.close(); hy.close()


def test_T4_align_on_device(pkg, O):
    """test/runtests.jl:111-134 on the device path: perfectly permuted datasets with Phi = 10 align, gammas follow."""
    rng = np.random.default_rng(11)
    K, N, n = 5, 10, 10000
    _, _, sw, g = _mk(pkg, O, rng, n, N, K, seed=9)
    s = rng.integers(1, N + 1, size=(n, K))
    gam = rng.gamma(1.0 / N, 1.0, size=(N, K))
    for k in range(1, K):
        shuf = rng.permutation(N) + 1
        s[:, k] = shuf[s[:, 0] - 1]
        gam[:, k] = gam[np.argsort(shuf), 0]
    g.set(0, gamma=gam, Phi=np.full(10, 10.0), s=s)
    for _ in range(10):
        g.step(pkg.STEP_BEGIN); g.step(pkg.STEP_ALIGN)
        st = g.get(0)
        assert (st["s"][:, 1:] == st["s"][:, :1]).all() == (st["gamma"][:, 1:] == st["gamma"][:, :1]).all()
    assert (st["s"][:, 1:] == st["s"][:, :1]).all()
    assert (st["gamma"][:, 1:] == st["gamma"][:, :1]).all()
    g.close(); sw.close()


@pytest.mark.parametrize("K,N,n,P,kinds_mixed", [(1, 6, 240, 64, False), (2, 5, 300, 128, True), (3, 6, 300, 64, True), (4, 5, 200, 256, False)])
def test_gibbs_iterations_equal_oracle_chain(pkg, O, K, N, n, P, kinds_mixed):
    """Whole iterations of src/pmdi.jl:164-384 on the device (pmdi_gibbs_iterate) against the oracle chain
    (hypers restatement -> oracle sweep -> align restatement), two chains, several iterations: allocations equal
    after every iteration, hyper-parameters within tolerance."""
    from conftest import make_mixed
    rng = np.random.default_rng(40 + K)
    if kinds_mixed:
        data, kinds = make_mixed(rng, n)
        data, kinds = data[:K], kinds[:K]
    else:
        data, kinds = _data(rng, n, K)
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=2, seed=77)
    g = pkg.Gibbs(sw, rho=0.25)
    n1 = int(np.floor(0.25 * n))
    hys = [O.Hypers(n, N, K, seed=77 + c) for c in range(2)]
    orcs = [O.Oracle(data, kinds, N, P, seed=77 + c) for c in range(2)]
    for it in range(1, 5):
        g.iterate(1)
        res = g.results()
        for c in range(2):
            hy, orc = hys[c], orcs[c]
            Pi = hy.step(it)
            r = orc.sweep(it, np.array(hy.s), np.array(hy.order), n1, Pi, hy.Phi)
            hy.s[:] = r["s"]
            hy.align_labels(it)
            st = g.get(c)
            _assert_state(st, hy, f"(chain {c}, iteration {it})")
            assert int(res["p_star"][c]) == r["p_star"]
            assert res["stats"][c, 0] == r["stats"]["n_operations"] and res["stats"][c, 1] == r["stats"]["n_resamples"]
            assert np.allclose(res["logweight"][c], r["logweight"], rtol=1e-9, atol=1e-9)
    assert g.iterations == 4
    for o in orcs:
        o.close()
    for h in hys:
        h.close()
    g.close(); sw.close()


def test_gibbs_samples_and_feature_selection(pkg, O):
    """pmdi_gibbs_iterate with retained samples (uint8 [iter][chain][K][n]) and feature selection on."""
    import torch
    rng = np.random.default_rng(3)
    n, N, K, P = 200, 5, 2, 64
    data, kinds = _data(rng, n, K)
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=2, seed=5)
    g = pkg.Gibbs(sw, rho=0.25, feature_select=True)
    fl0 = g.get(1)["flags"]
    want = np.array([O.uniform(5 + 1, 0, 0, k, q, 16) < 0.5 for k in range(K) for q in range(3)], dtype=np.uint8)
    assert (fl0 == want).all()                                   # featureFlag = rand(Bool, D) (src/pmdi.jl:106)
    smp = torch.zeros((3, 2, K, n), dtype=torch.uint8, device="cuda")
    g.iterate(3, samples_ptr=smp.data_ptr())
    torch.cuda.synchronize()
    g.results()
    for c in range(2):
        assert (smp[2, c].cpu().numpy().T + 1 == g.get(c)["s"]).all()
    # the oracle chain with feature selection
    hy, orc = O.Hypers(n, N, K, seed=5), O.Oracle(data, kinds, N, P, seed=5)
    flags = [np.array([O.uniform(5, 0, 0, k, q, 16) < 0.5 for q in range(3)], dtype=np.uint8) for k in range(K)]
    for it in range(1, 4):
        Pi = hy.step(it)
        r = orc.sweep(it, np.array(hy.s), np.array(hy.order), 50, Pi, hy.Phi, flags)
        flags, _ = orc.feature_select(it, r["s"])
        hy.s[:] = r["s"]
        hy.align_labels(it)
        assert (smp[it - 1, 0].cpu().numpy().T + 1 == np.array(hy.s)).all(), f"iteration {it}"
    assert (g.get(0)["flags"] == np.concatenate(flags)).all()
    g.close(); sw.close(); hy.close(); orc.close()


def test_iterations_are_enqueued_without_waiting_for_the_device(pkg):
    """pmdi_gibbs_iterate only enqueues: the argument blocks of the sweep launches travel through pinned staging slots, so the
    host is back long before the device has finished (three iterations of a few HL-sized chains take seconds on the device)."""
    import time
    import torch
    from particlemdi_jl_amd import workloads
    w = workloads.make("HL", 0.5)
    sw = pkg.Sweeper(w["data"], w["kinds"], w["N"], w["P"], n_chains=8, seed=5)
    g = pkg.Gibbs(sw, rho=0.25)
    g.iterate(1)                      # first-use costs (module load, attribute calls) are not what is measured
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.iterate(3)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    g.results()
    print(f"enqueue {t_host * 1e3:.1f} ms, device {t_all * 1e3:.1f} ms")
    assert t_all > 0.2 and t_host < 0.25 * t_all, (t_host, t_all)
    g.close(); sw.close()


def test_a_failed_chain_keeps_its_state_and_its_error_sticks(pkg):
    """Pool capacity exceeded in one sweep of the device-resident driver: that chain keeps its allocations (s_out = s_in), the
    following iterations run on defined data, and pmdi_gibbs_results still reports the first error afterwards."""
    rng = np.random.default_rng(4)
    x = rng.normal(size=(200, 4))                     # no structure: particles diverge, the pool grows
    N, P = 10, 256
    sw = pkg.Sweeper([x], ["gaussian"], N, P, n_chains=2, pool_cap=N + 4)
    g = pkg.Gibbs(sw, rho=0.25)
    s0 = g.get(0)["s"]
    g.iterate(3)
    with pytest.raises(pkg.PmdiError) as e:
        g.results()
    assert e.value.code == -4                         # PMDI_E_POOL, although later sweeps of the chain may have succeeded
    st = g.get(0)
    assert st["s"].min() >= 1 and st["s"].max() <= N
    g.set(0, s=s0); g.set(1, s=s0)                    # a chain given a new state starts without its sticky error
    g.results()
    # the handle outlives what was made from it: pmdi_destroy refuses (PMDI_E_STATE) while the pmdi_gibbs is alive
    assert pkg.lib().pmdi_destroy(sw.h) == -6
    g.close()
    sw.close()
