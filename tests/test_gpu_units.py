"""Device plugin protocol (calc_logprob / cluster_add! / calc_logmarginal through the C ABI)
against the oracle.  Tolerances: integer statistics bit-exact; Gaussian floating point within
1e-6 relative (north_star) -- in practice ~1e-15, asserted at 1e-12."""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu

RTOL = 1e-12       # << the 1e-6 relative the north_star allows for Gaussian log-predictives


@pytest.fixture(scope="module")
def mixed(pkg):
    from conftest import make_mixed
    rng = np.random.default_rng(21)
    data, kinds = make_mixed(rng, 240)
    sw = pkg.Sweeper(data, kinds, 8, 32, n_chains=1, seed=1)
    yield data, kinds, sw
    sw.close()


def test_T1_gaussian_on_device(pkg):
    rng = np.random.default_rng(10)
    n = 1000
    x = rng.normal(size=(n, 1))
    sw = pkg.Sweeper([x], ["gaussian"], 4, 8)
    cb = sw.clusters(0, 1)
    for i in range(n):
        cb.add([i + 1])
    st = cb.stats()[0]
    nn, mu, Sig, lam, beta = st[0], st[1], st[2], st[3], st[4]
    assert nn == n and np.isclose(Sig, x.sum()) and np.isclose(mu, Sig / (n + 0.001))
    xbar = Sig / n
    s2 = ((x - xbar) ** 2).sum()
    assert np.isclose(beta, 0.5 + 0.5 * (s2 + (0.001 * n * xbar ** 2) / (n + 0.001)))
    assert np.isclose(lam, ((0.5 + n * 0.5) * (n + 0.001)) / (beta * (n + 1.001)))
    xc = (x[-1, 0] - mu) * np.sqrt(lam)
    assert np.isclose(stats.t.logpdf(xc, n + 1) + 0.5 * np.log(lam), cb.logprob([n])[0], rtol=1.5e-8)


def test_T2_categorical_on_device(pkg):
    rng = np.random.default_rng(11)
    x = rng.integers(1, 11, size=(1000, 1))
    x[0, 0] = 10
    sw = pkg.Sweeper([x], ["categorical"], 4, 8)
    cb = sw.clusters(0, 1)
    for i in range(1000):
        cb.add([i + 1])
    st = cb.stats()[0]
    assert st[0] == 1000
    for lvl in np.unique(x):
        assert (x == lvl).sum() == st[1 + lvl - 1]
    row1 = int(np.where(x[:, 0] == 1)[0][0]) + 1
    assert np.isclose(cb.logprob([row1])[0], np.log(((x == 1).sum() + 0.5) / 1005))


@pytest.mark.parametrize("k", [0, 1, 2])
@pytest.mark.parametrize("use_flags", [False, True])
def test_batch_matches_oracle(O, mixed, k, use_flags):
    data, kinds, sw = mixed
    rng = np.random.default_rng(30 + k)
    B, n, D = 16, data[k].shape[0], data[k].shape[1]
    flag = (rng.random(D) < 0.6).astype(np.uint8) if use_flags else None
    cb = sw.clusters(k, B)
    oc = [O.Cluster(data[k], kinds[k]) for _ in range(B)]
    for step in range(25):
        rows = rng.integers(0, n, size=B)
        cb.add(rows + 1, flag)
        for b in range(B):
            oc[b].add(int(rows[b]), flag)
    probe = rng.integers(0, n, size=B)
    got = cb.logprob(probe + 1, flag)
    want = np.array([oc[b].logprob(int(probe[b]), flag) for b in range(B)])
    lm_got = cb.logmarginal()
    lm_want = np.stack([oc[b].logmarginal() for b in range(B)])
    if kinds[k] == "gaussian":
        assert np.allclose(got, want, rtol=RTOL, atol=0)
        assert np.allclose(lm_got, lm_want, rtol=RTOL, atol=0)
    else:   # integer statistics + host-built tables: bit-exact
        assert (got == want).all()
        assert (lm_got == lm_want).all()
    st = cb.stats()
    for b in range(B):
        o = oc[b].stats()
        assert st[b, 0] == o["n"]
        if kinds[k] == "gaussian":
            assert (st[b, 1:1 + D] == o["mu"]).all() and (st[b, 1 + D:1 + 2 * D] == o["Sigma"]).all()
            assert (st[b, 1 + 2 * D:1 + 3 * D] == o["lambda"]).all() and (st[b, 1 + 3 * D:1 + 4 * D] == o["beta"]).all()
        elif kinds[k] == "categorical":
            L = int(data[k].max())
            assert (st[b, 1:].reshape(D, L).T == o["counts"]).all()
        else:
            assert (st[b, 1:] == o["Sigma"]).all()


def test_empty_cluster_is_prior_predictive(O, mixed):
    data, kinds, sw = mixed
    for k in range(3):
        cb = sw.clusters(k, 4)
        rows = np.array([1, 2, 3, 4])
        got = cb.logprob(rows)
        want = np.array([O.Cluster(data[k], kinds[k]).logprob(int(r - 1)) for r in rows])
        if kinds[k] == "gaussian":
            assert np.allclose(got, want, rtol=RTOL)
        else:
            assert (got == want).all()


@pytest.mark.gpu
def test_label_counts_on_device(pkg):
    # countn(s[:, k], n) for every label (src/update_hypers.jl:72) == numpy bincount
    rng = np.random.default_rng(5)
    n, N, K, Cn = 1237, 9, 2, 3
    data = [rng.normal(size=(n, 2)), rng.normal(size=(n, 3))]
    sw = pkg.Sweeper(data, ["gaussian", "gaussian"], N, 8, n_chains=Cn, seed=1)
    s = rng.integers(1, N + 1, size=(Cn, K, n))
    s[0, 0, :] = 4                                      # one label holds everything
    got = sw.label_counts(s)
    want = np.stack([[np.bincount(s[c, k] - 1, minlength=N) for k in range(K)] for c in range(Cn)])
    assert got.shape == (Cn, K, N) and (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nlab", [0, 12, 40])       # 0: byte-compare kernel; 12 / 40: one-hot int8 MFMA kernel, 1 / 2 label blocks
@pytest.mark.parametrize("S,K,n,lo,hi", [(1, 1, 1, 0, 1), (37, 2, 53, 0, 53), (130, 3, 257, 64, 201), (65, 1, 1000, 500, 1000)])
def test_psm_counts_on_device(pkg, O, S, K, n, lo, hi, nlab):
    # co-clustering counts of generate_psm (consensus_map.jl:50-56), integer-exact
    import torch
    from particlemdi_jl_amd import psm
    rng = np.random.default_rng(S * 7 + n)
    smp = rng.integers(0, nlab if nlab else 12, size=(S, K, n)).astype(np.uint8)
    got = psm.psm_counts_device(torch.from_numpy(smp).cuda(), lo, hi, n_labels=nlab).cpu().numpy()
    assert (got == O.psm_counts(smp, lo, hi)).all()
    # and the posterior-similarity rows built from it equal the host mirror's
    rows_dev = psm.psm_rows(torch.from_numpy(smp).cuda(), lo, hi, n_labels=nlab).cpu().numpy()
    rows_host = psm.psm_rows(smp, lo, hi, host=True)
    assert (rows_dev == rows_host).all()


@pytest.mark.gpu
def test_generate_psm_from_an_output_file_on_the_device(pkg, tmp_path):
    """generate_psm(outputFile, burnin, thin) (consensus_map.jl:31-65) end to end -- native reader, HIP counts,
    division and "Overall" -- equals the host mirror (which tests/test_host.py pins to the reference's loops)."""
    from particlemdi_jl_amd.psm import generate_psm
    rng = np.random.default_rng(12)
    K, n, iters, N = 3, 70, 40, 6
    path = tmp_path / "out.csv"
    w = pkg.CsvWriter(path, K, n, data_names=["a", "b", "c"])
    for _ in range(iters):
        w.row(rng.gamma(2.0, 1.0, K), rng.gamma(1.0, 1.0, 3), -1.0, rng.integers(1, N + 1, size=(n, K)))
    w.close()
    dev, host = generate_psm(str(path), 4, 3), generate_psm(str(path), 4, 3, host=True)
    assert dev.names == host.names == ["a", "b", "c", "Overall"]
    for x, y in zip(dev.psm, host.psm):
        assert np.array_equal(x, y)


def test_allgather_samples_through_the_c_abi(pkg, O):
    """the one collective of the path behind the C ABI (pmdi_comm_*, pmdi_allgather_samples), here in its
    one-rank form on the one GPU of the test box (both ways of forming the communicator), feeding the PSM counts:
    chains -> retained samples -> RCCL all-gather -> pmdi_psm_counts_device == the oracle's counts."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(12)
    n, N, K, P, chains, T = 96, 5, 2, 32, 3, 4
    z = rng.integers(0, 3, n)
    data = [rng.normal(size=(n, 3)) + 2.0 * (z[:, None] - 1) for _ in range(K)]
    sw = pkg.Sweeper(data, ["gaussian"] * K, N, P, n_chains=chains, seed=4)
    g = pkg.Gibbs(sw, rho=0.25)
    smp = torch.zeros((T, chains, K, n), dtype=torch.uint8, device="cuda")
    g.iterate(T, samples_ptr=smp.data_ptr())
    g.results()
    comm = pkg.Comm(0, 1, 0, pkg.Comm.unique_id())
    out = torch.zeros((1,) + tuple(smp.shape), dtype=torch.uint8, device="cuda")
    comm.allgather(smp.data_ptr(), out.data_ptr(), smp.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (out[0] == smp).all()
    comm.close()
    # one process, G devices (G = 1 here): pmdi_comm_init_all + the group form of the call
    L = pkg.lib()
    h = (C.c_void_p * 1)()
    assert L.pmdi_comm_init_all(1, None, h) == 0, L.pmdi_last_error()
    assert L.pmdi_comm_size(h[0]) == 1 and L.pmdi_comm_rank(h[0]) == 0
    out2 = torch.zeros_like(out)
    send = (C.c_void_p * 1)(smp.data_ptr()); recv = (C.c_void_p * 1)(out2.data_ptr())
    assert L.pmdi_allgather_samples(h, 1, send, recv, smp.numel(), None) == 0, L.pmdi_last_error()
    torch.cuda.synchronize()
    assert (out2[0] == smp).all()
    L.pmdi_comm_destroy(h[0])
    # the consumer: pooled samples (S = T * chains, K, n) -> co-clustering counts
    from particlemdi_jl_amd.psm import psm_counts_device
    pooled = out[0].reshape(T * chains, K, n).contiguous()
    got = psm_counts_device(pooled, 0, n, n_labels=N).cpu().numpy()
    want = O.psm_counts(pooled.cpu().numpy(), 0, n)
    assert (got == want).all()
    g.close(); sw.close()
