"""Parity of the HIP sweep (through the C ABI) with the oracle: golden vectors, seeded random
cases, the reference's T5 invariants on the exported device state, edge cases, batched chains,
and the device-pointer entry point."""
import ctypes as C

import numpy as np
import pytest

from _cases import check_work_counters, expected_work_counters, golden_cases, load_golden, replay, t5_invariants
from conftest import make_mixed, random_hypers

pytestmark = pytest.mark.gpu


class GpuRunner:
    def __init__(self, pkg, data, kinds, N, P, seed, q1, **kw):
        self.sw = pkg.Sweeper(data, kinds, N, P, n_chains=1, seed=seed, q1_mode=q1, **kw)

    def sweep(self, it, s, order, n1, Pi, Phi, flags, trace=False):
        r = self.sw.sweep(it, s[None], order[None], n1, Pi[None], np.atleast_1d(Phi)[None],
                          flags=None if flags is None else flags[None], trace=trace)
        out = {"s": r["s"][0], "p_star": int(r["p_star"][0]), "logweight": r["logweight"][0], "stats": r["stats"][0]}
        if trace:
            out["trace"] = r["trace"][0]
        return out

    def feature_select(self, it, s):
        f, p = self.sw.feature_select(it, s[None])
        return f[0], p[0]


@pytest.mark.parametrize("case", golden_cases())
def test_golden(pkg, case):
    z, data, kinds = load_golden(case)
    replay(z, data, kinds, lambda d, k, N, P, seed, q1: GpuRunner(pkg, d, k, N, P, seed, q1))


def _compare_run(pkg, O, data, kinds, N, P, iters, seed, n1, q1=0, flags=None, block=0, check_state=True, q2=0):
    rng = np.random.default_rng(seed)
    n, K = data[0].shape[0], len(data)
    g = GpuRunner(pkg, data, kinds, N, P, seed, q1, block_threads=block, q2_mode=q2)
    o = O.Oracle(data, kinds, N, P, seed=seed, q1_mode=q1, q2_mode=q2)
    Dcum = np.cumsum([d.shape[1] for d in data])[:-1]
    s = rng.integers(1, N + 1, size=(n, K))
    rec = o.debug_steps(n - n1 + 1)
    for it in range(1, iters + 1):
        order = rng.permutation(n) + 1
        Pi, Phi = random_hypers(rng, N, K)
        rg = g.sweep(it, s, order, n1, Pi, Phi, flags, trace=True)
        ro = o.sweep(it, s, order, n1, Pi, Phi, flags=None if flags is None else np.split(flags, Dcum), trace=True)
        bad = np.where(~np.isclose(rg["trace"], ro["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
        assert bad.size == 0, f"first diverging swept observation: {bad[0]} gpu={rg['trace'][bad[0]]} cpu={ro['trace'][bad[0]]}"
        assert (rg["s"] == ro["s"]).all()
        assert rg["p_star"] == ro["p_star"]
        assert np.allclose(rg["logweight"], ro["logweight"], rtol=1e-9, atol=1e-8)
        for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
            assert rg["stats"][key] == ro["stats"][key], key
        # the work counters behind bench.py's algorithmic byte count: clusters updated / cloned / moved per dataset
        wk, (up, mv) = g.sw.work_counters()[0], o.work()
        assert (wk[:, 1] == up).all() and (wk[:, 3] == mv).all() and wk[:, 2].sum() == ro["stats"]["n_clones"]
        # ... and, pinned to the oracle's per-step record: clusters evaluated (the ones a class leader reads at src/pmdi.jl:232),
        # distinct columns of particle[:, :, k] met by the resampling events, columns made by copy-on-write splits
        check_work_counters(wk, rec, ro["trace"], N, int(g.sw.swept_by()[0]))
        s = ro["s"]
    if check_state:
        eg, eo = g.sw.export_state(0), o.export()
        for key in ("particle", "counts", "cluster_n", "max_id"):
            assert (eg[key] == eo[key]).all(), key
        t5_invariants(eg, N, P, K, n)
    return g


@pytest.mark.parametrize("P,iters,block", [(2, 2, 0), (64, 3, 0), (1024, 3, 0), (1024, 2, 128), (1024, 2, 256), (1024, 2, 512), (1024, 2, 1024), (300, 2, 0)])
def test_T5_problem(pkg, O, P, iters, block):
    # the reference's integration problem: test/runtests.jl:136-162
    rng = np.random.default_rng(0)
    data = [np.vstack([rng.normal(2, 1, (50, 16)), rng.normal(-2, 1, (50, 16))]) for _ in range(3)]
    _compare_run(pkg, O, data, ["gaussian"] * 3, 10, P, iters, 40 + P, 25, block=block)


@pytest.mark.parametrize("q1", [0, 1])
def test_mixed_types(pkg, O, q1):
    rng = np.random.default_rng(1)
    data, kinds = make_mixed(rng, 300)
    _compare_run(pkg, O, data, kinds, 12, 256, 3, 50 + q1, 75, q1=q1)


@pytest.mark.parametrize("P,n,N", [(64, 300, 12), (1024, 200, 8), (300, 150, 6)])
def test_q2_mode_history_permuted_like___pmdi(pkg, O, P, n, N):
    """q2_mode = 1: the allocation history follows the ancestors at every resampling (src/__pmdi.jl:285).  The device
    logs the ancestor tables and traces the selected particle's lineage back at the end; the oracle permutes literally."""
    rng = np.random.default_rng(4)
    data, kinds = make_mixed(rng, n)
    g = _compare_run(pkg, O, data, kinds, N, P, 3, 90 + P, n // 4, q2=1)
    # ... and it is a different trajectory from pmdi()'s (q2_mode = 0) on the same inputs
    rng = np.random.default_rng(90 + P)
    s = rng.integers(1, N + 1, size=(n, 3))
    order = rng.permutation(n) + 1
    Pi, Phi = random_hypers(rng, N, 3)
    a = g.sweep(1, s, order, n // 4, Pi, Phi, None)
    b = GpuRunner(pkg, data, kinds, N, P, 90 + P, 0).sweep(1, s, order, n // 4, Pi, Phi, None)
    assert a["stats"]["n_resamples"] == b["stats"]["n_resamples"] > 0 and a["p_star"] == b["p_star"]
    assert (a["s"] != b["s"]).any()


def test_both_forms_of_the_sweep_for_K_gt_1(pkg, O, monkeypatch):
    """K > 1 runs either as one workgroup per chain or as K cooperating workgroups (one per dataset, hand-off per swept
    observation); small batches default to the split form, so the single-workgroup form is forced here -- and the split
    form with more chains than fit at once (workgroups of a chain queue behind each other's chains)."""
    rng = np.random.default_rng(21)
    data, kinds = make_mixed(rng, 240)
    monkeypatch.setenv("PMDI_KSPLIT", "0")
    g = _compare_run(pkg, O, data, kinds, 9, 256, 3, 33, 60)
    assert not g.sw.split
    monkeypatch.setenv("PMDI_KSPLIT", "1")
    g = _compare_run(pkg, O, data, kinds, 9, 256, 3, 33, 60)
    assert g.sw.split
    # 700 chains x 3 datasets = 2 100 workgroups: several times the resident capacity
    Cn, N, P, n = 700, 6, 64, 240
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=Cn, seed=1234)
    assert sw.split
    s = rng.integers(1, N + 1, size=(Cn, n, 3))
    order = np.stack([rng.permutation(n) + 1 for _ in range(Cn)])
    hyp = [random_hypers(rng, N, 3) for _ in range(Cn)]
    r = sw.sweep(1, s, order, 60, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]))
    for c in (0, 1, 7, 8, 349, 698, 699):
        o = O.Oracle(data, kinds, N, P, seed=1234 + c).sweep(1, s[c], order[c], 60, hyp[c][0], hyp[c][1])
        assert (r["s"][c] == o["s"]).all() and int(r["p_star"][c]) == o["p_star"]
        assert r["stats"][c]["n_operations"] == o["stats"]["n_operations"] and r["stats"][c]["n_clones"] == o["stats"]["n_clones"]


def test_feature_flags(pkg, O):
    rng = np.random.default_rng(2)
    data, kinds = make_mixed(rng, 200)
    fl = (rng.random(8 + 6 + 5) < 0.6).astype(np.uint8)
    _compare_run(pkg, O, data, kinds, 12, 200, 2, 60, 50, flags=fl)
    fl0 = np.zeros(8 + 6 + 5, dtype=np.uint8)       # every feature off: only the prior Pi drives the draws
    _compare_run(pkg, O, data, kinds, 6, 64, 1, 61, 50, flags=fl0)


@pytest.mark.parametrize("N,P,n1,n", [(2, 2, 1, 40), (2, 33, 40, 40), (10, 17, 2, 60), (64, 128, 16, 130), (5, 4096, 10, 50),
                                      (65, 64, 20, 200), (100, 96, 30, 260), (128, 256, 10, 300)])
def test_edges(pkg, O, N, P, n1, n):
    # n1 = 1: no known prefix; n1 = n: a single swept observation; N = 2 / N = 64: one class per <= 64-lane group;
    # N = 65 / 100 / 128: a class's CDF spans two 64-lane chunks of a wave (the reference allows any N <= n, src/pmdi.jl:54);
    # P not a multiple of the wave size; P = 4096: four particles per lane
    rng = np.random.default_rng(N * 1000 + P)
    z = rng.integers(0, 2, n)
    g = rng.normal(size=(n, 3)) + 3.0 * z[:, None]
    c = 1 + z[:, None] * np.ones((1, 2), dtype=np.int64) + rng.integers(0, 2, (n, 2))
    _compare_run(pkg, O, [g, c], ["gaussian", "categorical"], N, P, 2, 70 + N, n1)


@pytest.mark.parametrize("N,P,block", [(129, 64, 0), (150, 256, 0), (200, 128, 0), (255, 256, 0), (180, 512, 512)])
def test_more_than_128_labels(pkg, O, N, P, block):
    """N up to 255 (the reference: any N <= n; labels travel as bytes here): beyond 128 labels the mutation CDF's cumsum
    (src/pmdi.jl:240) is Base's pairwise one -- c[1] = e1, the other N - 1 elements split once into two leaves --, formed by one
    wave with four labels per lane.  Mixed data types, two sweeps, everything equal to the oracle."""
    rng = np.random.default_rng(12)
    data, kinds = make_mixed(rng, 320)
    _compare_run(pkg, O, data, kinds, N, P, 2, 40 + N, 60, block=block)


def test_all_three_step_paths_are_exercised(pkg, O, monkeypatch):
    """The general kernel's fast (LDS tables), converted (LDS census overflow) and fallback (burn-in) steps all occur
    somewhere in this run, and the result is still the oracle's.  (PMDI_SETTLED=0: from the second sweep on the chain would
    otherwise go to the settled-chain kernel, which has no such paths to count.)"""
    monkeypatch.setenv("PMDI_SETTLED", "0")
    rng = np.random.default_rng(11)
    n = 400
    x = rng.normal(size=(n, 3))                    # no structure: particles diverge quickly
    N, P = 16, 1024
    g = GpuRunner(pkg, [x], ["gaussian"], N, P, 5, 0)
    o = O.Oracle([x], ["gaussian"], N, P, seed=5)
    s = rng.integers(1, N + 1, size=(n, 1))
    tot = {"steps_fast": 0, "steps_converted": 0, "steps_fallback": 0}
    for it in range(1, 4):
        order = rng.permutation(n) + 1
        Pi, Phi = random_hypers(rng, N, 1)
        rg = g.sweep(it, s, order, 100, Pi, Phi, None)
        ro = o.sweep(it, s, order, 100, Pi, Phi)
        assert (rg["s"] == ro["s"]).all() and rg["p_star"] == ro["p_star"]
        assert rg["stats"]["n_clones"] == ro["stats"]["n_clones"]
        for key in tot:
            tot[key] += rg["stats"][key]
        s = ro["s"]
    assert all(v > 0 for v in tot.values()), tot


def test_chains_are_independent_and_seeded(pkg, O):
    rng = np.random.default_rng(3)
    data, kinds = make_mixed(rng, 150)
    N, P, K, n, Cn = 8, 128, 3, 150, 5
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=Cn, seed=900)
    s = rng.integers(1, N + 1, size=(Cn, n, K))
    order = np.stack([rng.permutation(n) + 1 for _ in range(Cn)])
    hyp = [random_hypers(rng, N, K) for _ in range(Cn)]
    r = sw.sweep(1, s, order, 37, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]))
    for c in range(Cn):          # chain c == a single-chain oracle run with seed + c
        o = O.Oracle(data, kinds, N, P, seed=900 + c).sweep(1, s[c], order[c], 37, hyp[c][0], hyp[c][1])
        assert (r["s"][c] == o["s"]).all() and int(r["p_star"][c]) == o["p_star"]
        assert r["stats"][c]["n_operations"] == o["stats"]["n_operations"]


def test_three_launch_groups_give_the_same_chains(pkg, O, monkeypatch):
    """Automatic width (P > 256): a sweep is split into a heaviest / heavy / light launch by what the
    chains' previous sweep looked like.  With thresholds that spread these chains over all three
    launches, every chain must still equal its single-chain oracle run, iteration after iteration."""
    thr = 400
    monkeypatch.setenv("PMDI_LIGHT_IDS", str(thr))   # live clusters per step above which a chain is heavy
    monkeypatch.setenv("PMDI_VERY_HEAVY", "2")       # the first two heavy chains of the launch order
    rng = np.random.default_rng(21)
    n, N, P, Cn, n1 = 260, 8, 512, 6, 60
    z = rng.integers(0, 3, n)
    x = rng.normal(size=(n, 6)) + 3.0 * (z[:, None] - 1)
    sw = pkg.Sweeper([x], ["gaussian"], N, P, n_chains=Cn, seed=700)
    assert sw.block_threads == 512
    orc = [O.Oracle([x], ["gaussian"], N, P, seed=700 + c) for c in range(Cn)]
    s = rng.integers(1, N + 1, size=(Cn, n, 1))
    mixed = 0
    for it in range(1, 6):
        order = np.stack([rng.permutation(n) + 1 for _ in range(Cn)])
        hyp = [random_hypers(rng, N, 1) for _ in range(Cn)]
        r = sw.sweep(it, s, order, n1, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]))
        for c in range(Cn):
            o = orc[c].sweep(it, s[c], order[c], n1, hyp[c][0], hyp[c][1])
            assert (r["s"][c] == o["s"]).all() and int(r["p_star"][c]) == o["p_star"], (it, c)
            for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
                assert r["stats"][c][key] == o["stats"][key], (it, c, key)
        s = r["s"].copy()
        ops = [st["n_operations"] / (n - n1 + 1) for st in r["stats"]]
        if it < 5 and min(ops) <= thr < max(ops):    # the NEXT sweep runs light and heavy chains side by side
            mixed += 1
    assert mixed >= 1


def test_pool_overflow_is_reported(pkg):
    rng = np.random.default_rng(4)
    x = rng.normal(size=(200, 4))                     # no structure: particles diverge, the pool grows
    N, P = 10, 256
    sw = pkg.Sweeper([x], ["gaussian"], N, P, pool_cap=N + 4)
    Pi, Phi = random_hypers(rng, N, 1)
    with pytest.raises(pkg.PmdiError) as e:
        sw.sweep(1, rng.integers(1, N + 1, size=(1, 200, 1)), (rng.permutation(200) + 1)[None], 50, Pi[None], Phi[None])
    assert e.value.code == -4                         # PMDI_E_POOL


def test_bad_arguments(pkg):
    rng = np.random.default_rng(5)
    x = rng.normal(size=(50, 2))
    sw = pkg.Sweeper([x], ["gaussian"], 4, 8)
    Pi, Phi = random_hypers(rng, 4, 1)
    s = rng.integers(1, 5, size=(1, 50, 1))
    order = (rng.permutation(50) + 1)[None]
    with pytest.raises(pkg.PmdiError) as e:
        sw.sweep(1, s, order, 0, Pi[None], Phi[None])            # rho*n < 1
    assert e.value.code == -1
    with pytest.raises(pkg.PmdiError) as e:
        sw.sweep(1, s * 0 + 5, order, 10, Pi[None], Phi[None])   # label outside 1..N
    assert e.value.code == -5
    with pytest.raises(pkg.PmdiError):
        pkg.Sweeper([np.zeros((10, 2), dtype=np.int64)], ["categorical"], 3, 4)   # level < 1


def test_device_pointer_entry_matches_host_entry(pkg, O):
    import torch
    from particlemdi_jl_amd._lib import _check, lib
    rng = np.random.default_rng(6)
    data, kinds = make_mixed(rng, 120)
    N, P, K, n, n1 = 7, 64, 3, 120, 30
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=1, seed=77)
    s = rng.integers(1, N + 1, size=(n, K))
    order = rng.permutation(n) + 1
    Pi, Phi = random_hypers(rng, N, K)
    want = O.Oracle(data, kinds, N, P, seed=77).sweep(1, s, order, n1, Pi, Phi)
    dev = torch.device("cuda", 0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    s_d = t((s.T - 1)[None], torch.int32); o_d = t((order - 1)[None], torch.int32)
    Pi_d = t(Pi.T[None], torch.float64); lp_d = t(np.log(1.0 + Phi)[None], torch.float64)
    so = torch.empty_like(s_d); lw = torch.empty((1, P), dtype=torch.float64, device=dev)
    ps = torch.empty(1, dtype=torch.int32, device=dev); st = torch.zeros((1, 8), dtype=torch.int64, device=dev)
    er = torch.ones(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    _check(lib().pmdi_sweep_device(sw.h, 1, C.c_void_p(s_d.data_ptr()), C.c_void_p(o_d.data_ptr()), n1,
                                   C.c_void_p(Pi_d.data_ptr()), C.c_void_p(lp_d.data_ptr()), None, 0.0,
                                   C.c_void_p(so.data_ptr()), C.c_void_p(lw.data_ptr()), C.c_void_p(ps.data_ptr()),
                                   C.c_void_p(st.data_ptr()), C.c_void_p(er.data_ptr()), C.c_void_p(stream.cuda_stream)))
    torch.cuda.synchronize()
    assert int(er.item()) == 0
    assert (so.cpu().numpy()[0].T + 1 == want["s"]).all()
    assert int(ps.item()) + 1 == want["p_star"]
    assert int(st[0, 0].item()) == want["stats"]["n_operations"]


def test_split_form_stops_all_partners_when_one_runs_out_of_pool(pkg, monkeypatch):
    """K cooperating workgroups per chain (PMDI_KSPLIT=1): when one dataset's pool overflows, its workgroup poisons the chain's
    arrival counter and the partners stop at their next hand-off -- PMDI_E_POOL comes back at once, nobody sits out the
    20-second watchdog of the hand-off."""
    import time
    monkeypatch.setenv("PMDI_KSPLIT", "1")
    rng = np.random.default_rng(4)
    n, N, P = 200, 10, 256
    z = rng.integers(0, 3, n)
    data = [rng.normal(size=(n, 4)), rng.normal(size=(n, 4)) + 6.0 * (z[:, None] - 1), rng.normal(size=(n, 3)) + 6.0 * (z[:, None] - 1)]
    sw = pkg.Sweeper(data, ["gaussian"] * 3, N, P, n_chains=3, pool_cap=N + 6)     # dataset 0 has no structure: its pool grows
    assert sw.split
    s = np.repeat(np.repeat((z + 1)[None, :, None], 3, axis=2), 3, axis=0)
    s[:, :, 0] = rng.integers(1, N + 1, size=(3, n))
    order = np.stack([rng.permutation(n) + 1 for _ in range(3)])
    hyp = [random_hypers(rng, N, 3) for _ in range(3)]
    t0 = time.perf_counter()
    with pytest.raises(pkg.PmdiError) as e:
        sw.sweep(1, s, order, 50, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]))
    assert e.value.code == -4                         # PMDI_E_POOL
    assert time.perf_counter() - t0 < 10.0            # ... not the watchdog
    sw.close()


def _gauss_planted(rng, n, K, D=12, sep=3.0):
    z = rng.integers(0, 3, n)
    return [rng.normal(size=(n, D + k)) + sep * (z[:, None] - 1) for k in range(K)], z


@pytest.mark.parametrize("K,P,n,N", [(1, 256, 300, 6), (2, 512, 300, 8), (4, 1024, 400, 10), (3, 256, 240, 20)])
def test_settled_chain_kernel_equals_oracle(pkg, O, monkeypatch, K, P, n, N):
    """The settled-chain kernel (csrc/pmdi_sweep2.hip) on chains that look settled -- the planted clustering with a few labels
    scrambled --, forced from the first sweep (PMDI_SETTLED=2; a chain whose step outgrows its tables is handed back to the general
    kernel inside the same sweep: same results either way): trace, allocations, picked particle, log-weights, counters, work
    counters and the exported state equal the oracle's."""
    monkeypatch.setenv("PMDI_SETTLED", "2")
    monkeypatch.setenv("PMDI_KSPLIT", "0")          # (a small batch with K > 1 would default to K workgroups per chain)
    rng = np.random.default_rng(100 + K)
    data, z = _gauss_planted(rng, n, K)
    kinds = ["gaussian"] * K
    n1 = n // 4
    C = 3
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=C, seed=900)
    assert sw.settled
    orcs = [O.Oracle(data, kinds, N, P, seed=900 + c) for c in range(C)]
    recs = [o.debug_steps(n - n1 + 1) for o in orcs]
    s = np.repeat(np.repeat((z + 1)[None, :, None], K, axis=2), C, axis=0)
    idx = rng.random(s.shape) < 0.04
    s[idx] = rng.integers(1, N + 1, size=int(idx.sum()))
    for it in range(1, 4):
        order = np.stack([rng.permutation(n) + 1 for _ in range(C)])
        hyp = [random_hypers(rng, N, K) for _ in range(C)]
        for h in hyp:
            h[0][:3] += 1.0; h[0][:] = h[0] / h[0].sum(0)
        rg = sw.sweep(it, s, order, n1, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]), trace=True)
        wk = sw.work_counters()
        kern = sw.swept_by()
        for c in range(C):
            ro = orcs[c].sweep(it, s[c], order[c], n1, hyp[c][0], hyp[c][1], trace=True)
            bad = np.where(~np.isclose(rg["trace"][c], ro["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
            assert bad.size == 0, f"chain {c} iteration {it}: first diverging swept observation {bad[0]}: gpu={rg['trace'][c][bad[0]]} cpu={ro['trace'][bad[0]]}"
            assert (rg["s"][c] == ro["s"]).all() and int(rg["p_star"][c]) == ro["p_star"]
            assert np.allclose(rg["logweight"][c], ro["logweight"], rtol=1e-9, atol=1e-8)
            for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
                assert rg["stats"][c][key] == ro["stats"][key], key
            up, mv = orcs[c].work()
            assert (wk[c][:, 1] == up).all() and (wk[c][:, 3] == mv).all()
            check_work_counters(wk[c], recs[c], ro["trace"], N, int(kern[c]))
            eg, eo = sw.export_state(c), orcs[c].export()
            assert (eg["particle"] == eo["particle"]).all() and (eg["max_id"] == eo["max_id"]).all()
            t5_invariants(eg, N, P, K, n)
            s[c] = ro["s"]
    gb = sw.given_back()
    print(f"K={K} P={P}: chains handed back to the general kernel (reachable, chosen, classes, total) = {gb.tolist()} of {3 * C} chain-sweeps")
    assert gb[3] <= 2 * C          # most of these sweeps fit the settled-chain kernel (the hand-back path is exercised, not the norm)
    sw.close()


def _mixed_planted(rng, n, kinds, sep=3.0):
    """Datasets of the given cluster types sharing one planted 3-cluster structure (as tests/test_emu_sweep2.py)."""
    z = rng.integers(0, 3, n)
    data = []
    for j, kind in enumerate(kinds):
        if kind == "gaussian":
            data.append(rng.normal(size=(n, 9 + j)) + sep * (z[:, None] - 1))
        elif kind == "categorical":
            pr = rng.dirichlet(0.5 * np.ones(4), size=(3, 7 + j))
            x = np.empty((n, 7 + j), dtype=np.int64)
            for q in range(7 + j):
                for c in range(3):
                    m = z == c
                    x[m, q] = 1 + rng.choice(4, size=int(m.sum()), p=pr[c, q])
            data.append(x)
        else:
            data.append(rng.geometric(0.15 + 0.3 * z[:, None], size=(n, 6 + j)) - 1)
    return data, z


@pytest.mark.parametrize("kinds,P,n,N,settle", [
    (("gaussian", "categorical"), 1024, 400, 30, 30.0),                       # BASELINE config 3's shape
    (("categorical", "negbinom"), 512, 300, 8, 1.0),
    (("gaussian", "gaussian", "categorical", "negbinom"), 2048, 300, 50, 200.0),   # BASELINE config 4's shape: eight-wave workgroups
    (("negbinom",), 256, 300, 6, 1.0),
    (("gaussian", "gaussian"), 2048, 240, 12, 5.0),
], ids=["cfg3-shape", "cat+nb", "cfg4-shape", "nb-alone", "P2048"])
def test_settled_chain_kernel_mixed_types_equal_oracle(pkg, O, monkeypatch, kinds, P, n, N, settle):
    """The settled-chain kernel on Categorical / NegBinom datasets beside Gaussian ones (categorical_cluster.jl:29-51,
    negbinom_cluster.jl:22-51) and on 2 048 particles (512-thread workgroups): forced from the first sweep on planted chains; trace,
    allocations, picked particle, log-weights, counters, work counters and exported state equal the oracle's; integer types bit-exact."""
    monkeypatch.setenv("PMDI_SETTLED", "2")
    monkeypatch.setenv("PMDI_KSPLIT", "0")
    kinds = list(kinds)
    K = len(kinds)
    rng = np.random.default_rng(200 + K + P)
    data, z = _mixed_planted(rng, n, kinds)
    n1 = n // 4
    C = 3
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=C, seed=901)
    assert sw.settled
    orcs = [O.Oracle(data, kinds, N, P, seed=901 + c) for c in range(C)]
    recs = [o.debug_steps(n - n1 + 1) for o in orcs]
    s = np.repeat(np.repeat((z + 1)[None, :, None], K, axis=2), C, axis=0)
    idx = rng.random(s.shape) < 0.04
    s[idx] = rng.integers(1, N + 1, size=int(idx.sum()))
    all_int = all(k != "gaussian" for k in kinds)
    for it in range(1, 4):
        order = np.stack([rng.permutation(n) + 1 for _ in range(C)])
        hyp = [random_hypers(rng, N, K) for _ in range(C)]
        for h in hyp:
            h[0][:3] += settle; h[0][:] = h[0] / h[0].sum(0)
        rg = sw.sweep(it, s, order, n1, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]), trace=True)
        wk = sw.work_counters()
        kern = sw.swept_by()
        for c in range(C):
            ro = orcs[c].sweep(it, s[c], order[c], n1, hyp[c][0], hyp[c][1], trace=True)
            bad = np.where(~np.isclose(rg["trace"][c], ro["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
            assert bad.size == 0, f"chain {c} iteration {it}: first diverging swept observation {bad[0]}: gpu={rg['trace'][c][bad[0]]} cpu={ro['trace'][bad[0]]}"
            assert (rg["s"][c] == ro["s"]).all() and int(rg["p_star"][c]) == ro["p_star"]
            # (integer types: the log-predictives are the oracle's bits -- host-built tables, same order of additions; the increment's
            # log(f[N]) is the device's log either way)
            assert np.allclose(rg["logweight"][c], ro["logweight"], rtol=1e-12 if all_int else 1e-9, atol=1e-9 if all_int else 1e-8)
            for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
                assert rg["stats"][c][key] == ro["stats"][key], key
            up, mv = orcs[c].work()
            assert (wk[c][:, 1] == up).all() and (wk[c][:, 3] == mv).all()
            check_work_counters(wk[c], recs[c], ro["trace"], N, int(kern[c]))
            eg, eo = sw.export_state(c), orcs[c].export()
            assert (eg["particle"] == eo["particle"]).all() and (eg["max_id"] == eo["max_id"]).all()
            t5_invariants(eg, N, P, K, n)
            s[c] = ro["s"]
    gb = sw.given_back()
    print(f"{'+'.join(kinds)} P={P}: chains handed back (reachable, chosen, classes, total) = {gb.tolist()} of {3 * C} chain-sweeps")
    assert gb[3] <= 2 * C
    sw.close()


@pytest.mark.parametrize("kinds,P,N,flags", [
    (("gaussian", "gaussian", "gaussian", "gaussian"), 1024, 40, False),          # 4-bit class slots (the headline shape): 16 classes
    (("gaussian", "categorical"), 512, 40, True),                                   # 5-bit slots, tables squeezed to 16 classes
    (("categorical", "negbinom", "gaussian"), 256, 30, False),
    (("gaussian", "gaussian", "categorical", "negbinom"), 2048, 50, False),       # 512-thread workgroups
    (("gaussian",), 512, 40, False),
], ids=["K4-P1024", "gau+cat-flags", "cat+nb+gau", "cfg4-shape", "K1"])
def test_hand_over_mid_sweep_equals_oracle(pkg, O, kinds, P, N, flags):
    """The settled-chain kernel hands a chain over to the general kernel's code IN PLACE at the observation whose step does not fit its
    tables (here: more than 16 particle classes -- the class capacity is squeezed to 16 and the prior leaves mass on many empty
    labels, so hand-overs happen at assorted positions of the sweep, in one or several datasets of the same observation, and again
    in later iterations after the chain has returned to the settled-chain kernel).  Everything the reference defines is compared with
    the oracle: per-observation trace, allocations, picked particle, log-weights, counters, work counters, the exported state."""
    kinds = list(kinds)
    K = len(kinds)
    rng = np.random.default_rng(300 + K + P)
    n = 260
    data, z = _mixed_planted(rng, n, kinds)
    n1 = n // 4
    C = 4
    fl = None
    if flags:
        fl = np.concatenate([(rng.random(d.shape[1]) < 0.7).astype(np.uint8) for d in data])
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=C, seed=903, tuning={"settled": 2, "sticky": 0, "ksplit": 0, "s2_cls": 16})
    assert sw.settled
    orcs = [O.Oracle(data, kinds, N, P, seed=903 + c) for c in range(C)]
    recs = [o.debug_steps(n - n1 + 1) for o in orcs]
    Dcum = np.cumsum([d.shape[1] for d in data])[:-1]
    s = np.repeat(np.repeat((z + 1)[None, :, None], K, axis=2), C, axis=0)
    idx = rng.random(s.shape) < 0.08
    s[idx] = rng.integers(1, N + 1, size=int(idx.sum()))
    seen = set()
    for it in range(1, 5):
        order = np.stack([rng.permutation(n) + 1 for _ in range(C)])
        hyp = [random_hypers(rng, N, K) for _ in range(C)]
        for h in hyp:
            h[0][:3] += 0.6; h[0][:] = h[0] / h[0].sum(0)
        rg = sw.sweep(it, s, order, n1, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]),
                      None if fl is None else np.stack([fl] * C), trace=True)
        wk = sw.work_counters()
        kern = sw.swept_by()
        for c in range(C):
            ro = orcs[c].sweep(it, s[c], order[c], n1, hyp[c][0], hyp[c][1], flags=None if fl is None else np.split(fl, Dcum), trace=True)
            bad = np.where(~np.isclose(rg["trace"][c], ro["trace"], rtol=1e-9, atol=1e-9).all(axis=1))[0]
            assert bad.size == 0, f"chain {c} iteration {it} (kernel {kern[c]}): first diverging swept observation {bad[0]}: gpu={rg['trace'][c][bad[0]]} cpu={ro['trace'][bad[0]]}"
            assert (rg["s"][c] == ro["s"]).all() and int(rg["p_star"][c]) == ro["p_star"]
            assert np.allclose(rg["logweight"][c], ro["logweight"], rtol=1e-9, atol=1e-8)
            for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
                assert rg["stats"][c][key] == ro["stats"][key], key
            up, mv = orcs[c].work()
            assert (wk[c][:, 1] == up).all() and (wk[c][:, 3] == mv).all()
            check_work_counters(wk[c], recs[c], ro["trace"], N, int(kern[c]))
            eg, eo = sw.export_state(c), orcs[c].export()
            assert (eg["particle"] == eo["particle"]).all() and (eg["max_id"] == eo["max_id"]).all()
            t5_invariants(eg, N, P, K, n)
            seen.add(int(kern[c]))
            s[c] = ro["s"]
    gb = sw.given_back()
    print(f"{'+'.join(kinds)} P={P}: kernels that finished the chain-sweeps {sorted(seen)}; hand-overs {gb.tolist()} of {4 * C} chain-sweeps")
    assert 2 in seen and gb[3] >= 2          # chains were handed over mid-sweep ...
    sw.close()


def test_settled_chain_kernel_hands_back_what_does_not_fit(pkg, O, monkeypatch):
    """From the random start of src/pmdi.jl:63-66 a chain has dozens of particle classes (more than the sixteen the settled-chain
    kernel holds per dataset): forced onto that kernel it is handed back at once and the general kernel sweeps it: results equal the
    oracle's, and the counter of handed-back chains moves."""
    monkeypatch.setenv("PMDI_SETTLED", "2")
    monkeypatch.setenv("PMDI_KSPLIT", "0")
    rng = np.random.default_rng(77)
    data, _ = _gauss_planted(rng, 200, 2, sep=0.5)
    g = _compare_run(pkg, O, data, ["gaussian"] * 2, 20, 1024, 2, 78, 50)
    assert g.sw.settled and g.sw.given_back()[3] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("ticket", ["1", "0"])
def test_settled_launch_positions_drawn_by_ticket(pkg, O, monkeypatch, ticket):
    """The settled-chain launch lets a workgroup DRAW its position in the launch order (SweepArgs::ticket) instead of taking
    blockIdx.x: more chains than workgroup slots, every chain kept on that kernel (PMDI_SETTLED=2: from the random start most of them
    are handed over, so the general kernel's code has to find the drawn position too).  Every chain equals its single-chain oracle run
    over several sweeps (the launch order changes with the chains' costs), with the ticket and without."""
    monkeypatch.setenv("PMDI_SETTLED", "2")
    monkeypatch.setenv("PMDI_KSPLIT", "0")
    monkeypatch.setenv("PMDI_TICKET", ticket)
    monkeypatch.setenv("PMDI_STICKY", "0")          # (a handed-over chain starts its next sweep on the settled-chain kernel again)
    rng = np.random.default_rng(505)
    n, N, P, K, Cn, n1 = 200, 20, 1024, 2, 700, 50
    data, _ = _gauss_planted(rng, n, K, sep=0.5)            # (the shape of test_settled_chain_kernel_hands_back_what_does_not_fit)
    kinds = ["gaussian"] * K
    sw = pkg.Sweeper(data, kinds, N, P, n_chains=Cn, seed=4100)
    assert sw.settled
    check = [0, 1, 17, 255, 256, 511, 512, 513, 698, 699]
    orc = {c: O.Oracle(data, kinds, N, P, seed=4100 + c) for c in check}
    s = rng.integers(1, N + 1, size=(Cn, n, K))
    by_all = set()
    for it in range(1, 4):
        order = np.stack([rng.permutation(n) + 1 for _ in range(Cn)])
        hyp = [random_hypers(rng, N, K) for _ in range(Cn)]
        r = sw.sweep(it, s, order, n1, np.stack([h[0] for h in hyp]), np.stack([h[1] for h in hyp]))
        by = sw.swept_by()
        assert set(by.tolist()) <= {1, 2}
        by_all |= set(by.tolist())
        for c in check:
            o = orc[c].sweep(it, s[c], order[c], n1, hyp[c][0], hyp[c][1])
            assert (r["s"][c] == o["s"]).all() and int(r["p_star"][c]) == o["p_star"], (it, c)
            for key in ("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"):
                assert r["stats"][c][key] == o["stats"][key], (it, c, key)
        # the chains that are not compared one by one: every one swept, none twice (allocations in range, no error reported)
        assert (r["s"] >= 1).all() and (r["s"] <= N).all()
        s = r["s"].copy()
    assert by_all == {1, 2}, by_all
