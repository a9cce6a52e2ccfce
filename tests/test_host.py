"""Host-side pieces that need no GPU: the native CSV writer (byte-level), the pmdi() argument checks,
preprocessing helpers, workloads, the PSM host mirror.  (The hyper-parameter updates are device kernels now:
tests/test_gpu_hypers.py; their oracle is pinned in tests/test_oracle_hypers.py.)"""
import itertools

import numpy as np
import pytest


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
    with pytest.raises(ValueError):
        pmdi([x], ["GaussianCluster", "GaussianCluster"], 5, 8, 0.25, 1, out)      # src/pmdi.jl:50
    with pytest.raises(ValueError):
        pmdi([x, x[:20]], ["GaussianCluster"] * 2, 5, 8, 0.25, 1, out)             # :52
    with pytest.raises(ValueError):
        pmdi([x], ["GaussianCluster"], 5, 8, 1.0, 1, out)                          # :53
    with pytest.raises(ValueError):
        pmdi([x], ["GaussianCluster"], 1, 8, 0.25, 1, out)                         # :54
    with pytest.raises(ValueError):
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


def test_csv_writer_bytes(pkg, tmp_path):
    """header and rows of src/pmdi.jl:147-158,379 byte for byte, from hand-derived expectations:
    writedlm(io, [M; Phi; ll; s[1:n*K]]', ',') prints a Float64 row (the Int allocations are promoted)."""
    from particlemdi_jl_amd import CsvWriter
    p = tmp_path / "o.csv"
    w = CsvWriter(p, 2, 3)
    s = np.array([[1, 4], [2, 5], [3, 6]])                       # n x K; s[1:6] is column-major: 1,2,3,4,5,6
    w.row([2.0, 1.9650000000000003], [0.25], 0.0, s)
    w.row([1e-5, 123456.7], [1234567.8], 12.5, s[::-1])
    w.close()
    assert p.read_bytes() == (
        b"MassParameter_1,MassParameter_2,phi_1_2,ll,K1_n1,K1_n2,K1_n3,K2_n1,K2_n2,K2_n3\n"
        b"2.0,1.9650000000000003,0.25,0.0,1.0,2.0,3.0,4.0,5.0,6.0\n"
        b"1.0e-5,123456.7,1.2345678e6,12.5,3.0,2.0,1.0,6.0,5.0,4.0\n")
    # K = 1 still has one phi column (calculate_Phi_lab(1) = [1 1], src/misc.jl:2); custom data names
    p1 = tmp_path / "o1.csv"
    w = CsvWriter(p1, 1, 2, data_names=["expr"])
    w.row([2.0], [0.0], 0.0, np.array([[7], [10]]))
    w.close()
    assert p1.read_bytes() == b"MassParameter_1,phi_1_1,ll,expr_n1,expr_n2\n2.0,0.0,0.0,7.0,10.0\n"
    # feature-selection file (src/pmdi.jl:111-116): <name>_d<d> header, Bool rows
    pf = tmp_path / "f.csv"
    w = CsvWriter(pf, 2, 3, data_names=["a", "b"], feature_D=[2, 1])
    w.flags([1, 0, 1])
    w.close()
    assert pf.read_bytes() == b"a_d1,a_d2,b_d1\ntrue,false,true\n"


def test_bench_byte_models():
    """bench.py's algorithmic byte count: the column-table model against a hand count, and against the per-particle-table model
    of the earlier builds (same counters, more bytes)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pmdi_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    w = {"N": 4, "K": 1, "kinds": ["gaussian"], "D": [3], "data": [None]}
    C, n, n1, P = 2, 10, 3, 8
    work = np.zeros((C, 1, 8), dtype=np.int64)
    work[:, 0, :7] = [5, 4, 1, 2, 1, 6, 3]          # evaluated, updated, cloned, moved, move events, columns at resampling, splits
    stats = np.zeros((C, 8), dtype=np.int64); stats[:, 1] = 5
    n_s, N, D = n - n1 + 1, 4, 3
    per_chain = (5 * (16 * D + 4) + 4 * (32 * D + 8) + 2 * 32 * D                       # pool: evaluated, updated, moved
                 + 5 * 16 * P + 6 * N * 8 + 3 * N * 8                                     # resampling: ids + columns gathered, live columns, splits
                 + n_s * (8 * D + 9 * P) + (n1 - 1) * (8 * D + 8) + N * P * 12 + P * 4     # per step, prefix
                 + n * 4 + n_s)                                                            # output
    assert b.algorithmic_bytes(w, P, n, n1, work, stats) == C * per_chain
    assert b.algorithmic_bytes(w, P, n, n1, work, stats, layout="particle") > C * per_chain


def _literal_generate_psm(path, burnin, thin):
    """src/output_analysis/consensus_map.jl:31-65, line by line, in numpy (1-based arithmetic kept)."""
    from math import comb
    lines = [ln for ln in open(path).read().split("\n") if ln]
    names = lines[0].split(",")
    output = np.array([[float(x) for x in ln.split(",")] for ln in lines[burnin + 1:]])
    K = sum("MassParameter" in s for s in names)
    first = K + comb(K, 2) + (K == 1) + 2                      # 1-based first allocation column
    output = output[::thin, first - 1:]
    n_obs = output.shape[1] / K
    assert n_obs % 1 == 0
    n_obs, n_iter = int(n_obs), output.shape[0]
    psm = [np.eye(n_obs) for _ in range(K + (K > 1))]
    uniq = []
    for x in names[first - 1:]:
        if x.split("_")[0] not in uniq:
            uniq.append(x.split("_")[0])
    for k in range(1, K + 1):
        for j in range(1, n_obs):
            for i in range(j + 1, n_obs + 1):
                psm[k - 1][i - 1, j - 1] = np.sum(output[:, i + n_obs * (k - 1) - 1] == output[:, j + n_obs * (k - 1) - 1]) / n_iter
    if K > 1:
        for k in range(K):
            psm[K] += psm[k] / K
        psm[K][np.diag_indices(n_obs)] = 1.0
    return psm, uniq + (["Overall"] if K > 1 else [])


@pytest.mark.parametrize("K,burnin,thin", [(1, 0, 1), (2, 3, 2), (3, 1, 3)])
def test_generate_psm_reads_the_output_file_like_the_reference(pkg, tmp_path, K, burnin, thin):
    """SURVEY 8 f3/f4, reader side: file written by the native writer -> native reader (header, burn-in, thinning, the column
    offset with its K == 1 special case, names) -> PSM; against consensus_map.jl:31-65 restated literally."""
    from particlemdi_jl_amd.psm import generate_psm
    rng = np.random.default_rng(K)
    n, iters, N = 9, 14, 4
    names = ["alpha", "beta", "gamma"][:K]
    path = tmp_path / "out.csv"
    w = pkg.CsvWriter(path, K, n, data_names=names)
    rows = []
    for _ in range(iters):
        s = rng.integers(1, N + 1, size=(n, K))
        rows.append(s)
        w.row(rng.gamma(2.0, 1.0, K), rng.gamma(1.0, 1.0, max(1, K * (K - 1) // 2)), -123.5, s)
    w.close()
    smp, nm = pkg.read_allocations(path, burnin, thin)
    want = np.stack(rows[burnin::thin]).transpose(0, 2, 1)
    assert smp.shape == want.shape and (smp == want).all() and nm == names
    got = generate_psm(str(path), burnin, thin, host=True)
    ref, ref_names = _literal_generate_psm(path, burnin, thin)
    assert got.names == ref_names and len(got.psm) == len(ref)
    for a, b in zip(got.psm, ref):
        assert np.array_equal(a, b)


def test_read_allocations_rejects_what_the_reference_rejects(pkg, tmp_path):
    p = tmp_path / "bad.csv"
    p.write_text("MassParameter_1,MassParameter_2,phi_1_2,ll,a_n1,a_n2,b_n1\n1.0,1.0,0.5,-3.0,1.0,2.0,1.0\n")
    with pytest.raises(pkg.PmdiError, match="different number of observations"):      # consensus_map.jl:41
        pkg.read_allocations(p)
    p.write_text("MassParameter_1,phi_1_1,ll,a_n1,a_n2\n1.0,0.5,-3.0,1.5,2.0\n")
    with pytest.raises(pkg.PmdiError, match="not a label"):
        pkg.read_allocations(p)
    p.write_text("MassParameter_1,phi_1_1,ll,a_n1,a_n2\n1.0,0.5,-3.0,1.0\n")
    with pytest.raises(pkg.PmdiError, match="fields"):
        pkg.read_allocations(p)


def test_read_allocations_line_endings_and_short_files(pkg, tmp_path):
    p = tmp_path / "crlf.csv"
    p.write_bytes(b"MassParameter_1,phi_1_1,ll,a_n1,a_n2\r\n1.0,0.5,-3.0,1.0,2.0\r\n\r\n2.0,0.25,-2.0,2.0,2.0")      # CRLF, blank line, no final newline
    smp, names = pkg.read_allocations(p)
    assert names == ["a"] and smp.tolist() == [[[1, 2]], [[2, 2]]]
    smp, _ = pkg.read_allocations(p, burnin=1, thin=5)
    assert smp.tolist() == [[[2, 2]]]
    smp, _ = pkg.read_allocations(p, burnin=7)
    assert smp.shape == (0, 1, 2)
    from particlemdi_jl_amd.psm import generate_psm
    with pytest.raises(ValueError, match="no rows left"):
        generate_psm(str(p), burnin=7, host=True)
