"""src/misc.jl helpers and the RNG, as restated by the oracle."""
import numpy as np


def test_philox_known_answers(O):
    # Random123 kat_vectors for philox4x32-10
    assert [hex(v) for v in O.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in O.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_uniform_range_and_mean(O):
    u = np.array([O.uniform(7, 1, pos, 0, p, 0) for pos in range(50) for p in range(200)])
    assert (u >= 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 0.01
    assert O.uniform(7, 1, 3, 0, 5, 0) != O.uniform(7, 1, 3, 0, 5, 1)    # the site decorrelates


def test_ess(O):
    P = 257
    assert O.calc_ess(np.full(P, 3.7)) == P                       # uniform weights
    lw = np.full(P, -1e9); lw[5] = 0.0
    assert O.calc_ess(lw) == 1.0                                  # one-hot
    rng = np.random.default_rng(0)
    lw = rng.normal(size=P)
    w = np.exp(lw - lw.max())
    assert np.isclose(O.calc_ess(lw), w.sum() ** 2 / (w ** 2).sum(), rtol=1e-12)


def jl_cumsum(v):
    """Julia Base.accumulate_pairwise! (base/accumulate.jl), restated independently in Python."""
    v = list(map(float, v)); n = len(v); c = [0.0] * n
    def rec(s, i1, m):
        if m < 128:
            s_ = v[i1]; c[i1] = s + s_
            for i in range(i1 + 1, i1 + m):
                s_ = s_ + v[i]; c[i] = s + s_
            return s_
        n2 = m >> 1
        s_ = rec(s, i1, n2)
        s_ += rec(s + s_, i1 + n2, m - n2)
        return s_
    c[0] = v[0]
    if n > 1:
        rec(v[0], 1, n - 1)
    return np.array(c)


def test_draw_partstar(O):
    P = 1024
    # uniform weights: systematic resampling picks every particle once; slot 1 forced to 1
    ps = O.draw_partstar(np.zeros(P), 0.37, 0.5)
    want = np.arange(1, P + 1); j = int(0.5 * P)
    want = np.concatenate([[1], want[:j], want[j + 1:]])
    assert (ps == want).all()
    # general weights: counts follow the pairwise-cumsum CDF and repeated-addition u
    rng = np.random.default_rng(3)
    lw = rng.normal(scale=2.0, size=P)
    ps = O.draw_partstar(lw, 0.81, 0.0)        # slot 0 replaced by 1: partstar[0] = 1 anyway
    cdf = jl_cumsum(np.exp(lw - lw.max()))
    u = 0.81 / P; raw = []
    for p in range(P):
        while len(raw) < P and cdf[p] / cdf[-1] >= u:
            u += 1.0 / P; raw.append(p + 1)
    raw = np.array(raw)
    assert ps[0] == 1 and (ps[1:] == raw[1:]).all() and (np.diff(ps) >= 0).all()


def test_phi_upweight(O):
    rng = np.random.default_rng(4)
    P, K = 50, 3
    ss = rng.integers(1, 4, size=(P, K))
    Phi = np.array([0.3, 1.1, 0.05])
    lw0 = rng.normal(size=P)
    got = O.phi_upweight(lw0, ss, Phi)
    want = lw0.copy()
    pairs = [(0, 1), (0, 2), (1, 2)]
    for i, (a, b) in enumerate(pairs):
        want += (ss[:, a] == ss[:, b]) * np.log(1 + Phi[i])
    assert np.allclose(got, want, rtol=0, atol=1e-15)


def test_psm_counts_known_answers(O):
    # generate_psm's inner expression (consensus_map.jl:53): sum(output[:, i] .== output[:, j])
    rng = np.random.default_rng(8)
    S, K, n = 37, 2, 53
    smp = rng.integers(1, 6, size=(S, K, n)).astype(np.uint8)
    smp[:, 0, 7] = smp[:, 0, 3]                       # two observations always together
    smp[:, 1, 9] = 200                                # a label nobody else uses
    got = O.psm_counts(smp, 0, n)
    want = (smp[:, :, :, None] == smp[:, :, None, :]).sum(axis=0).astype(np.int32)
    assert (got == want).all()
    assert (got[:, np.arange(n), np.arange(n)] == S).all() and got[0, 7, 3] == S and (np.delete(got[1, 9], 9) == 0).all()
    blk = O.psm_counts(smp, 11, 30)                   # a block of rows == the same rows of the full matrix
    assert (blk == want[:, 11:30, :]).all()
