"""TEST INFRASTRUCTURE: a second, independently written restatement of ParticleMDI's conditional-SMC sweep, used to
cross-check the C oracle (oracle/pmdi_oracle.c) in tests/test_oracle_second_opinion.py.

Where the C oracle re-engineers the reference's data structures (flat pools, incremental bookkeeping, 0-based
indices), this file stays as close to the Julia source as Python allows: mutable cluster OBJECTS in per-dataset lists
that are `deepcopy`'d (src/pmdi.jl:297,336), 1-based arrays (index 0 unused), the literal loops of src/pmdi.jl:165-350
and src/misc.jl:15-59, scalar libm calls (math.exp / math.log and libm's lgamma through ctypes = the same glibc the oracle links).  Pure
Python loops: small problems only.  Random numbers: the oracle's counter-based uniforms at the reference's draw sites
(src/pmdi.jl:253 allocation, src/misc.jl:28 and :43 resampling, src/pmdi.jl:350 particle pick).
"""
import copy
import ctypes
import ctypes.util
import math

import numpy as np

# CPython's math.lgamma is its own Lanczos code, not libm's: bind the C library's (what the oracle links and what
# SpecialFunctions v0.8.0's loggamma called through openlibm's port of the same fdlibm routine)
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.lgamma.restype = ctypes.c_double
_libm.lgamma.argtypes = [ctypes.c_double]
lgamma = _libm.lgamma

SITE_DRAW, SITE_RESAMPLE_U, SITE_RESAMPLE_SLOT, SITE_PSTAR = 0, 1, 2, 3


class GaussianCluster:                      # src/datatypes/gaussian_cluster.jl:11-22
    def __init__(self, data):
        D = data.shape[1]
        self.n = 0
        self.mu = [0.0] * D
        self.Sigma = [0.0] * D
        self.lam = [1.0] * D
        self.beta = [0.5] * D


def calc_logprob_gaussian(obs, cl, flag):   # gaussian_cluster.jl:37-52
    out = sum(flag) * (math.log(1 / math.sqrt(math.pi)) + lgamma(0.5 * cl.n + 1.0) - lgamma(0.5 * cl.n + 0.5))
    for q in range(len(obs)):
        if flag[q]:
            out += 0.5 * (math.log(cl.lam[q] / (cl.n + 1.0)))
            out -= (0.5 * cl.n + 1.0) * math.log(1.0 + (1.0 / (cl.n + 1.0)) * ((obs[q] - cl.mu[q]) ** 2.0) * cl.lam[q])
    return out


def cluster_add_gaussian(cl, obs, flag):    # gaussian_cluster.jl:54-66
    cl.n += 1
    for q in range(len(obs)):
        if flag[q]:
            cl.Sigma[q] += obs[q]
            cl.beta[q] += (cl.n - 1 + 0.001) * (obs[q] - cl.mu[q]) ** 2 / (2 * (cl.n + 0.001))
            cl.mu[q] = cl.Sigma[q] / (cl.n + 0.001)
            cl.lam[q] = ((0.5 * cl.n + 0.5) * (cl.n + 0.001)) / (cl.beta[q] * (cl.n + 1.001))


class CategoricalCluster:                   # categorical_cluster.jl:2-11
    def __init__(self, data):
        self.n = 0
        self.counts = np.zeros((int(data.max()) + 1, data.shape[1]), dtype=np.int64)      # row 0 unused (levels are 1-based)
        self.nlevels = [0.5 * float(data[:, d].max()) for d in range(data.shape[1])]


def calc_logprob_categorical(obs, cl, flag):    # categorical_cluster.jl:29-41
    acc = 0.0
    for q in range(len(obs)):
        if flag[q]:
            acc += math.log(cl.nlevels[q] + cl.n)
    out = -acc
    for q in range(len(obs)):
        if flag[q]:
            if cl.n == 0:
                out += math.log(0.5)
            else:
                out += math.log(0.5 + int(cl.counts[obs[q], q]))
    return out


def cluster_add_categorical(cl, obs, flag):     # categorical_cluster.jl:43-51
    cl.n += 1
    for q in range(len(obs)):
        if flag[q]:
            cl.counts[obs[q], q] += 1


class NegBinomCluster:                      # negbinom_cluster.jl:6-11
    def __init__(self, data):
        self.n = 0
        self.Sigma = [0] * data.shape[1]


def calc_logprob_negbinom(obs, cl, flag):   # negbinom_cluster.jl:22-41
    out = 0.0
    lg = lgamma
    for q in range(len(obs)):
        if flag[q]:
            x, S = int(obs[q]), cl.Sigma[q]
            out += lg(1 + cl.n + 1) + lg(1 + x + S) + lg(1 + cl.n + 1 + S) - lg(1 + cl.n + 1 + 1 + x + S) - lg(1 + cl.n) - lg(1 + S)
    return out


def cluster_add_negbinom(cl, obs, flag):    # negbinom_cluster.jl:43-51
    cl.n += 1
    for q in range(len(obs)):
        if flag[q]:
            cl.Sigma[q] += int(obs[q])


TYPES = {"gaussian": (GaussianCluster, calc_logprob_gaussian, cluster_add_gaussian),
         "categorical": (CategoricalCluster, calc_logprob_categorical, cluster_add_categorical),
         "negbinom": (NegBinomCluster, calc_logprob_negbinom, cluster_add_negbinom)}


def jl_cumsum(v):
    """Base.cumsum(::Vector{Float64}) = accumulate_pairwise! (blocks of 128)."""
    n = len(v)
    c = [0.0] * n
    if n == 0:
        return c
    c[0] = v[0]

    def rec(s, i1, m):
        if m < 128:
            s_ = v[i1]
            c[i1] = s + s_
            for i in range(i1 + 1, i1 + m):
                s_ = s_ + v[i]
                c[i] = s + s_
            return s_
        m2 = m >> 1
        s_ = rec(s, i1, m2)
        s_ = s_ + rec(s + s_, i1 + m2, m - m2)
        return s_
    if n > 1:
        rec(v[0], 1, n - 1)
    return c


def calc_ESS(logweight):                    # src/misc.jl:15-25
    num = den = 0.0
    max_l = max(logweight)
    for l in logweight:
        w = math.exp(l - max_l)
        num += w
        den += w ** 2
    return (num ** 2) / den


def draw_partstar(logweight, particles, u01, uslot):    # src/misc.jl:27-47
    u = u01 / particles
    max_l = max(logweight)
    pprob = jl_cumsum([math.exp(l - max_l) for l in logweight])
    partstar = []
    for p in range(1, particles + 1):
        while len(partstar) < particles and pprob[p - 1] / pprob[-1] >= u:
            u += 1 / particles
            partstar.append(p)
    while len(partstar) < particles:
        partstar.append(particles)
    # shuffle!; partstar[1] = 1; sort!  ==  one uniformly chosen element is replaced by 1
    j = min(int(uslot * particles), particles - 1)
    del partstar[j]
    return [1] + partstar


def sweep(data, kinds, N, P, s, order_obs, n1, Pi, Phi, flags, uniform, it, lw_init, q1_mode=0, q2_mode=0, shadow=None):
    """One iteration's sweep (src/pmdi.jl:165-171, 188-350, 373).  1-based like the reference: s (n+1, K+1) labels,
    order_obs list of 1-based rows, Pi (N+1, K+1).  uniform(it, pos, k, p, site) -> (0,1) with pos / k / p 0-based
    (the oracle's key layout).  shadow: an optional tests/_py_columns.ColumnShadow that is told of every change of
    particle[:, :, k] and checked against it (it never feeds back into the sweep).  Returns (s_new (n, K) 1-based labels, p_star, logweight list, counters)."""
    K, n = len(data), data[0].shape[0]
    new, logp, add = zip(*[TYPES[k] for k in kinds])
    logweight = [lw_init] * P
    # :165-171
    counts = [None] + [[0] * (N * P + 2) for _ in range(K)]
    for k in range(1, K + 1):
        counts[k][1] = P * N
    new_id = [None] + [[[0] * (P + 1) for _ in range(N + 1)] for _ in range(K)]          # new_id[k][n][class]
    particle_id = [None] + [[1] * (P + 1) for _ in range(K)]
    particle = [None] + [[[1] * (P + 1) for _ in range(N + 1)] for _ in range(K)]        # particle[k][n][p]
    clusters = [None] + [[None] * (N * P + 2) for _ in range(K)]
    sstar = [None] + [[[0] * (n + 1) for _ in range(P + 1)] for _ in range(K)]           # sstar[k][p][i]
    sstar_id = [None] + [[0] * (P + 1) for _ in range(K)]
    n_ops = n_res = n_clones = sum_cls = 0
    # :188-207
    for k in range(1, K + 1):
        clusters[k][1] = new[k - 1](data[k - 1])
        clust_ids, idn = {}, 2
        for i in order_obs[:n1 - 1]:
            u = s[i][k]
            if u not in clust_ids:                          # unique(), first appearance
                clusters[k][idn] = new[k - 1](data[k - 1])
                counts[k][idn] = P
                counts[k][1] -= P
                clust_ids[u] = idn
                for p in range(1, P + 1):
                    particle[k][u][p] = idn
                idn += 1
        for i in order_obs[:n1 - 1]:
            idc = clust_ids[s[i][k]]
            for p in range(1, P + 1):
                sstar[k][p][i] = s[i][k]
            add[k - 1](clusters[k][idc], data[k - 1][i - 1], flags[k - 1])
        if shadow:
            shadow.init(k, [0] + [particle[k][nn][1] for nn in range(1, N + 1)])
    # :209-342
    for pos in range(n1 - 1, n):
        i = order_obs[pos]
        for k in range(1, K + 1):
            if q1_mode == 1:
                new_id[k] = [[0] * (P + 1) for _ in range(N + 1)]
            fprob_done = [False] * (P + 1)
            fprob_dict = [[0.0] * (P + 1) for _ in range(N + 2)]
            cluster_update = {}
            obs = data[k - 1][i - 1]
            maxid = max(max(row[1:]) for row in particle[k][1:])
            logprob = [0.0] * (maxid + 1)
            for idc in range(1, maxid + 1):
                logprob[idc] = logp[k - 1](obs, clusters[k][idc], flags[k - 1])
            n_ops += maxid
            curr_id = 0
            for p in range(1, P + 1):
                idc = particle_id[k][p]
                if fprob_done[idc]:
                    fprob = [fprob_dict[nn][idc] for nn in range(1, N + 1)]
                    logweight[p - 1] += fprob_dict[N + 1][idc]
                else:
                    fprob = [logprob[particle[k][nn][p]] for nn in range(1, N + 1)]
                    mx = max(fprob)
                    for nn in range(N):
                        fprob[nn] -= mx
                        fprob[nn] = math.exp(fprob[nn])
                        fprob[nn] *= Pi[nn + 1][k]
                    fprob = jl_cumsum(fprob)
                    inc = math.log(fprob[N - 1]) + mx
                    fprob_dict[N + 1][idc] = inc
                    logweight[p - 1] += inc
                    last = fprob[N - 1]
                    fprob = [f / last for f in fprob]
                    for nn in range(1, N + 1):
                        fprob_dict[nn][idc] = fprob[nn - 1]
                    fprob_done[idc] = True
                    sum_cls += 1
                if p != 1:
                    new_s = 1
                    u = uniform(it, pos, k - 1, p - 1, SITE_DRAW)
                    for _ in range(1, N):
                        if fprob[new_s - 1] > u:
                            break
                        new_s += 1
                else:
                    new_s = s[i][k]
                sstar_id[k][p] = particle[k][new_s][p]
                sstar[k][p][i] = new_s
                if new_id[k][new_s][idc] == 0:
                    curr_id += 1
                    new_id[k][new_s][idc] = curr_id
                    particle_id[k][p] = curr_id
                else:
                    particle_id[k][p] = new_id[k][new_s][idc]
            max_k = max(max(row[1:]) for row in particle[k][1:])
            tgt_of = {}
            for pc in sstar_id[k][1:]:
                if pc not in cluster_update:
                    cluster_update[pc] = True
                    ncopies = sum(1 for x in sstar_id[k][1:] if x == pc)
                    if ncopies == counts[k][pc]:
                        idc = pc
                    else:
                        idc = max_k + 1
                        counts[k][pc] -= ncopies
                        counts[k][idc] = ncopies
                        clusters[k][idc] = copy.deepcopy(clusters[k][pc])
                        max_k += 1
                        n_clones += 1
                    add[k - 1](clusters[k][idc], obs, flags[k - 1])
                    tgt_of[pc] = idc
                    if idc != pc:
                        for part in range(1, P + 1):
                            s_id = sstar[k][part][i]
                            if particle[k][s_id][part] == pc:
                                particle[k][s_id][part] = idc
            if shadow:
                shadow.step(k, [0] + [sstar[k][p][i] for p in range(1, P + 1)], list(sstar_id[k]), tgt_of)
                shadow.check(k, particle[k])
        if K > 1:                                            # Phi_upweight! (src/misc.jl:50-59)
            pr = 0
            for k1 in range(1, K):
                for k2 in range(k1 + 1, K + 1):
                    phi_log = math.log(1 + Phi[pr])
                    for p in range(1, P + 1):
                        logweight[p - 1] += (sstar[k1][p][i] == sstar[k2][p][i]) * phi_log
                    pr += 1
        if calc_ESS(logweight) <= 0.5 * P:                   # :317
            n_res += 1
            partstar = draw_partstar(logweight, P, uniform(it, pos, 0, 0, SITE_RESAMPLE_U), uniform(it, pos, 0, 0, SITE_RESAMPLE_SLOT))
            logweight = [1.0] * P
            for k in range(1, K + 1):
                particle[k] = [None] + [[0] + [particle[k][nn][a] for a in partstar] for nn in range(1, N + 1)]
                particle_id[k] = [0] + [particle_id[k][a] for a in partstar]
                if q2_mode == 1:                             # src/__pmdi.jl:285 (pmdi() itself rebinds a local and drops it, :321-324)
                    sstar[k] = [None] + [list(sstar[k][a]) for a in partstar]
                counts[k] = [0] * (N * P + 2)
                ids = sorted({particle[k][nn][p] for nn in range(1, N + 1) for p in range(1, P + 1)})
                occ = shadow.resample(k, partstar, {idc: inew for inew, idc in enumerate(ids, start=1)}) if shadow else None
                for inew, idc in enumerate(ids, start=1):
                    if idc != inew:
                        for nn in range(1, N + 1):
                            for p in range(1, P + 1):
                                if particle[k][nn][p] == idc:
                                    particle[k][nn][p] = inew
                        clusters[k][inew] = copy.deepcopy(clusters[k][idc])
                    counts[k][inew] = sum(1 for nn in range(1, N + 1) for p in range(1, P + 1) if particle[k][nn][p] == inew)
                if shadow:
                    shadow.check(k, particle[k])
                    assert occ == {inew: counts[k][inew] for inew in range(1, len(ids) + 1)}
    # :345-350, StatsBase sample(::Weights)
    mx = max(logweight)
    w = [math.exp(l - mx) for l in logweight]
    tot = 0.0
    for x in w:
        tot += x
    t = uniform(it, 0, 0, 0, SITE_PSTAR) * tot
    p_star, cw = 1, w[0]
    while cw < t and p_star < P:
        p_star += 1
        cw += w[p_star - 1]
    s_new = np.array([[sstar[k][p_star][i] for k in range(1, K + 1)] for i in range(1, n + 1)], dtype=np.int64)   # :373
    state = {"particle": particle, "counts": counts, "clusters": clusters}
    return s_new, p_star, logweight, {"n_operations": n_ops, "n_resamples": n_res, "n_clones": n_clones, "sum_classes": sum_cls}, state
