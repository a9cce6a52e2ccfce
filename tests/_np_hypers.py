"""TEST INFRASTRUCTURE: a second, independently written restatement of src/update_hypers.jl and
align_labels! (src/misc.jl:61-96) on numpy arrays, used to cross-check the C oracle
(oracle/pmdi_oracle_hypers.c) in tests/test_oracle_hypers.py.  It was the product's host mirror in
round 1; the product now runs these updates on the device (csrc/pmdi_hypers.hip).

The N^K tables (c_combn, Gamma_c, Phi_index; src/pmdi.jl:69-92) are held as K-dimensional arrays:
row i of the reference's tables is the base-N digit vector of i with column k the (k-1)-th digit,
i.e. axis k-1 of an (N,)*K array.

Draws: `draws` is an object with uniform/normal/gamma(…, it, pos, k, site) methods -- the oracle's
Philox samplers in the cross-check (same variates as the C restatement), numpy's Generator otherwise.
"""
import numpy as np
from scipy.special import gammaln
from scipy.stats import binom, gamma as gamma_dist

EPS = np.finfo(np.float64).eps


def phi_lab(K):
    """calculate_Phi_lab (src/misc.jl:1-13), 0-based pairs."""
    if K < 2:
        return np.zeros((1, 2), dtype=np.int64)
    return np.array([(a, b) for a in range(K - 1) for b in range(a + 1, K)], dtype=np.int64)


class NumpyDraws:
    def __init__(self, rng):
        self.rng = rng

    def uniform(self, it, pos, k, site):
        return self.rng.random()

    def normal(self, it, pos, k, site):
        return self.rng.normal()

    def gamma(self, shape, it, pos, k, site):
        return self.rng.gamma(shape, 1.0)


class HyperState:
    """M, gamma, Phi, v, Z of one chain (src/pmdi.jl:59-96)."""

    def __init__(self, n_obs, N, K, rng, draws=None):
        self.n, self.N, self.K, self.rng = int(n_obs), int(N), int(K), rng
        self.draws = draws or NumpyDraws(rng)
        self.it = 0
        self.M = np.ones(K) * 2.0                                           # :59
        self.gamma = rng.gamma(1.0 / N, 1.0, size=(N, K)) + EPS             # :60
        self.npairs = K * (K - 1) // 2 if K > 1 else 1
        self.Phi = rng.gamma(1.0, 0.2, size=self.npairs) if K > 1 else np.zeros(1)   # :61
        self.pairs = phi_lab(K)
        # s[:, k] = sampleCategorical(n, gamma[:, k])  (:63-66)
        self.s = np.empty((self.n, K), dtype=np.int64)
        for k in range(K):
            p = self.gamma[:, k] / self.gamma[:, k].sum()
            self.s[:, k] = rng.choice(N, size=self.n, p=p) + 1
        # Gamma_c: log(gamma) gathered over all combinations, built ONCE from the
        # initial gamma and never refreshed (src/pmdi.jl:75-79)
        self._sumGamma = self._outer_sum(np.log(self.gamma))
        self.Z = self.update_Z()
        self.v = self.update_v()

    # -- helpers over the (N,)*K table ----------------------------------------
    def _outer_sum(self, cols):
        K, N = self.K, self.N
        out = np.zeros((N,) * K)
        for k in range(K):
            shape = [1] * K
            shape[k] = N
            out = out + cols[:, k].reshape(shape)
        return out

    def _agree_mask(self, pair):
        K, N = self.K, self.N
        a, b = self.pairs[pair]
        ia = np.arange(N).reshape([N if d == a else 1 for d in range(K)])
        ib = np.arange(N).reshape([N if d == b else 1 for d in range(K)])
        return np.broadcast_to(ia == ib, (N,) * K)

    def _norm_temp(self):
        nt = self._sumGamma.copy()
        if self.K > 1:
            phi_log = np.log(self.Phi + 1.0)
            for i in range(self.npairs):
                nt = nt + self._agree_mask(i) * phi_log[i]
        return np.exp(nt)

    # -- src/update_hypers.jl -------------------------------------------------
    def update_v(self):                                                     # :1-3
        self.v = self.draws.gamma(self.n, self.it, 0, 0, 11) * (1.0 / self.Z)
        return self.v

    def update_Z(self):                                                     # :29-39
        self.Z = float(self._norm_temp().sum())
        return self.Z

    def update_M(self):                                                     # :5-26
        N = self.N
        for k in range(self.K):
            g = self.gamma[:, k]
            cur = self.M[k]
            ll = -gamma_dist.logpdf(g, cur / N, scale=1.0).sum()
            ll0 = -gamma_dist.logpdf(cur, 2.0, scale=0.25)
            prop = cur + self.draws.normal(self.it, 0, k, 6) / 10.0
            if prop <= 0.0:
                alpha = 0.0
            else:
                nll = -gamma_dist.logpdf(g, prop / N, scale=1.0).sum()
                nll0 = -gamma_dist.logpdf(prop, 2.0, scale=0.25)
                with np.errstate(over="ignore"):
                    alpha = np.exp(-nll - nll0 + ll + ll0)
            if self.draws.uniform(self.it, 0, k, 7) < alpha:
                self.M[k] = prop

    def update_gamma(self):                                                 # :64-92
        N, K = self.N, self.K
        alpha_star = np.empty((N, K))
        for k in range(K):
            alpha_star[:, k] = self.M[k] / N + np.bincount(self.s[:, k] - 1, minlength=N)
        nt = self._norm_temp()
        for k in range(K):
            for nn in range(N):
                sl = [slice(None)] * K
                sl[k] = nn
                sl = tuple(sl)
                old = self.gamma[nn, k] + 0.0
                beta_star = 1.0 + self.v * nt[sl].sum() / self.gamma[nn, k]
                self.gamma[nn, k] = self.draws.gamma(alpha_star[nn, k], self.it, nn, k, 8) * (1.0 / beta_star) + EPS
                nt[sl] *= self.gamma[nn, k] / old

    def update_Phi(self):                                                   # :95-128
        if self.K < 2:
            return
        nt = self._norm_temp()
        for i in range(self.npairs):
            a, b = self.pairs[i]
            cur = self.Phi[i] + 0.0
            n_agree = int((self.s[:, a] == self.s[:, b]).sum())
            mask = self._agree_mask(i)
            beta_star = 5.0 + (self.v * nt[mask].sum() / (1.0 + cur))
            r = np.arange(n_agree + 1)
            w = gammaln(r + 1.0) + binom.logpmf(r, n_agree, 0.5) - r * np.log(1.0 / beta_star)
            w = np.exp(w - w.max())
            t = self.draws.uniform(self.it, 0, i, 9) * w.sum()
            pick = min(int(np.searchsorted(np.cumsum(w), t, side="left")), n_agree)   # StatsBase sample(::Weights)
            self.pick = pick
            alpha_star = 1.0 + pick
            self.Phi[i] = self.draws.gamma(alpha_star, self.it, 0, i, 10) * (1.0 / beta_star)
            nt[mask] *= (1.0 + self.Phi[i]) / (1.0 + cur)

    def Pi(self):                                                           # src/pmdi.jl:179
        return self.gamma / self.gamma.sum(axis=0, keepdims=True)

    def step_pmdi_order(self):
        """Hyper updates in pmdi()'s order (src/pmdi.jl:176-185): M, gamma, Pi, Phi, Z, v."""
        self.update_M()
        self.update_gamma()
        Pi = self.Pi()
        self.update_Phi()
        self.update_Z()
        self.update_v()
        return Pi

    # -- align_labels! (src/misc.jl:61-96) ------------------------------------
    def align_labels(self):
        """align_labels! through N x N contingency tables.  The reference recounts
        `count_equals(label_rows, ...)` over the n observations for every (label, new_label) pair; the
        counts it needs are entries of T[k][j][a, b] = #{i : s[i, k] == a and s[i, j] == b}:
            count_equals(label_rows, label)[j]     = T[k][j][label, label]
            count_equals(new_rows, new_label)[j]   = T[k][j][new, new]
            count_equals(label_rows, new_label)[j] = T[k][j][label, new]
            count_equals(new_rows, label)[j]       = T[k][j][new, label]
        and an accepted swap exchanges two rows of T[k][j] (and two columns of T[j][k]).  Same integers,
        same floating-point expressions in the same order, same random numbers consumed: the result is
        identical to `_align_labels_by_recount` (the line-by-line restatement, kept as the test oracle)."""
        K, N = self.K, self.N
        if K == 1:
            return
        s, gam = self.s, self.gamma
        phi_log = np.log(self.Phi + 1.0)
        z = s - 1
        T = {(k, j): np.bincount(z[:, k] * N + z[:, j], minlength=N * N).reshape(N, N).astype(np.int64)
             for k in range(K) for j in range(K) if j != k}
        for k in range(K):
            others = [j for j in range(K) if j != k]
            rel = np.array([phi_log[i] for i in range(self.npairs)
                            if self.pairs[i][0] == k or self.pairs[i][1] == k])
            col = s[:, k]
            _, first = np.unique(col, return_index=True)
            occupied = col[np.sort(first)].tolist()                          # unique(), first appearance
            perm = np.arange(N + 1)                                          # label of the start of this k -> label now
            for oi, label in enumerate(occupied):
                a = label - 1
                if T[(k, others[0])][a].sum() == 0:                          # all(label_ind .== false)
                    continue
                for new_label in range(1, N + 1):
                    if new_label == label:
                        continue
                    b = new_label - 1
                    c_ll = np.array([T[(k, j)][a, a] for j in others], dtype=np.float64)
                    c_nn = np.array([T[(k, j)][b, b] for j in others], dtype=np.float64)
                    c_ln = np.array([T[(k, j)][a, b] for j in others], dtype=np.float64)
                    c_nl = np.array([T[(k, j)][b, a] for j in others], dtype=np.float64)
                    lps = (c_ll * rel + c_nn * rel).sum()
                    lps_swap = (c_ln * rel + c_nl * rel).sum()
                    with np.errstate(over="ignore"):
                        accept = np.exp(lps_swap - lps)
                    if self.draws.uniform(self.it, oi * N + new_label - 1, k, 12) < accept:
                        for j in others:
                            T[(k, j)][[a, b], :] = T[(k, j)][[b, a], :]
                            T[(j, k)][:, [a, b]] = T[(j, k)][:, [b, a]]
                        ia, ib = perm == label, perm == new_label
                        perm[ia], perm[ib] = new_label, label
                        gam[b, k], gam[a, k] = gam[a, k], gam[b, k]
                        label = new_label
                        a = label - 1
            s[:, k] = perm[col]

    def _align_labels_by_recount(self):
        K, N = self.K, self.N
        if K == 1:
            return
        s, gam = self.s, self.gamma
        phi_log = np.log(self.Phi + 1.0)
        for k in range(K):
            others = [j for j in range(K) if j != k]
            rel = np.array([phi_log[i] for i in range(self.npairs)
                            if self.pairs[i][0] == k or self.pairs[i][1] == k])
            # pair order in relevant_Phis follows Phi_lab; columns of label_rows follow `others`;
            # both enumerate the other datasets in increasing order, as the reference does
            occupied = list(dict.fromkeys(s[:, k].tolist()))               # unique(), first appearance
            for oi, label in enumerate(occupied):
                label_ind = s[:, k] == label
                if not label_ind.any():
                    continue
                label_rows = s[label_ind][:, others]
                for new_label in range(1, N + 1):
                    if new_label == label:
                        continue
                    new_ind = s[:, k] == new_label
                    new_rows = s[new_ind][:, others]
                    ce = lambda A, b: (A == b).sum(axis=0).astype(np.float64)
                    lps = (ce(label_rows, label) * rel + ce(new_rows, new_label) * rel).sum()
                    lps_swap = (ce(label_rows, new_label) * rel + ce(new_rows, label) * rel).sum()
                    with np.errstate(over="ignore"):
                        accept = np.exp(lps_swap - lps)
                    if self.draws.uniform(self.it, oi * N + new_label - 1, k, 12) < accept:
                        s[label_ind, k] = new_label
                        s[new_ind, k] = label
                        gam[new_label - 1, k], gam[label - 1, k] = gam[label - 1, k], gam[new_label - 1, k]
                        label = new_label
                        label_ind = s[:, k] == label
                        label_rows = s[label_ind][:, others]
