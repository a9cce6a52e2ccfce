"""TEST INFRASTRUCTURE: the device's representation of particle[:, :, k] -- its DISTINCT COLUMNS plus a column index per
particle (particlemdi.jl_amd/csrc/pmdi_sweep.hip: columns_apply, the unanimous-step store, the per-column resampling pass;
DESIGN.md section 4.3) -- restated in Python and driven as a SHADOW of the literal N x P table of tests/_py_sweep.py:
after every step and every resampling event the expanded columns must equal the reference's table.  This checks the
algorithm (grouping rule, in-place rule, compaction, occupancy = multiplicity x entries, "never more than P columns,
never an empty one") on the CPU; the HIP code itself is checked against the oracle's exported table by the GPU tests.

Indices here are 0-based for columns and particles, labels 1..N as in _py_sweep.
"""
import random


class ColumnShadow:
    def __init__(self, K, N, P, order_seed=None):
        self.K, self.N, self.P = K, N, P
        self.tab = [None] * (K + 1)       # tab[k][c] = list of N + 1 ids (index 0 unused)
        self.col = [None] * (K + 1)       # col[k][p], p = 1..P
        self.rng = random.Random(order_seed) if order_seed is not None else None    # which group asks first is arbitrary on the device
        self.max_cols = 0
        self.splits = 0

    # -- known prefix (src/pmdi.jl:169-171,195-198): every particle on one column
    def init(self, k, column):
        self.tab[k] = [list(column)]
        self.col[k] = [0] * (self.P + 1)

    def get(self, k, label, p):
        return self.tab[k][self.col[k][p]][label]

    # -- one (observation, dataset) step: labels drawn, chosen clusters, chosen -> updated id (identity if not cloned)
    def step(self, k, news, chosen, tgt_of):
        P, tab, col = self.P, self.tab[k], self.col[k]
        for p in range(1, P + 1):
            assert tab[col[p]][news[p]] == chosen[p]
        if len({(news[p], chosen[p]) for p in range(1, P + 1)}) == 1:
            # unanimous step: every live column holds the chosen cluster under the drawn label
            ns0, c0 = news[1], chosen[1]
            if tgt_of[c0] != c0:
                for c in range(len(tab)):
                    assert tab[c][ns0] == c0
                    tab[c][ns0] = tgt_of[c0]
            return
        writer = [False] + [tgt_of[chosen[p]] != chosen[p] for p in range(1, P + 1)]
        keep = {col[p] for p in range(1, P + 1) if not writer[p]}            # columns that keep a particle which does not write
        groups = {}
        for p in range(1, P + 1):
            if writer[p]:
                groups.setdefault((col[p], news[p]), []).append(p)
        keys = list(groups)
        if self.rng:
            self.rng.shuffle(keys)
        reused, newcol = set(), {}
        originals = {c: list(tab[c]) for c, _ in keys}                        # copies are made from the originals ...
        for (c, ns) in keys:
            tgt = tgt_of[chosen[groups[(c, ns)][0]]]
            if c not in keep and c not in reused:
                reused.add(c); newcol[(c, ns)] = (c, tgt, True)
            else:
                cp = list(originals[c]); cp[ns] = tgt
                tab.append(cp); newcol[(c, ns)] = (len(tab) - 1, tgt, False)
                self.splits += 1
        for (c, ns), (nc, tgt, inplace) in newcol.items():                    # ... the in-place entries are written afterwards
            if inplace:
                tab[c][ns] = tgt
            for p in groups[(c, ns)]:
                col[p] = nc
        self._invariants(k)

    # -- a resampling event: ancestors, then the renumbering old id -> new id (src/pmdi.jl:322,329-337)
    def resample(self, k, partstar, relabel):
        P, N, tab = self.P, self.N, self.tab[k]
        newc = [0] + [self.col[k][a] for a in partstar]
        mult = [0] * len(tab)
        for p in range(1, P + 1):
            mult[newc[p]] += 1
        cmap, out = {}, []
        for c in range(len(tab)):
            if mult[c]:
                cmap[c] = len(out)
                out.append([0] + [relabel[v] for v in tab[c][1:]])
        occupancy = {}
        for c in range(len(tab)):
            for v in tab[c][1:]:
                if mult[c]:
                    occupancy[relabel[v]] = occupancy.get(relabel[v], 0) + mult[c]
        self.tab[k] = out
        self.col[k] = [0] + [cmap[newc[p]] for p in range(1, P + 1)]
        self._invariants(k)
        return occupancy

    def _invariants(self, k):
        used = {self.col[k][p] for p in range(1, self.P + 1)}
        assert used == set(range(len(self.tab[k]))), "a column without a particle, or a particle without a column"
        assert len(self.tab[k]) <= self.P
        self.max_cols = max(self.max_cols, len(self.tab[k]))

    def check(self, k, particle_k):
        """particle_k[label][p] (1-based) is the reference's table."""
        for nn in range(1, self.N + 1):
            for p in range(1, self.P + 1):
                assert self.get(k, nn, p) == particle_k[nn][p], (k, nn, p)
