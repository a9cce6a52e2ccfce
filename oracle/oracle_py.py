"""ctypes binding of the CPU oracle (oracle/pmdi_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpmdi_oracle.so")

GAUSSIAN, CATEGORICAL, NEGBINOM = 0, 1, 2
KIND_BY_NAME = {"gaussian": GAUSSIAN, "categorical": CATEGORICAL, "negbinom": NEGBINOM}


def build(force=False):
    deps = [os.path.join(_HERE, f) for f in ("pmdi_oracle.c", "pmdi_oracle_hypers.c", "pmdi_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(d) for d in deps)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libpmdi_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Dataset(C.Structure):
    _fields_ = [("kind", C.c_int32), ("D", C.c_int32),
                ("xf", C.POINTER(C.c_double)), ("xi", C.POINTER(C.c_int64))]


class Stats(C.Structure):
    _fields_ = [("n_operations", C.c_int64), ("n_resamples", C.c_int64),
                ("n_clones", C.c_int64), ("max_id", C.c_int64),
                ("sum_classes", C.c_int64), ("seconds", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    L.pmdi_oracle_create.restype = vp
    L.pmdi_oracle_create.argtypes = [i32, i64, i32, i32, C.POINTER(_Dataset), u64, i32, i32, i32]
    L.pmdi_oracle_destroy.argtypes = [vp]
    L.pmdi_oracle_sweep.restype = C.c_int
    L.pmdi_oracle_sweep.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, dbl, vp, vp, vp, vp, vp]
    L.pmdi_oracle_feature_select.restype = C.c_int
    L.pmdi_oracle_feature_select.argtypes = [vp, i64, vp, vp, vp]
    L.pmdi_oracle_work.restype = None
    L.pmdi_oracle_work.argtypes = [vp, vp, vp]
    L.pmdi_oracle_debug_steps.restype = C.c_int
    L.pmdi_oracle_debug_steps.argtypes = [vp, vp]
    L.pmdi_oracle_debug_columns.restype = None
    L.pmdi_oracle_debug_columns.argtypes = [vp, vp]
    L.pmdi_oracle_export.restype = C.c_int
    L.pmdi_oracle_export.argtypes = [vp, vp, vp, vp, vp]
    L.pmdi_oracle_cluster_new.restype = vp
    L.pmdi_oracle_cluster_new.argtypes = [C.POINTER(_Dataset), i64]
    L.pmdi_oracle_cluster_free.argtypes = [vp]
    L.pmdi_oracle_cluster_add.argtypes = [vp, i64, vp]
    L.pmdi_oracle_cluster_logprob.restype = dbl
    L.pmdi_oracle_cluster_logprob.argtypes = [vp, i64, vp]
    L.pmdi_oracle_cluster_logmarginal.argtypes = [vp, vp]
    L.pmdi_oracle_cluster_stats.restype = i64
    L.pmdi_oracle_cluster_stats.argtypes = [vp, vp]
    L.pmdi_oracle_psm_counts.restype = None
    L.pmdi_oracle_psm_counts.argtypes = [vp, i64, C.c_int32, i64, i64, i64, vp]
    L.pmdi_oracle_calc_ess.restype = dbl
    L.pmdi_oracle_calc_ess.argtypes = [vp, i64]
    L.pmdi_oracle_draw_partstar.argtypes = [vp, i64, dbl, dbl, vp]
    L.pmdi_oracle_phi_upweight.argtypes = [vp, vp, i32, vp, i64]
    L.pmdi_oracle_philox4x32_10.argtypes = [vp, vp, vp]
    L.pmdi_oracle_uniform.restype = dbl
    L.pmdi_oracle_uniform.argtypes = [u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.pmdi_oracle_hypers_create.restype = vp
    L.pmdi_oracle_hypers_create.argtypes = [i64, i32, i32, u64]
    L.pmdi_oracle_hypers_destroy.argtypes = [vp]
    for f in ("update_M", "update_gamma", "update_Phi", "align_labels", "shuffle"):
        getattr(L, "pmdi_oracle_hypers_" + f).restype = None
        getattr(L, "pmdi_oracle_hypers_" + f).argtypes = [vp, i64]
    L.pmdi_oracle_hypers_update_Z.restype = dbl
    L.pmdi_oracle_hypers_update_Z.argtypes = [vp]
    L.pmdi_oracle_hypers_update_v.restype = dbl
    L.pmdi_oracle_hypers_update_v.argtypes = [vp, i64]
    L.pmdi_oracle_hypers_step.restype = None
    L.pmdi_oracle_hypers_step.argtypes = [vp, i64, vp]
    L.pmdi_oracle_hypers_get.argtypes = [vp, C.c_int, vp]
    L.pmdi_oracle_hypers_set.argtypes = [vp, C.c_int, vp]
    L.pmdi_oracle_hypers_s.restype = C.POINTER(C.c_int64)
    L.pmdi_oracle_hypers_s.argtypes = [vp]
    L.pmdi_oracle_hypers_order.restype = C.POINTER(C.c_int64)
    L.pmdi_oracle_hypers_order.argtypes = [vp]
    L.pmdi_oracle_normal.restype = dbl
    L.pmdi_oracle_normal.argtypes = [u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.pmdi_oracle_gamma.restype = dbl
    L.pmdi_oracle_gamma.argtypes = [dbl, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _mk_datasets(data, kinds):
    """data: list of (n, D) arrays; returns (ctypes array, keepalive list)."""
    K = len(data)
    arr = (_Dataset * K)()
    keep = []
    for k, (x, kind) in enumerate(zip(data, kinds)):
        kind = KIND_BY_NAME.get(kind, kind)
        arr[k].kind = kind
        arr[k].D = x.shape[1]
        if kind == GAUSSIAN:
            xf = np.asfortranarray(x, dtype=np.float64)
            keep.append(xf)
            arr[k].xf = xf.ctypes.data_as(C.POINTER(C.c_double))
            arr[k].xi = None
        else:
            xi = np.asfortranarray(x, dtype=np.int64)
            keep.append(xi)
            arr[k].xi = xi.ctypes.data_as(C.POINTER(C.c_int64))
            arr[k].xf = None
    return arr, keep


class Oracle:
    """One chain of the restated sweep (src/pmdi.jl:165-172,188-350)."""

    def __init__(self, data, kinds, N, P, seed=0, q1_mode=0, q2_mode=0, faithful_cost=0):
        self.L = lib()
        self.K = len(data)
        self.n = int(data[0].shape[0])
        self.N, self.P = int(N), int(P)
        self.D = [int(x.shape[1]) for x in data]
        self._ds, self._keep = _mk_datasets(data, kinds)
        self.h = self.L.pmdi_oracle_create(self.K, self.n, self.N, self.P, self._ds,
                                           int(seed), q1_mode, q2_mode, faithful_cost)
        if not self.h:
            raise ValueError("pmdi_oracle_create rejected the configuration")

    def close(self):
        if self.h:
            self.L.pmdi_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sweep(self, it, s, order_obs, n1, Pi, Phi, flags=None, lw_init=None, trace=False):
        """s: (n, K) labels 1..N; order_obs: (n,) 1-based; Pi: (N, K)."""
        K, n, N, P = self.K, self.n, self.N, self.P
        s_in = np.asfortranarray(np.asarray(s, dtype=np.int64).reshape(n, K))
        order = np.ascontiguousarray(order_obs, dtype=np.int64)
        Pi_f = np.asfortranarray(np.asarray(Pi, dtype=np.float64).reshape(N, K))
        Phi_c = np.ascontiguousarray(np.atleast_1d(Phi), dtype=np.float64)
        if flags is None:
            flags = [np.ones(d, dtype=np.uint8) for d in self.D]
        flags = [np.ascontiguousarray(f, dtype=np.uint8) for f in flags]
        fl = (C.c_void_p * K)(*[f.ctypes.data for f in flags])
        if lw_init is None:
            lw_init = 0.0 if it == 1 else 1.0
        s_out = np.zeros((n, K), dtype=np.int64, order="F")
        lw = np.zeros(P, dtype=np.float64)
        p_star = C.c_int64(0)
        st = Stats()
        tr = None
        if trace:
            tr = np.zeros((n - n1 + 1, 2 + 2 * K), dtype=np.float64)
        rc = self.L.pmdi_oracle_sweep(self.h, int(it), _ptr(s_in), _ptr(order), int(n1), _ptr(Pi_f),
                                      _ptr(Phi_c), C.cast(fl, C.c_void_p), float(lw_init),
                                      _ptr(s_out), _ptr(lw), C.addressof(p_star), C.addressof(st),
                                      _ptr(tr))
        if rc != 0:
            raise RuntimeError(f"pmdi_oracle_sweep failed rc={rc}")
        out = {"s": s_out, "logweight": lw, "p_star": int(p_star.value), "stats": st.as_dict()}
        if trace:
            out["trace"] = tr
        return out

    def feature_select(self, it, s_traj):
        K, n = self.K, self.n
        s_f = np.asfortranarray(np.asarray(s_traj, dtype=np.int64).reshape(n, K))
        flags = [np.zeros(d, dtype=np.uint8) for d in self.D]
        probs = [np.zeros(d, dtype=np.float64) for d in self.D]
        fl = (C.c_void_p * K)(*[f.ctypes.data for f in flags])
        pr = (C.c_void_p * K)(*[f.ctypes.data for f in probs])
        rc = self.L.pmdi_oracle_feature_select(self.h, int(it), _ptr(s_f), C.cast(fl, C.c_void_p),
                                               C.cast(pr, C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"pmdi_oracle_feature_select failed rc={rc}")
        return flags, probs

    def debug_steps(self, n_swept):
        """Switch the per-step record on: returns the (n_swept, K, 8) int64 array the next sweeps fill
        (columns documented at pmdi_oracle_debug_steps in pmdi_oracle.c); debug_steps(0) switches it off."""
        if not n_swept:
            self.L.pmdi_oracle_debug_steps(self.h, None)
            self._dbg = None
            return None
        self._dbg = np.zeros((int(n_swept), self.K, 8), dtype=np.int64)
        if self.L.pmdi_oracle_debug_steps(self.h, _ptr(self._dbg)) != 0:
            raise MemoryError("pmdi_oracle_debug_steps")
        return self._dbg

    def work(self):
        """(updates[K], moved[K]) of the last sweep: cluster_add! calls (:300), renumbering deepcopies (:336)."""
        up = np.zeros(self.K, dtype=np.int64); mv = np.zeros(self.K, dtype=np.int64)
        self.L.pmdi_oracle_work(self.h, _ptr(up), _ptr(mv))
        return up, mv

    def export(self):
        K, N, P = self.K, self.N, self.P
        cap = N * P + 1
        particle = np.zeros((K, P, N), dtype=np.int64)
        counts = np.zeros((K, cap), dtype=np.int64)
        cn = np.zeros((K, cap), dtype=np.int64)
        mx = np.zeros(K, dtype=np.int64)
        self.L.pmdi_oracle_export(self.h, _ptr(particle), _ptr(counts), _ptr(cn), _ptr(mx))
        return {"particle": particle, "counts": counts, "cluster_n": cn, "max_id": mx}


class Cluster:
    """Stand-alone cluster: the calc_logprob / cluster_add! / calc_logmarginal protocol."""

    def __init__(self, x, kind):
        self.L = lib()
        self.kind = KIND_BY_NAME.get(kind, kind)
        self.n_rows, self.D = x.shape
        self._ds, self._keep = _mk_datasets([x], [self.kind])
        self.h = self.L.pmdi_oracle_cluster_new(self._ds, self.n_rows)
        if not self.h:
            raise ValueError("bad dataset")
        self.Lmax = int(np.max(x)) if self.kind == CATEGORICAL else 0

    def add(self, row, flag=None):
        f = None if flag is None else np.ascontiguousarray(flag, dtype=np.uint8)
        self.L.pmdi_oracle_cluster_add(self.h, int(row), _ptr(f))

    def logprob(self, row, flag=None):
        f = None if flag is None else np.ascontiguousarray(flag, dtype=np.uint8)
        return self.L.pmdi_oracle_cluster_logprob(self.h, int(row), _ptr(f))

    def logmarginal(self):
        out = np.zeros(self.D)
        self.L.pmdi_oracle_cluster_logmarginal(self.h, _ptr(out))
        return out

    def stats(self):
        m = 1 + max(4 * self.D, self.D * max(self.Lmax, 1))
        out = np.zeros(m)
        k = self.L.pmdi_oracle_cluster_stats(self.h, _ptr(out))
        out = out[:k]
        n = int(out[0])
        D = self.D
        if self.kind == GAUSSIAN:
            return {"n": n, "mu": out[1:1 + D], "Sigma": out[1 + D:1 + 2 * D],
                    "lambda": out[1 + 2 * D:1 + 3 * D], "beta": out[1 + 3 * D:1 + 4 * D]}
        if self.kind == CATEGORICAL:
            return {"n": n, "counts": out[1:].reshape(D, self.Lmax).T.astype(np.int64)}
        return {"n": n, "Sigma": out[1:].astype(np.int64)}

    def __del__(self):
        try:
            if self.h:
                self.L.pmdi_oracle_cluster_free(self.h)
                self.h = None
        except Exception:
            pass


def psm_counts(samples, row_lo, row_hi):
    """samples (S, K, n) uint8 -> (K, row_hi-row_lo, n) int32 co-clustering counts (consensus_map.jl:50-56)."""
    smp = np.ascontiguousarray(samples, dtype=np.uint8)
    S, K, n = smp.shape
    out = np.zeros((K, row_hi - row_lo, n), dtype=np.int32)
    lib().pmdi_oracle_psm_counts(_ptr(smp), S, K, n, row_lo, row_hi, _ptr(out))
    return out


def calc_ess(lw):
    lw = np.ascontiguousarray(lw, dtype=np.float64)
    return lib().pmdi_oracle_calc_ess(_ptr(lw), lw.size)


def draw_partstar(lw, u01, uslot):
    lw = np.ascontiguousarray(lw, dtype=np.float64)
    out = np.zeros(lw.size, dtype=np.int64)
    lib().pmdi_oracle_draw_partstar(_ptr(lw), lw.size, float(u01), float(uslot), _ptr(out))
    return out


def phi_upweight(lw, sstar_i, Phi):
    lw = np.array(lw, dtype=np.float64)
    ss = np.asfortranarray(sstar_i, dtype=np.int64)
    P, K = ss.shape
    Phi = np.ascontiguousarray(np.atleast_1d(Phi), dtype=np.float64)
    lib().pmdi_oracle_phi_upweight(_ptr(lw), _ptr(ss), K, _ptr(Phi), P)
    return lw


def philox(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().pmdi_oracle_philox4x32_10(_ptr(c), _ptr(k), _ptr(o))
    return o


def uniform(seed, it, pos, k, p, site):
    return lib().pmdi_oracle_uniform(int(seed), it, pos, k, p, site)


class Hypers:
    """M, gamma, Phi, v, Z, s, order_obs of one chain and the per-iteration host work around the sweep
    (oracle/pmdi_oracle_hypers.c: src/pmdi.jl:59-96,172-185, src/update_hypers.jl, align_labels!), with the
    reference's N^K tables kept literally."""

    def __init__(self, n, N, K, seed=0):
        self.L = lib()
        self.n, self.N, self.K = int(n), int(N), int(K)
        self.npairs = max(1, K * (K - 1) // 2)
        self.h = self.L.pmdi_oracle_hypers_create(self.n, self.N, self.K, int(seed))
        if not self.h:
            raise ValueError("pmdi_oracle_hypers_create rejected the configuration (N^K too large?)")
        self.s = np.ctypeslib.as_array(self.L.pmdi_oracle_hypers_s(self.h), shape=(self.K, self.n)).T      # (n, K) view
        self.order = np.ctypeslib.as_array(self.L.pmdi_oracle_hypers_order(self.h), shape=(self.n,))

    def close(self):
        if self.h:
            self.s = self.order = None
            self.L.pmdi_oracle_hypers_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _get(self, what, size):
        out = np.zeros(size)
        self.L.pmdi_oracle_hypers_get(self.h, what, _ptr(out))
        return out

    def _set(self, what, v):
        a = np.ascontiguousarray(v, dtype=np.float64)
        self.L.pmdi_oracle_hypers_set(self.h, what, _ptr(a))

    M = property(lambda self: self._get(0, self.K), lambda self, v: self._set(0, v))
    # gamma, gamma0: (N, K)
    gamma = property(lambda self: self._get(1, self.N * self.K).reshape(self.K, self.N).T,
                     lambda self, v: self._set(1, np.asarray(v, dtype=np.float64).T))
    Phi = property(lambda self: self._get(2, self.npairs), lambda self, v: self._set(2, v))
    gamma0 = property(lambda self: self._get(4, self.N * self.K).reshape(self.K, self.N).T,
                      lambda self, v: self._set(4, np.asarray(v, dtype=np.float64).T))
    v = property(lambda self: self._get(3, 2)[0], lambda self, x: self._set(3, [x, self._get(3, 2)[1]]))
    Z = property(lambda self: self._get(3, 2)[1], lambda self, x: self._set(3, [self._get(3, 2)[0], x]))

    def update_M(self, it): self.L.pmdi_oracle_hypers_update_M(self.h, int(it))
    def update_gamma(self, it): self.L.pmdi_oracle_hypers_update_gamma(self.h, int(it))
    def update_Phi(self, it): self.L.pmdi_oracle_hypers_update_Phi(self.h, int(it))
    def update_Z(self): return self.L.pmdi_oracle_hypers_update_Z(self.h)
    def update_v(self, it): return self.L.pmdi_oracle_hypers_update_v(self.h, int(it))
    def align_labels(self, it): self.L.pmdi_oracle_hypers_align_labels(self.h, int(it))
    def shuffle(self, it): self.L.pmdi_oracle_hypers_shuffle(self.h, int(it))

    def step(self, it):
        """shuffle!, update_M!, update_gamma!, Pi, update_Phi!, update_Z, update_v (src/pmdi.jl:172-185); returns Pi (N, K)."""
        Pi = np.zeros((self.K, self.N))
        self.L.pmdi_oracle_hypers_step(self.h, int(it), _ptr(Pi))
        return Pi.T.copy()


def normal(seed, it, pos, k, p0, site):
    return lib().pmdi_oracle_normal(int(seed), it, pos, k, p0, site)


def gamma(shape, seed, it, pos, k, site):
    return lib().pmdi_oracle_gamma(float(shape), int(seed), it, pos, k, site)
