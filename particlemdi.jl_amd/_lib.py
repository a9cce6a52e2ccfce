"""ctypes binding of libpmdi_hip.so (include/pmdi_hip.h) and its build recipe.

There is no CPU fallback in this package: if the HIP library is missing or no
gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("PMDI_LIB_PATH") or os.path.join(_PKG, "libpmdi_hip.so")   # override: A/B builds only
_SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("pmdi_sweep.hip", "pmdi_sweep2.hip", "pmdi_kernels.hip", "pmdi_hypers.hip", "pmdi_api.cpp", "pmdi_csv.cpp", "pmdi_comm.cpp")]
_HEADERS = [os.path.join(_PKG, "csrc", "pmdi_internal.h"), os.path.join(_PKG, "csrc", "pmdi_device.h"),
            os.path.join(_PKG, "csrc", "pmdi_sweep2_body.h"), os.path.join(_PKG, "csrc", "pmdi_arith.h"),
            os.path.join(_ROOT, "include", "pmdi_hip.h")]

GAUSSIAN, CATEGORICAL, NEGBINOM = 0, 1, 2
KIND_BY_NAME = {"gaussian": GAUSSIAN, "categorical": CATEGORICAL, "negbinom": NEGBINOM,
                "GaussianCluster": GAUSSIAN, "CategoricalCluster": CATEGORICAL,
                "NegBinomCluster": NEGBINOM}
ABI_VERSION = 2
KMAX = 8

EXPORTS = [
    "pmdi_create", "pmdi_destroy", "pmdi_last_error", "pmdi_abi_version", "pmdi_sweep",
    "pmdi_sweep_device", "pmdi_feature_select", "pmdi_export_state", "pmdi_clusters_new",
    "pmdi_clusters_free", "pmdi_cluster_add", "pmdi_calc_logprob", "pmdi_calc_logmarginal",
    "pmdi_cluster_stats", "pmdi_sum_D", "pmdi_pool_cap", "pmdi_categorical_L", "pmdi_phase_timers",
    "pmdi_block_threads", "pmdi_lds_bytes", "pmdi_chain_costs", "pmdi_label_counts_device", "pmdi_psm_counts_device",
    "pmdi_gibbs_create", "pmdi_gibbs_destroy", "pmdi_gibbs_iterate", "pmdi_gibbs_step", "pmdi_gibbs_iterations",
    "pmdi_gibbs_get", "pmdi_gibbs_set", "pmdi_gibbs_results", "pmdi_gibbs_device_view", "pmdi_gibbs_pack_samples",
    "pmdi_csv_open", "pmdi_csv_write_row", "pmdi_csv_write_gibbs", "pmdi_csv_open_features", "pmdi_csv_write_flags",
    "pmdi_csv_close", "pmdi_csv_read_allocations", "pmdi_format_float64", "pmdi_work_counters", "pmdi_shader_clock_hz", "pmdi_is_split",
    "pmdi_comm_unique_id", "pmdi_comm_init_rank", "pmdi_comm_init_all", "pmdi_comm_destroy", "pmdi_comm_rank", "pmdi_comm_size",
    "pmdi_allgather_samples", "pmdi_settled_kernel", "pmdi_chain_swept_by", "pmdi_tuning_default", "pmdi_tuning_from_env",
]


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): one object per
    source under particlemdi.jl_amd/build/ (rebuilt when the source or a header is newer), then one link."""
    if os.environ.get("PMDI_NO_BUILD") and os.path.exists(LIB_PATH):      # A/B runs of prebuilt variant libraries (PMDI_LIB_PATH)
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("PMDI_EXTRA_HIPCC_FLAGS", "").split()
    bdir = os.path.join(_PKG, "build" if not os.environ.get("PMDI_LIB_PATH") else "build_ab")
    os.makedirs(bdir, exist_ok=True)
    import fcntl
    with open(os.path.join(bdir, ".lock"), "w") as lock:      # ranks of one node may call build() at the same moment
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(force, verbose, hipcc, extra, bdir)


def _build_locked(force, verbose, hipcc, extra, bdir):
    hdr_time = max(os.path.getmtime(p) for p in _HEADERS)
    stamp = os.path.join(bdir, "flags.txt")
    flags_now = " ".join(extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    procs, objs = [], []
    for src in _SOURCES:
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time)):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-x", "hip",
               "-c", src, "-o", obj] + extra
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if procs or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(flags_now)
    return LIB_PATH


class PmdiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class _Dataset(C.Structure):
    _fields_ = [("kind", C.c_int32), ("D", C.c_int32), ("ld", C.c_int64),
                ("xf", C.POINTER(C.c_double)), ("xi", C.POINTER(C.c_int64))]


class _Tuning(C.Structure):
    """pmdi_tuning (include/pmdi_hip.h): kernel-selection knobs, -1 = automatic.  The library never reads the environment by itself;
    this wrapper asks it to (pmdi_tuning_from_env) so that the PMDI_* variables keep working for tests, bench.py and the scripts."""
    _fields_ = [(name, C.c_int32) for name in (
        "settled", "continue_inplace", "sticky", "light_ids", "s2_cols", "s2_idcap", "s2_cls", "ksplit", "requeue_ksplit", "split",
        "heavy_threads", "two_per_cu", "very_heavy", "start_gate", "terms_cap", "lds_target", "phase_timers", "profiled", "ticket")] + [("reserved", C.c_int32 * 5)]


class _Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("K", C.c_int32),
                ("N", C.c_int32), ("P", C.c_int32), ("n_chains", C.c_int32), ("n", C.c_int64),
                ("seed", C.c_uint64), ("q1_mode", C.c_int32), ("q2_mode", C.c_int32),
                ("pool_cap", C.c_int64), ("block_threads", C.c_int32), ("reserved", C.c_int32),
                ("tuning", C.POINTER(_Tuning))]


class SweepStats(C.Structure):
    _fields_ = [("n_operations", C.c_int64), ("n_resamples", C.c_int64), ("n_clones", C.c_int64),
                ("max_id", C.c_int64), ("sum_classes", C.c_int64), ("steps_fast", C.c_int64),
                ("steps_converted", C.c_int64), ("steps_fallback", C.c_int64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_lib = None


def lib():
    """Load libpmdi_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PmdiError(-2, f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    if not os.environ.get("PMDI_NO_TORCH_PRELOAD"):
        # torch bundles its own libamdhip64 (same SONAME): it must be loaded FIRST so that this
        # library binds to the same HIP runtime and device pointers / streams can be shared
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.pmdi_last_error.restype = C.c_char_p
    L.pmdi_abi_version.restype = C.c_int
    L.pmdi_create.restype = C.c_int
    L.pmdi_create.argtypes = [C.POINTER(_Config), C.POINTER(_Dataset), C.POINTER(vp)]
    L.pmdi_destroy.argtypes = [vp]
    L.pmdi_sweep.restype = C.c_int
    L.pmdi_sweep.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, dbl, vp, vp, vp, vp, vp]
    L.pmdi_sweep_device.restype = C.c_int
    L.pmdi_sweep_device.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp]
    L.pmdi_feature_select.restype = C.c_int
    L.pmdi_feature_select.argtypes = [vp, i64, vp, vp, vp]
    L.pmdi_export_state.restype = C.c_int
    L.pmdi_export_state.argtypes = [vp, i32, vp, vp, vp, vp]
    L.pmdi_clusters_new.restype = C.c_int
    L.pmdi_clusters_new.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.pmdi_clusters_free.argtypes = [vp]
    L.pmdi_cluster_add.restype = C.c_int
    L.pmdi_cluster_add.argtypes = [vp, vp, vp]
    L.pmdi_calc_logprob.restype = C.c_int
    L.pmdi_calc_logprob.argtypes = [vp, vp, vp, vp]
    L.pmdi_calc_logmarginal.restype = C.c_int
    L.pmdi_calc_logmarginal.argtypes = [vp, vp]
    L.pmdi_cluster_stats.restype = C.c_int
    L.pmdi_cluster_stats.argtypes = [vp, vp, vp]
    L.pmdi_sum_D.restype = C.c_int
    L.pmdi_sum_D.argtypes = [vp]
    L.pmdi_pool_cap.restype = i64
    L.pmdi_pool_cap.argtypes = [vp]
    L.pmdi_categorical_L.restype = C.c_int
    L.pmdi_categorical_L.argtypes = [vp, i32]
    L.pmdi_block_threads.restype = C.c_int
    L.pmdi_block_threads.argtypes = [vp]
    L.pmdi_lds_bytes.restype = i64
    L.pmdi_lds_bytes.argtypes = [vp]
    L.pmdi_psm_counts_device.restype = C.c_int
    L.pmdi_psm_counts_device.argtypes = [C.c_int32, vp, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, vp, vp]
    L.pmdi_label_counts_device.restype = C.c_int
    L.pmdi_label_counts_device.argtypes = [vp, vp, vp, vp]
    L.pmdi_chain_costs.restype = C.c_int
    L.pmdi_chain_costs.argtypes = [vp, vp]
    L.pmdi_comm_unique_id.restype = C.c_int
    L.pmdi_comm_unique_id.argtypes = [vp]
    L.pmdi_comm_init_rank.restype = C.c_int
    L.pmdi_comm_init_rank.argtypes = [i32, i32, i32, vp, C.POINTER(vp)]
    L.pmdi_comm_init_all.restype = C.c_int
    L.pmdi_comm_init_all.argtypes = [i32, vp, vp]
    L.pmdi_comm_destroy.argtypes = [vp]
    L.pmdi_comm_rank.argtypes = [vp]
    L.pmdi_comm_size.argtypes = [vp]
    L.pmdi_allgather_samples.restype = C.c_int
    L.pmdi_allgather_samples.argtypes = [vp, i32, vp, vp, i64, vp]
    L.pmdi_is_split.argtypes = [vp]
    L.pmdi_settled_kernel.restype = C.c_int
    L.pmdi_settled_kernel.argtypes = [vp, vp]
    L.pmdi_chain_swept_by.restype = C.c_int
    L.pmdi_chain_swept_by.argtypes = [vp, vp]
    L.pmdi_tuning_default.restype = None
    L.pmdi_tuning_default.argtypes = [C.POINTER(_Tuning)]
    L.pmdi_tuning_from_env.restype = None
    L.pmdi_tuning_from_env.argtypes = [C.POINTER(_Tuning)]
    L.pmdi_shader_clock_hz.restype = i64
    L.pmdi_shader_clock_hz.argtypes = [vp]
    L.pmdi_work_counters.restype = C.c_int
    L.pmdi_work_counters.argtypes = [vp, vp]
    L.pmdi_phase_timers.restype = C.c_int
    L.pmdi_phase_timers.argtypes = [vp, i32, vp]
    L.pmdi_gibbs_create.restype = C.c_int
    L.pmdi_gibbs_create.argtypes = [vp, dbl, i32, C.POINTER(vp)]
    L.pmdi_gibbs_destroy.argtypes = [vp]
    L.pmdi_gibbs_iterate.restype = C.c_int
    L.pmdi_gibbs_iterate.argtypes = [vp, i64, vp, vp]
    L.pmdi_gibbs_step.restype = C.c_int
    L.pmdi_gibbs_step.argtypes = [vp, i32, vp]
    L.pmdi_gibbs_iterations.restype = i64
    L.pmdi_gibbs_iterations.argtypes = [vp]
    L.pmdi_gibbs_get.restype = C.c_int
    L.pmdi_gibbs_get.argtypes = [vp, i32] + [vp] * 8
    L.pmdi_gibbs_set.restype = C.c_int
    L.pmdi_gibbs_set.argtypes = [vp, i32] + [vp] * 8
    L.pmdi_gibbs_results.restype = C.c_int
    L.pmdi_gibbs_results.argtypes = [vp, vp, vp, vp, vp]
    L.pmdi_gibbs_device_view.restype = C.c_int
    L.pmdi_gibbs_device_view.argtypes = [vp, vp]
    L.pmdi_gibbs_pack_samples.restype = C.c_int
    L.pmdi_gibbs_pack_samples.argtypes = [vp, vp, vp]
    L.pmdi_csv_open.restype = C.c_int
    L.pmdi_csv_open.argtypes = [C.c_char_p, i32, i64, vp, C.POINTER(vp)]
    L.pmdi_csv_write_row.restype = C.c_int
    L.pmdi_csv_write_row.argtypes = [vp, vp, vp, dbl, vp]
    L.pmdi_csv_write_gibbs.restype = C.c_int
    L.pmdi_csv_write_gibbs.argtypes = [vp, vp, i32, dbl]
    L.pmdi_csv_open_features.restype = C.c_int
    L.pmdi_csv_open_features.argtypes = [C.c_char_p, i32, vp, vp, C.POINTER(vp)]
    L.pmdi_csv_write_flags.restype = C.c_int
    L.pmdi_csv_write_flags.argtypes = [vp, vp]
    L.pmdi_csv_close.argtypes = [vp]
    L.pmdi_csv_read_allocations.restype = C.c_int
    L.pmdi_csv_read_allocations.argtypes = [C.c_char_p, i64, i64, vp, vp, vp, vp, i64, C.c_char_p, i32]
    L.pmdi_format_float64.restype = C.c_int
    L.pmdi_format_float64.argtypes = [dbl, C.c_char_p, i32]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise PmdiError(rc, lib().pmdi_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Sweeper:
    """A batch of `n_chains` independent chains on one MI355X: the device side of
    pmdi()'s per-iteration sweep (src/pmdi.jl:165-171, 188-350, 354-370)."""

    def __init__(self, data, kinds, N, P, n_chains=1, seed=0, device=0, q1_mode=0, q2_mode=0,
                 pool_cap=0, block_threads=0, tuning=None):
        """tuning: dict of pmdi_tuning fields (include/pmdi_hip.h) that override the PMDI_* environment variables for this handle."""
        L = lib()
        self.K = len(data)
        if self.K > KMAX:
            raise PmdiError(-1, f"K={self.K} > {KMAX}")
        self.n = int(data[0].shape[0])
        self.N, self.P, self.C = int(N), int(P), int(n_chains)
        self.D = [int(x.shape[1]) for x in data]
        self.kinds = [KIND_BY_NAME.get(k, k) for k in kinds]
        if len(self.kinds) != self.K:
            raise PmdiError(-1, "Number of datatypes not equal to number of datasets")
        ds = (_Dataset * self.K)()
        self._keep = []
        for k, (x, kind) in enumerate(zip(data, self.kinds)):
            if x.shape[0] != self.n:
                raise PmdiError(-1, "Datasets don't have same number of observations.")
            ds[k].kind, ds[k].D, ds[k].ld = kind, x.shape[1], self.n
            if kind == GAUSSIAN:
                xf = np.asfortranarray(x, dtype=np.float64)
                self._keep.append(xf)
                ds[k].xf = xf.ctypes.data_as(C.POINTER(C.c_double))
            else:
                xi = np.asfortranarray(x, dtype=np.int64)
                self._keep.append(xi)
                ds[k].xi = xi.ctypes.data_as(C.POINTER(C.c_int64))
        self.device = int(device)
        tun = _Tuning()
        L.pmdi_tuning_from_env(C.byref(tun))
        for name, value in (tuning or {}).items():        # explicit knobs win over the environment
            setattr(tun, name, int(value))
        cfg = _Config(ABI_VERSION, device, self.K, self.N, self.P, self.C, self.n, int(seed),
                      q1_mode, q2_mode, int(pool_cap), int(block_threads), 0, C.pointer(tun))
        h = C.c_void_p()
        _check(L.pmdi_create(C.byref(cfg), ds, C.byref(h)))
        self.h = h
        self.sumD = L.pmdi_sum_D(h)
        self.cap = L.pmdi_pool_cap(h)
        self.block_threads = L.pmdi_block_threads(h)
        self.lds_bytes = L.pmdi_lds_bytes(h)
        self.clock_hz = L.pmdi_shader_clock_hz(h)
        self.split = bool(L.pmdi_is_split(h))
        self.settled = bool(L.pmdi_settled_kernel(h, None))
        self.npairs = max(1, self.K * (self.K - 1) // 2)
        self._keep = None  # the library copied the data

    def close(self):
        if getattr(self, "h", None):
            lib().pmdi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sweep(self, it, s, order_obs, n1, Pi, Phi, flags=None, lw_init=None, trace=False):
        """Per chain (leading axis C, dropped when n_chains == 1 inputs are 2-D):
        s (C, n, K) labels 1..N; order_obs (C, n) 1-based; Pi (C, N, K); Phi (C, npairs);
        flags (C, sumD) bytes or None."""
        C_, K, n, N, P = self.C, self.K, self.n, self.N, self.P
        s_a = np.asarray(s, dtype=np.int64).reshape(C_, n, K)
        s_in = np.ascontiguousarray(np.transpose(s_a, (0, 2, 1)))          # column-major n x K per chain
        order = np.ascontiguousarray(np.asarray(order_obs, dtype=np.int64).reshape(C_, n))
        Pi_a = np.asarray(Pi, dtype=np.float64).reshape(C_, N, K)
        Pi_in = np.ascontiguousarray(np.transpose(Pi_a, (0, 2, 1)))
        Phi_in = np.ascontiguousarray(np.asarray(Phi, dtype=np.float64).reshape(C_, self.npairs))
        fl = None
        if flags is not None:
            fl = np.ascontiguousarray(np.asarray(flags, dtype=np.uint8).reshape(C_, self.sumD))
        if lw_init is None:
            lw_init = 0.0 if it == 1 else 1.0
        s_out = np.zeros((C_, K, n), dtype=np.int64)
        lw = np.zeros((C_, P), dtype=np.float64)
        p_star = np.zeros(C_, dtype=np.int64)
        st = (SweepStats * C_)()
        tr = np.zeros((C_, n - n1 + 1, 2 + 2 * K), dtype=np.float64) if trace else None
        _check(lib().pmdi_sweep(self.h, int(it), _ptr(s_in), _ptr(order), int(n1), _ptr(Pi_in),
                                _ptr(Phi_in), _ptr(fl), float(lw_init), _ptr(s_out), _ptr(lw),
                                _ptr(p_star), C.cast(st, C.c_void_p), _ptr(tr)))
        out = {"s": np.transpose(s_out, (0, 2, 1)).copy(), "logweight": lw, "p_star": p_star,
               "stats": [st[c].as_dict() for c in range(C_)]}
        if trace:
            out["trace"] = tr
        return out

    def feature_select(self, it, s_traj):
        C_, K, n = self.C, self.K, self.n
        s_a = np.asarray(s_traj, dtype=np.int64).reshape(C_, n, K)
        s_in = np.ascontiguousarray(np.transpose(s_a, (0, 2, 1)))
        flags = np.zeros((C_, self.sumD), dtype=np.uint8)
        prob = np.zeros((C_, self.sumD), dtype=np.float64)
        _check(lib().pmdi_feature_select(self.h, int(it), _ptr(s_in), _ptr(flags), _ptr(prob)))
        return flags, prob

    def export_state(self, chain=0):
        K, N, P, cap = self.K, self.N, self.P, self.cap
        particle = np.zeros((K, P, N), dtype=np.int64)
        counts = np.zeros((K, cap), dtype=np.int64)
        cn = np.zeros((K, cap), dtype=np.int64)
        mx = np.zeros(K, dtype=np.int64)
        _check(lib().pmdi_export_state(self.h, int(chain), _ptr(particle), _ptr(counts), _ptr(cn), _ptr(mx)))
        return {"particle": particle, "counts": counts, "cluster_n": cn, "max_id": mx}

    def label_counts(self, s):
        """countn for every label (src/update_hypers.jl:72): s is (n_chains, K, n), 1-based labels as in
        the reference; returns (n_chains, K, N) int64 occupancies computed on the device."""
        import torch
        s0 = np.ascontiguousarray(np.asarray(s).reshape(self.C, self.K, self.n) - 1, dtype=np.int32)
        dev = torch.device("cuda", self.device)
        ds = torch.from_numpy(s0).to(dev)
        out = torch.zeros((self.C, self.K, self.N), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev)
        _check(lib().pmdi_label_counts_device(self.h, C.c_void_p(ds.data_ptr()), C.c_void_p(out.data_ptr()),
                                              C.c_void_p(st.cuda_stream)))
        return out.cpu().numpy().astype(np.int64)

    def given_back(self):
        """Chains the settled-chain kernel has handed back to the general kernel so far (include/pmdi_hip.h, pmdi_settled_kernel):
        [0] unused (0), [1] steps with more than 32 particle classes (a subset of [2]), [2] more than 16 particle classes or cluster
        ids beyond 16 bits, [3] total."""
        out = np.zeros(4, dtype=np.int64)
        lib().pmdi_settled_kernel(self.h, _ptr(out))
        return out

    def swept_by(self):
        """Which kernel finished each chain's last sweep: 0 general, 1 settled-chain, 2 general after a hand-back."""
        out = np.zeros(self.C, dtype=np.int32)
        _check(lib().pmdi_chain_swept_by(self.h, _ptr(out)))
        return out

    def chain_costs(self):
        out = np.zeros(self.C, dtype=np.int64)
        _check(lib().pmdi_chain_costs(self.h, _ptr(out)))
        return out

    def work_counters(self):
        """(n_chains, K, 8) int64: evaluated / updated / cloned clusters, moved ids, move events (pmdi_work_counters)."""
        out = np.zeros((self.C, self.K, 8), dtype=np.int64)
        _check(lib().pmdi_work_counters(self.h, _ptr(out)))
        return out

    def phase_timers(self, chain=0):
        out = np.zeros(16, dtype=np.int64)
        _check(lib().pmdi_phase_timers(self.h, int(chain), _ptr(out)))
        return out

    def clusters(self, k, B):
        return ClusterBatch(self, k, B)


class CsvWriter:
    """pmdi()'s output file (src/pmdi.jl:147-158, 377-383) written by the native, byte-compatible writer."""

    def __init__(self, path, K, n, data_names=None, feature_D=None):
        names = None
        if data_names is not None:
            names = (C.c_char_p * K)(*[str(x).encode() for x in data_names])
        h = C.c_void_p()
        if feature_D is None:
            _check(lib().pmdi_csv_open(str(path).encode(), int(K), int(n), names, C.byref(h)))
        else:
            D = np.ascontiguousarray(feature_D, dtype=np.int32)
            _check(lib().pmdi_csv_open_features(str(path).encode(), int(K), _ptr(D), names, C.byref(h)))
        self.h = h

    def row(self, M, Phi, ll, s):
        """s: (n, K) labels 1..N"""
        M = np.ascontiguousarray(M, dtype=np.float64); Phi = np.ascontiguousarray(Phi, dtype=np.float64)
        s_ = np.ascontiguousarray(np.asarray(s, dtype=np.int64).T)
        _check(lib().pmdi_csv_write_row(self.h, _ptr(M), _ptr(Phi), float(ll), _ptr(s_)))

    def gibbs_row(self, gibbs, chain, ll):
        _check(lib().pmdi_csv_write_gibbs(self.h, gibbs.h, int(chain), float(ll)))

    def flags(self, fl):
        f = np.ascontiguousarray(fl, dtype=np.uint8)
        _check(lib().pmdi_csv_write_flags(self.h, _ptr(f)))

    def close(self):
        if getattr(self, "h", None):
            lib().pmdi_csv_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_allocations(path, burnin=0, thin=1):
    """The allocation samples of a pmdi() output file as generate_psm reads them (consensus_map.jl:32-47), through the native
    reader: (samples uint8 (rows kept, K, n_obs), dataset names)."""
    K, n, S = C.c_int32(0), C.c_int64(0), C.c_int64(0)
    names = C.create_string_buffer(1 << 16)
    pth = str(path).encode()
    _check(lib().pmdi_csv_read_allocations(pth, int(burnin), int(thin), C.byref(K), C.byref(n), C.byref(S), None, 0, names, len(names)))
    out = np.zeros((S.value, K.value, n.value), dtype=np.uint8)
    _check(lib().pmdi_csv_read_allocations(pth, int(burnin), int(thin), C.byref(K), C.byref(n), C.byref(S), _ptr(out), out.size, None, 0))
    return out, names.value.decode().split("\n")


def format_float64(x):
    buf = C.create_string_buffer(64)
    k = lib().pmdi_format_float64(float(x), buf, 64)
    if k < 0:
        raise PmdiError(k, lib().pmdi_last_error().decode())
    return buf.value.decode()


class Comm:
    """One rank of the RCCL communicator behind pmdi_allgather_samples (one process per GPU)."""

    def __init__(self, device, n_ranks, rank, unique_id):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        h = C.c_void_p()
        _check(lib().pmdi_comm_init_rank(int(device), int(n_ranks), int(rank), _ptr(uid), C.byref(h)))
        self.h, self.n_ranks, self.rank = h, int(n_ranks), int(rank)

    @staticmethod
    def unique_id():
        uid = np.zeros(128, dtype=np.uint8)
        _check(lib().pmdi_comm_unique_id(_ptr(uid)))
        return uid

    def allgather(self, send_ptr, recv_ptr, bytes_per_rank, stream=None):
        comms = (C.c_void_p * 1)(self.h)
        send = (C.c_void_p * 1)(send_ptr)
        recv = (C.c_void_p * 1)(recv_ptr)
        st = (C.c_void_p * 1)(stream or None)
        _check(lib().pmdi_allgather_samples(comms, 1, send, recv, int(bytes_per_rank), st))

    def close(self):
        if getattr(self, "h", None):
            lib().pmdi_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GibbsView(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in ("M", "gamma", "gamma0", "Phi", "vZ", "Pi", "log1p_phi", "feature_prob", "logweight",
                                          "s", "order_obs", "p_star", "err", "stats", "feature_flag")] + [("n1", C.c_int64)]


STEP_BEGIN, STEP_HYPERS, STEP_SWEEP, STEP_FEATSEL, STEP_ALIGN = range(5)


class Gibbs:
    """Device-resident Gibbs chains of a Sweeper's handle (include/pmdi_hip.h, pmdi_gibbs_*): M, gamma, Phi, v, Z,
    the allocations and order_obs of every chain live on the MI355X; an iteration (src/pmdi.jl:164-384) is a
    handful of kernel launches and nothing crosses PCIe."""

    def __init__(self, sweeper, rho=0.25, feature_select=False):
        self.sw = sweeper
        h = C.c_void_p()
        _check(lib().pmdi_gibbs_create(sweeper.h, float(rho), int(bool(feature_select)), C.byref(h)))
        self.h = h
        self.n1 = int(np.floor(rho * sweeper.n))

    def close(self):
        if getattr(self, "h", None):
            lib().pmdi_gibbs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def iterate(self, n_iter=1, samples_ptr=None, stream=None):
        _check(lib().pmdi_gibbs_iterate(self.h, int(n_iter), C.c_void_p(samples_ptr) if samples_ptr else None,
                                        C.c_void_p(stream) if stream else None))

    def pack_samples(self, out_ptr, stream=None):
        _check(lib().pmdi_gibbs_pack_samples(self.h, C.c_void_p(out_ptr), C.c_void_p(stream) if stream else None))

    def step(self, what, stream=None):
        _check(lib().pmdi_gibbs_step(self.h, int(what), C.c_void_p(stream) if stream else None))

    @property
    def iterations(self):
        return int(lib().pmdi_gibbs_iterations(self.h))

    def get(self, chain=0):
        sw = self.sw
        K, N, n = sw.K, sw.N, sw.n
        M = np.zeros(K); gam = np.zeros((K, N)); gam0 = np.zeros((K, N)); Phi = np.zeros(sw.npairs); vZ = np.zeros(2)
        s = np.zeros((K, n), dtype=np.int64); order = np.zeros(n, dtype=np.int64); fl = np.zeros(sw.sumD, dtype=np.uint8)
        _check(lib().pmdi_gibbs_get(self.h, int(chain), _ptr(M), _ptr(gam), _ptr(gam0), _ptr(Phi), _ptr(vZ), _ptr(s), _ptr(order), _ptr(fl)))
        return {"M": M, "gamma": gam.T.copy(), "gamma0": gam0.T.copy(), "Phi": Phi, "v": vZ[0], "Z": vZ[1],
                "s": s.T.copy(), "order": order, "flags": fl}

    def set(self, chain=0, M=None, gamma=None, gamma0=None, Phi=None, v=None, Z=None, s=None, order=None, flags=None):
        """gamma, gamma0: (N, K); s: (n, K) labels 1..N; order: (n,) 1-based."""
        f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        vZ = None
        if v is not None or Z is not None:
            cur = self.get(chain)
            vZ = np.array([cur["v"] if v is None else v, cur["Z"] if Z is None else Z], dtype=np.float64)
        g = None if gamma is None else np.ascontiguousarray(np.asarray(gamma, dtype=np.float64).T)
        g0 = None if gamma0 is None else np.ascontiguousarray(np.asarray(gamma0, dtype=np.float64).T)
        s_ = None if s is None else np.ascontiguousarray(np.asarray(s, dtype=np.int64).T)
        o_ = None if order is None else np.ascontiguousarray(order, dtype=np.int64)
        fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        _check(lib().pmdi_gibbs_set(self.h, int(chain), _ptr(f64(M)), _ptr(g), _ptr(g0), _ptr(f64(Phi)), _ptr(vZ), _ptr(s_), _ptr(o_), _ptr(fl)))

    def results(self):
        sw = self.sw
        stats = np.zeros((sw.C, 8), dtype=np.int64); err = np.zeros(sw.C, dtype=np.int32)
        ps = np.zeros(sw.C, dtype=np.int64); lw = np.zeros((sw.C, sw.P))
        _check(lib().pmdi_gibbs_results(self.h, _ptr(stats), _ptr(err), _ptr(ps), _ptr(lw)))
        return {"stats": stats, "p_star": ps, "logweight": lw}

    def view(self):
        v = GibbsView()
        _check(lib().pmdi_gibbs_device_view(self.h, C.addressof(v)))
        return v


class ClusterBatch:
    """B stand-alone clusters of dataset k on the device: the calc_logprob /
    cluster_add! / calc_logmarginal protocol of src/datatypes/*.jl."""

    def __init__(self, sweeper, k, B):
        self.sw, self.k, self.B = sweeper, int(k), int(B)
        self.D = sweeper.D[k]
        self.kind = sweeper.kinds[k]
        h = C.c_void_p()
        _check(lib().pmdi_clusters_new(sweeper.h, self.k, self.B, C.byref(h)))
        self.h = h

    def add(self, rows, flag=None):
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(rows, dtype=np.int64), (self.B,)))
        f = None if flag is None else np.ascontiguousarray(flag, dtype=np.uint8)
        _check(lib().pmdi_cluster_add(self.h, _ptr(r), _ptr(f)))

    def logprob(self, rows, flag=None):
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(rows, dtype=np.int64), (self.B,)))
        f = None if flag is None else np.ascontiguousarray(flag, dtype=np.uint8)
        out = np.zeros(self.B)
        _check(lib().pmdi_calc_logprob(self.h, _ptr(r), _ptr(f), _ptr(out)))
        return out

    def logmarginal(self):
        out = np.zeros((self.B, self.D))
        _check(lib().pmdi_calc_logmarginal(self.h, _ptr(out)))
        return out

    def stats(self):
        stride = C.c_int64(0)
        _check(lib().pmdi_cluster_stats(self.h, None, C.addressof(stride)))
        out = np.zeros((self.B, stride.value))
        _check(lib().pmdi_cluster_stats(self.h, _ptr(out), C.addressof(stride)))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().pmdi_clusters_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
