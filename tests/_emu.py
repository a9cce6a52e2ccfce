"""TEST INFRASTRUCTURE: ctypes binding of tests/emu/libpmdi_emu.so -- the device code of the settled-chain kernel
(particlemdi.jl_amd/csrc/pmdi_sweep2_body.h) compiled for the HOST and run in a lock-step workgroup emulator (tests/emu/wavesim.h).
Checks kernel logic against the oracle without a GPU; the product library never sees any of this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_EMU = os.path.join(_HERE, "emu")
_LIB = os.path.join(_EMU, "libpmdi_emu.so")
_CSRC = os.path.join(os.path.dirname(_HERE), "particlemdi.jl_amd", "csrc")


VARIANTS = {"": [], "xcap2": ["-DPM2_XCAP=2"]}      # (xcap2: two LDS entries for uncached clusters, so that the arena list is exercised)


def build(force=False, variant=""):
    _LIB = os.path.join(_EMU, f"libpmdi_emu{'_' + variant if variant else ''}.so")
    deps = [os.path.join(_EMU, "emu_sweep2.cpp"), os.path.join(_EMU, "wavesim.h"), os.path.join(_EMU, "lane_api_emu.h")] + \
           [os.path.join(_CSRC, f) for f in ("pmdi_sweep2_body.h", "pmdi_arith.h", "pmdi_internal.h")]
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(os.path.getmtime(d) for d in deps):
        return _LIB
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + VARIANTS[variant] + [os.path.join(_EMU, "emu_sweep2.cpp"), "-o", _LIB])
    return _LIB


_libs = {}


def lib(variant=""):
    if variant not in _libs:
        L = C.CDLL(build(variant=variant))
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.emu_create.restype = vp
        L.emu_create.argtypes = [i32, i64, i32, i32, vp, vp, C.c_uint64, i32, i32, i32, vp, i32, i32]
        L.emu_destroy.argtypes = [vp]
        L.emu_lds_bytes.restype = i64
        L.emu_lds_bytes.argtypes = [vp]
        L.emu_sweep.restype = C.c_int
        L.emu_sweep.argtypes = [vp, i64, vp, vp, i64, vp, vp, vp, dbl] + [vp] * 10
        _libs[variant] = L
    return _libs[variant]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class EmuSweeper:
    """One chain of the settled-chain kernel on the emulator; same call shape as the oracle's Oracle.sweep."""

    def __init__(self, data, N, P, seed=0, q1_mode=0, cols_l=64, idcap=128, variant="", kinds=None, cls=16, cdfl=0):
        self._L = lib(variant)
        self.K, self.n, self.N, self.P = len(data), int(data[0].shape[0]), int(N), int(P)
        self.D = np.array([x.shape[1] for x in data], dtype=np.int32)
        self._x = [np.ascontiguousarray(x, dtype=np.float64) for x in data]
        ptrs = (C.c_void_p * self.K)(*[x.ctypes.data for x in self._x])
        code = {"gaussian": 0, "categorical": 1, "negbinom": 2}
        self._kinds = np.array([code[k] for k in (kinds or ["gaussian"] * self.K)], dtype=np.int32)
        self.h = self._L.emu_create(self.K, self.n, self.N, self.P, _ptr(self.D), C.cast(ptrs, C.c_void_p), int(seed), int(q1_mode),
                                  int(cols_l), int(idcap), _ptr(self._kinds), int(cls), int(cdfl))
        if not self.h:
            raise ValueError("emu_create rejected the configuration")
        self.lds_bytes = self._L.emu_lds_bytes(self.h)

    def close(self):
        if self.h:
            self._L.emu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sweep(self, it, s, order_obs, n1, Pi, Phi, flags=None, lw_init=None, trace=False):
        K, n, N, P = self.K, self.n, self.N, self.P
        cap = N * P + 1
        s_in = np.ascontiguousarray((np.asarray(s, dtype=np.int64).reshape(n, K) - 1).T, dtype=np.int32)       # [K][n], 0-based
        order = np.ascontiguousarray(np.asarray(order_obs, dtype=np.int64) - 1, dtype=np.int32)
        Pi_in = np.ascontiguousarray(np.asarray(Pi, dtype=np.float64).reshape(N, K).T)                         # [K][N]
        lphi = np.ascontiguousarray(np.log(1.0 + np.atleast_1d(np.asarray(Phi, dtype=np.float64))))             # src/misc.jl:53
        fl = None if flags is None else np.ascontiguousarray(np.concatenate([np.asarray(f, dtype=np.uint8) for f in flags]))
        if lw_init is None:
            lw_init = 0.0 if it == 1 else 1.0
        s_out = np.zeros((K, n), dtype=np.int32); lw = np.zeros(P); pstar = np.zeros(1, dtype=np.int32)
        stats = np.zeros(8, dtype=np.int64); work = np.zeros((K, 8), dtype=np.int64)
        tr = np.zeros((n - n1 + 1, 2 + 2 * K)) if trace else None
        particle = np.zeros((K, P, N), dtype=np.int32); counts = np.zeros((K, cap), dtype=np.int32); cn = np.zeros((K, cap), dtype=np.int32)
        mx = np.zeros(K, dtype=np.int32)
        err = self._L.emu_sweep(self.h, int(it), _ptr(s_in), _ptr(order), int(n1), _ptr(Pi_in), _ptr(lphi), _ptr(fl), float(lw_init),
                              _ptr(s_out), _ptr(lw), _ptr(pstar), _ptr(stats), _ptr(work), _ptr(tr), _ptr(particle), _ptr(counts), _ptr(cn), _ptr(mx))
        out = {"err": err, "why": int(stats[7]), "s": s_out.T.astype(np.int64) + 1, "logweight": lw, "p_star": int(pstar[0]) + 1,
               "stats": dict(zip(("n_operations", "n_resamples", "n_clones", "max_id", "sum_classes"), stats[:5].tolist())),
               "work": work, "state": {"particle": particle.astype(np.int64), "counts": counts.astype(np.int64),
                                       "cluster_n": cn.astype(np.int64), "max_id": mx.astype(np.int64)}}
        if trace:
            out["trace"] = tr
        return out
