"""Batched Gibbs driver for many independent chains on one GPU (bench.py, multi-chain runs).

For K = 1 the host work per iteration is tiny (update_M!, update_gamma!, Pi; align_labels! is a
no-op, update_Phi! does not run: src/pmdi.jl:176-185, src/misc.jl:62), so the allocations never
leave the device: the sweep's s_out feeds the next sweep's s_in, label counts are reduced on the
device, and only C x N numbers cross PCIe per iteration.  torch is used for device buffers,
the per-iteration shuffle (src/pmdi.jl:172) and stream/event plumbing only.
"""
import ctypes as C

import numpy as np
from scipy.stats import gamma as gamma_dist

from ._lib import PmdiError, Sweeper, _check, lib

EPS = np.finfo(np.float64).eps


class BatchedHypersK1:
    """update_M!, update_gamma!, update_Z, update_v (src/update_hypers.jl) for C chains with one
    dataset each, vectorised over chains.  Keeps the reference's stale Gamma_c:
    norm_temp is rebuilt from the INITIAL gamma at every call."""

    def __init__(self, n_obs, N, C_, rng):
        self.n, self.N, self.C, self.rng = n_obs, N, C_, rng
        self.M = np.full(C_, 2.0)
        self.gamma = rng.gamma(1.0 / N, 1.0, size=(C_, N)) + EPS
        self.gamma0 = self.gamma.copy()                    # exp(Gamma_c), never refreshed
        self.Z = self.gamma0.sum(axis=1)
        self.v = rng.gamma(self.n, 1.0 / self.Z)

    def initial_s(self):
        p = self.gamma / self.gamma.sum(axis=1, keepdims=True)
        u = self.rng.random((self.C, self.n))
        return (u[:, :, None] > np.cumsum(p, axis=1)[:, None, :]).sum(axis=2).clip(0, self.N - 1)

    def step(self, counts):
        """counts: (C, N) occupancy of each label.  Returns Pi (C, N)."""
        N, rng = self.N, self.rng
        # update_M! (update_hypers.jl:5-26)
        cur = self.M
        ll = -gamma_dist.logpdf(self.gamma, (cur / N)[:, None], scale=1.0).sum(axis=1)
        ll0 = -gamma_dist.logpdf(cur, 2.0, scale=0.25)
        prop = cur + rng.normal(size=self.C) / 10.0
        ok = prop > 0.0
        safe = np.where(ok, prop, 1.0)
        nll = -gamma_dist.logpdf(self.gamma, (safe / N)[:, None], scale=1.0).sum(axis=1)
        nll0 = -gamma_dist.logpdf(safe, 2.0, scale=0.25)
        with np.errstate(over="ignore"):
            alpha = np.where(ok, np.exp(-nll - nll0 + ll + ll0), 0.0)
        acc = rng.random(self.C) < alpha
        self.M = np.where(acc, prop, cur)
        # update_gamma! (update_hypers.jl:64-92), K = 1: pertinent row of (k=1, n) is row n
        alpha_star = (self.M / N)[:, None] + counts
        nt = self.gamma0.copy()
        for nn in range(N):
            old = self.gamma[:, nn].copy()
            beta_star = 1.0 + self.v * nt[:, nn] / old
            self.gamma[:, nn] = rng.gamma(alpha_star[:, nn], 1.0 / beta_star) + EPS
            nt[:, nn] *= self.gamma[:, nn] / old
        Pi = self.gamma / self.gamma.sum(axis=1, keepdims=True)
        # update_Z, update_v (update_hypers.jl:29-39, 1-3)
        self.Z = self.gamma0.sum(axis=1)
        self.v = rng.gamma(self.n, 1.0 / self.Z)
        return Pi


class DeviceGibbsK1:
    """C chains of a single-dataset model, allocations resident on the device."""

    def __init__(self, data, kind, N, P, n_chains, seed=0, device=0, block_threads=0, pool_cap=0, rho=0.25):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.sw = Sweeper([data], [kind], N, P, n_chains=n_chains, seed=seed, device=device,
                          block_threads=block_threads, pool_cap=pool_cap)
        self.n, self.N, self.P, self.C = self.sw.n, N, P, n_chains
        self.n1 = int(np.floor(rho * self.n))
        self.rng = np.random.default_rng(seed)
        self.hy = BatchedHypersK1(self.n, N, n_chains, self.rng)
        s0 = self.hy.initial_s().astype(np.int32)                      # (C, n), 0-based
        self.s = torch.from_numpy(s0).to(self.dev).contiguous()        # [chain][K=1][n]
        self.s_next = torch.empty_like(self.s)
        self.order = torch.empty((n_chains, self.n), dtype=torch.int32, device=self.dev)
        self.Pi = torch.empty((n_chains, N), dtype=torch.float64, device=self.dev)
        self.logphi = torch.zeros((n_chains, 1), dtype=torch.float64, device=self.dev)
        self.lw = torch.empty((n_chains, P), dtype=torch.float64, device=self.dev)
        self.pstar = torch.empty(n_chains, dtype=torch.int32, device=self.dev)
        self.stats = torch.zeros((n_chains, 8), dtype=torch.int64, device=self.dev)
        self.err = torch.zeros(n_chains, dtype=torch.int32, device=self.dev)
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(seed + 12345)
        self.it = 0
        self.kernel_ms = []
        self.events = []          # (start, end) HIP events of every timed sweep, for interval arithmetic
        self.counts = torch.zeros((n_chains, N), dtype=torch.int32, device=self.dev)

    def iteration(self, time_kernel=False):
        torch = self.torch
        self.it += 1
        # shuffle!(order_obs) per chain (src/pmdi.jl:172), on the device
        self.order.copy_(torch.argsort(torch.rand((self.C, self.n), device=self.dev, generator=self.gen), dim=1))
        # label occupancy for update_gamma! (countn, update_hypers.jl:73)
        stream = torch.cuda.current_stream(self.dev)
        _check(lib().pmdi_label_counts_device(self.sw.h, C.c_void_p(self.s.data_ptr()), C.c_void_p(self.counts.data_ptr()),
                                              C.c_void_p(stream.cuda_stream)))
        Pi = self.hy.step(self.counts.cpu().numpy().astype(np.float64))
        self.Pi.copy_(torch.from_numpy(np.ascontiguousarray(Pi)))
        if time_kernel:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        _check(lib().pmdi_sweep_device(
            self.sw.h, self.it, C.c_void_p(self.s.data_ptr()), C.c_void_p(self.order.data_ptr()), self.n1,
            C.c_void_p(self.Pi.data_ptr()), C.c_void_p(self.logphi.data_ptr()), None,
            0.0 if self.it == 1 else 1.0, C.c_void_p(self.s_next.data_ptr()), C.c_void_p(self.lw.data_ptr()),
            C.c_void_p(self.pstar.data_ptr()), C.c_void_p(self.stats.data_ptr()), C.c_void_p(self.err.data_ptr()),
            C.c_void_p(stream.cuda_stream)))
        if time_kernel:
            e1.record(stream)
            self._pending = (e0, e1)
            self.events.append((e0, e1))
        self.s, self.s_next = self.s_next, self.s      # s = sstar[p_star,:,:] (:373); align_labels! is a no-op for K=1

    def finish_timing(self):
        e0, e1 = self._pending
        e1.synchronize()
        self.kernel_ms.append(e0.elapsed_time(e1))

    def check(self):
        self.torch.cuda.synchronize(self.dev)
        err = self.err.cpu().numpy()
        if (err != 0).any():
            raise PmdiError(int(err[err != 0][0]), "sweep kernel reported an error (pool capacity?)")
        return self.stats.cpu().numpy()
