// pmdi_arith.h -- the pure arithmetic of the sweep that has to be the same expression, in the same order, wherever it is
// evaluated: the counter-based uniforms and the Gaussian sufficient-statistic recurrences.  Plain C++ (no device builtins),
// shared by the settled-chain kernel (pmdi_sweep2_body.h) and by the host-side emulation of that kernel in tests/emu/ (test
// infrastructure).  Compile with -ffp-contract=off.  Reference lines are file:line relative to /root/reference.
#pragma once
#if defined(__HIPCC__)
#define PMDI_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PMDI_HD inline
#endif

namespace pmdi_arith {

// Philox4x32-10, key = (seed lo, seed hi), ctr = (p, pos, site<<16|k, iter): the specification shared with
// oracle/pmdi_oracle.c and pmdi_device.h's uniform01 (same bits; written with 64-bit products instead of __umulhi).
PMDI_HD double uniform01(unsigned long long seed, unsigned iter, unsigned pos, unsigned k, unsigned p, unsigned site)
{
    unsigned c0 = p, c1 = pos, c2 = (site << 16) | k, c3 = iter;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = (unsigned long long)0xD2511F53u * c0, m1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned hi0 = (unsigned)(m0 >> 32), lo0 = (unsigned)m0, hi1 = (unsigned)(m1 >> 32), lo1 = (unsigned)m1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // 52 random bits -> odd multiple of 2^-53: uniform on the open interval (0,1), never 0
    const unsigned long long m = ((unsigned long long)(c0 >> 6) << 26) | (unsigned long long)(c1 >> 6);
    return (double)(2 * m + 1) * (1.0 / 9007199254740992.0);
}

// cluster_add!(::GaussianCluster), one feature, on the (Sigma, beta) pair the pool stores (gaussian_cluster.jl:54-66):
// mu of the previous size is Sigma / (n - 1 + kappa0), recomputed from exactly the values the reference holds
PMDI_HD void gauss_add_sb(double x, int nnew, double &sg, double &bt)
{
    const double n = (double)nnew;
    const double mu_prev = (nnew == 1) ? 0.0 : sg / ((double)(nnew - 1) + 0.001);
    sg = sg + x;
    const double d = x - mu_prev;
    bt = bt + ((double)(nnew - 1) + 0.001) * (d * d) / (2.0 * (n + 0.001));
}

// mu and lambda of a cluster of size cn from (Sigma, beta): gaussian_cluster.jl:60-63 (cn == 0: the constructor's mu = 0, lambda = 1)
PMDI_HD void gauss_ml(int cn, double sg, double bt, double &mu, double &lam)
{
    if (cn == 0) { mu = 0.0; lam = 1.0; return; }
    const double n = (double)cn;
    mu = sg / (n + 0.001);
    lam = ((0.5 * n + 0.5) * (n + 0.001)) / (bt * (n + 1.001));
}

}  // namespace pmdi_arith
