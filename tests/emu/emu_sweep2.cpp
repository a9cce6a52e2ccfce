// emu_sweep2.cpp -- TEST INFRASTRUCTURE ONLY.  Runs the device code of the settled-chain kernel
// (particlemdi.jl_amd/csrc/pmdi_sweep2_body.h) on the host, one emulated workgroup (tests/emu/wavesim.h), so that its logic can be
// checked against the oracle without a GPU.  Not linked into, loaded by or shipped with the product library.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lane_api_emu.h"
#include "../../particlemdi.jl_amd/csrc/pmdi_sweep2_body.h"

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the arena of pmdi_api.cpp's layout_arena (sweep state), restated for host memory
size_t layout_arena(DsetDev &d, int N, int P, long long cap, long long n_rows_sstar)
{
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    const size_t ids = (size_t)cap + 1;
    d.o_particle[0] = take((size_t)N * P * 4);
    d.o_particle[1] = take((size_t)N * P * 4);
    d.o_col = take((size_t)P * 4);
    d.o_cgrp = take((size_t)(N > 3 ? N : 3) * P * 8);
    d.o_pid = take((size_t)P * 4);
    d.o_sid = take((size_t)P * 4);
    d.o_kv = take((size_t)P * 4);
    d.o_newid = take((size_t)N * P * 4);
    d.o_counts = take(ids * 4);
    d.o_ncop = take(ids * 4);
    d.o_firstc = take(ids * 4);
    d.o_lp = take(ids * 8);
    d.o_sstar = take((size_t)n_rows_sstar * P);
    d.o_clslead = take((size_t)P * 4);
    d.o_clsval = take((size_t)P * 4);
    d.o_cdf = take((size_t)P * (N + 2) * 8);
    d.o_dl = take((size_t)3 * P * 4);
    d.o_s2x = take((size_t)2048 * 12);
    d.o_cn = take(ids * 4);
    if (d.kind == K_GAUSSIAN) d.o_sb = take(ids * d.D * 16);
    else if (d.kind == K_CATEGORICAL) d.o_cnt = take(ids * d.D * d.L * 4);
    else d.o_nbs = take(ids * d.D * 8);
    return o;
}

struct Emu {
    int K, N, P;
    long long n, cap;
    unsigned long long seed;
    int q1;
    std::vector<std::vector<double>> x, gtab, lhtab, lgtab;
    std::vector<std::vector<int>> xi, maxcol;
    std::vector<std::vector<char>> arena;
    DsetDev ds[PMDI_KMAX_I];
    int Dmax = 0, sumD = 0;
    int cols_l, idcap, cls, cdfl;
};

struct RunArg { const SweepArgs *a; int K, PPL, NW; bool gauss_only; };

template <int K, int PPL, int NW> void body(const SweepArgs *a, bool gauss_only)
{
    if (gauss_only && NW == 4) {         // the all-Gaussian build of the 4-wave shapes (as pmdi_sweep2.hip picks it)
        pmdi_s2::Sweep2<K, PPL, NW, true> s;
        s.run(a, 0);
    } else {
        pmdi_s2::Sweep2<K, PPL, NW, false> s;
        s.run(a, 0);
    }
}

// the instantiations of pmdi_sweep2.hip (four waves: 1, 2, 4 particles per lane; eight waves: 4 per lane)
void entry(void *p)
{
    const RunArg *r = (const RunArg *)p;
#define CASE(K_, P_, W_) if (r->K == K_ && r->PPL == P_ && r->NW == W_) return body<K_, P_, W_>(r->a, r->gauss_only)
    CASE(1, 1, 4); CASE(1, 2, 4); CASE(1, 4, 4); CASE(2, 1, 4); CASE(2, 2, 4); CASE(2, 4, 4); CASE(3, 1, 4); CASE(3, 2, 4); CASE(3, 4, 4);
    CASE(4, 1, 4); CASE(4, 2, 4); CASE(4, 4, 4);
    CASE(1, 4, 8); CASE(2, 4, 8); CASE(3, 4, 8); CASE(4, 4, 8);
#undef CASE
    fprintf(stderr, "emu: unsupported K=%d PPL=%d NW=%d\n", r->K, r->PPL, r->NW);
    abort();
}

}  // namespace

extern "C" {

// data[k]: n x D_k row-major doubles (integer types: levels 1..L / counts >= 0 as doubles); kinds[k]: K_GAUSSIAN / K_CATEGORICAL / K_NEGBINOM
void *emu_create(int K, long long n, int N, int P, const int *D, const double *const *data, unsigned long long seed, int q1,
                 int cols_l, int idcap, const int *kinds, int cls, int cdfl)
{
    if (K < 1 || K > pmdi_s2::KMAX2 || (P != 256 && P != 512 && P != 1024 && P != 2048) || N > 64) return nullptr;
    Emu *e = new Emu();
    e->K = K; e->N = N; e->P = P; e->n = n; e->cap = (long long)N * P + 1; e->seed = seed; e->q1 = q1;
    {
        const int nw = P > 1024 ? 8 : 4;
        const int mx = pmdi_s2::class_slots_max(K, P / (64 * nw), nw);
        if (cls > mx) cls = mx;
    }
    e->cols_l = cols_l; e->idcap = idcap; e->cls = cls;
    e->cdfl = (cdfl > 0 && cdfl < cls && pmdi_s2::cdf_rows_in_arena(P > 1024 ? 8 : 4)) ? cdfl : cls;     // (only the 8-wave builds carry that path)
    memset(e->ds, 0, sizeof(e->ds));
    e->x.resize(K); e->gtab.resize(K); e->arena.resize(K); e->lhtab.resize(K); e->lgtab.resize(K); e->xi.resize(K); e->maxcol.resize(K);
    int flag_off = 0;
    for (int k = 0; k < K; ++k) {
        if (D[k] > 64) { delete e; return nullptr; }
        DsetDev &d = e->ds[k];
        d.kind = kinds ? kinds[k] : K_GAUSSIAN; d.D = D[k]; d.L = 0; d.flag_off = flag_off;
        flag_off += D[k];
        if (D[k] > e->Dmax) e->Dmax = D[k];
        if (d.kind == K_GAUSSIAN) {
            e->x[k].assign(data[k], data[k] + (size_t)n * D[k]);
            d.xf = e->x[k].data();
            e->gtab[k].resize((size_t)n + 1);
            for (long long m = 0; m <= n; ++m) {
                const double nn = (double)m;
                e->gtab[k][m] = (log(1.0 / sqrt(M_PI)) + lgamma(0.5 * nn + 1.0)) - lgamma(0.5 * nn + 0.5);   // as pmdi_create
            }
            d.gtab = e->gtab[k].data();
        } else {
            // the host-built tables of pmdi_create (pmdi_api.cpp), restated
            e->xi[k].resize((size_t)n * D[k]);
            e->maxcol[k].assign(D[k], 0);
            std::vector<long long> colsum(D[k], 0);
            long long gmax = 0;
            for (long long i = 0; i < n; ++i)
                for (int q = 0; q < D[k]; ++q) {
                    const long long v = (long long)data[k][(size_t)i * D[k] + q];
                    e->xi[k][(size_t)i * D[k] + q] = (int)v;
                    if (v > e->maxcol[k][q]) e->maxcol[k][q] = (int)v;
                    if (v > gmax) gmax = v;
                    colsum[q] += v;
                }
            d.xi = e->xi[k].data();
            if (d.kind == K_CATEGORICAL) {
                d.L = (int)gmax;
                d.maxcol = e->maxcol[k].data();
                e->lhtab[k].resize((size_t)(2 * n + gmax + 3));
                for (size_t j = 0; j < e->lhtab[k].size(); ++j) e->lhtab[k][j] = log(0.5 * (double)j);
                d.lhtab = e->lhtab[k].data();
            } else {
                long long smax = 0;
                for (int q = 0; q < D[k]; ++q) if (colsum[q] > smax) smax = colsum[q];
                const long long len = n + gmax + smax + 8;
                e->lgtab[k].resize((size_t)len);
                for (long long m = 0; m < len; ++m) e->lgtab[k][m] = lgamma((double)m);
                d.lgtab = e->lgtab[k].data();
                d.lgtab_len = len;
            }
        }
        d.stride = layout_arena(d, N, P, e->cap, n);
        e->arena[k].assign(d.stride, 0);
        d.arena = e->arena[k].data();
    }
    e->sumD = flag_off;
    return e;
}

void emu_destroy(void *h) { delete (Emu *)h; }

long long emu_lds_bytes(void *h)
{
    Emu *e = (Emu *)h;
    S2Layout L;
    pmdi_s2::make_layout(e->K, e->N, e->P, e->Dmax, e->cols_l, e->idcap, e->cls, e->cdfl, L);
    return L.total;
}

// one sweep of one chain; 0-based labels / order; returns err (0 ok, 1 requeue, -4 pool)
int emu_sweep(void *h, long long iter, const int *s_in, const int *order, long long n1, const double *Pi, const double *logphi,
              const unsigned char *flags, double lw_init, int *s_out, double *lw_out, int *pstar, long long *stats, long long *work,
              double *trace, int *particle, int *counts, int *cn, int *maxid_out)
{
    Emu *e = (Emu *)h;
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.K = e->K; a.N = e->N; a.P = e->P; a.cap = (int)e->cap;
    a.Dmax = e->Dmax; a.sumD = e->sumD; a.npairs = e->K > 1 ? e->K * (e->K - 1) / 2 : 1;
    a.q1 = e->q1; a.q2 = 0; a.trace_on = trace ? 1 : 0;
    a.iter = (unsigned)iter; a.n = e->n; a.n1 = n1; a.seed = e->seed; a.lw_init = lw_init;
    for (int k = 0; k < e->K; ++k) a.ds[k] = e->ds[k];
    a.s_in = s_in; a.order = order; a.Pi = Pi; a.logphi = logphi; a.flags = flags;
    a.s_out = s_out; a.lw_out = lw_out; a.pstar = pstar; a.stats = stats; a.trace = trace;
    int err = 0;
    long long cost = 0;
    std::vector<int> kstate(PMDI_KMAX_I * 2, 0);
    std::vector<long long> wk(PMDI_KMAX_I * 8, 0);
    a.err = &err; a.cost = &cost; a.kstate = kstate.data(); a.work = wk.data();
    pmdi_s2::make_layout(e->K, e->N, e->P, e->Dmax, e->cols_l, e->idcap, e->cls, e->cdfl, a.s2);
    const int nw = e->P > 1024 ? 8 : 4;
    bool go = true;
    for (int k = 0; k < e->K; ++k) go = go && e->ds[k].kind == K_GAUSSIAN;
    RunArg r{&a, e->K, e->P / (64 * nw), nw, go};
    wavesim::run_block(64 * nw, 0, (size_t)a.s2.total, entry, &r);
    if (work) for (int k = 0; k < e->K; ++k) for (int j = 0; j < 8; ++j) work[k * 8 + j] = wk[k * 8 + j];
    if (err == 0 && particle) {
        // what pmdi_export_state does: expand the columns back to particle[n, p, k]
        for (int k = 0; k < e->K; ++k) {
            const DsetDev &d = e->ds[k];
            const int *tab = (const int *)(d.arena + d.o_particle[0]);
            const int *col = (const int *)(d.arena + d.o_col);
            for (int p = 0; p < e->P; ++p)
                for (int nn = 0; nn < e->N; ++nn) particle[((size_t)k * e->P + p) * e->N + nn] = tab[(size_t)col[p] * e->N + nn];
            const int *cc = (const int *)(d.arena + d.o_counts), *cnn = (const int *)(d.arena + d.o_cn);
            for (long long id = 1; id <= e->cap; ++id) { counts[(size_t)k * e->cap + (id - 1)] = cc[id]; cn[(size_t)k * e->cap + (id - 1)] = cnn[id]; }
            maxid_out[k] = kstate[k * 2];
        }
    }
    return err;
}

}  // extern "C"
