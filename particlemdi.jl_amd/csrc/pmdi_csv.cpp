// pmdi_csv.cpp -- SURVEY 8 row f4: the output files of pmdi() (src/pmdi.jl:147-158, 377-383 and the
// feature-selection file of :111-116, :380-382), byte-compatible with Julia's writedlm(io, row', ',') so that
// every reader under src/output_analysis/ (consensus_map.jl:32-46, phi_plots.jl:16-25, nclust_plots.jl:17-36,
// feature_select_plots.jl:180-192) keeps working on files written by this library.  Host-only code.
//
// A row is [M; Phi; ll; s[1:(n_obs * K)]]' -- a Float64 vector (vcat promotes the Int allocations), so
// every field prints the way Julia prints a Float64: shortest digits that round-trip, plain notation for
// 1e-5 <= |x| < 1e6, otherwise d.ddde<exp> ("3.0", "0.0001", "1.0e-5", "1.2345678e6").
#include "../../include/pmdi_hip.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct pmdi_csv {
    FILE *f = nullptr;
    int K = 0;
    long long n = 0;
    int npairs = 1;
    std::string buf;
};

int pmdi_set_error(int code, const char *fmt, ...);   // pmdi_api.cpp

namespace {

// Julia Base.show(::Float64) (Ryu shortest): appends to out
void jl_float(std::string &out, double x)
{
    if (std::isnan(x)) { out += "NaN"; return; }
    if (std::isinf(x)) { out += x > 0 ? "Inf" : "-Inf"; return; }
    if (x == 0.0) { out += std::signbit(x) ? "-0.0" : "0.0"; return; }
    char tmp[64];
    auto r = std::to_chars(tmp, tmp + sizeof(tmp), x, std::chars_format::scientific);   // shortest round-trip digits
    *r.ptr = 0;
    const char *p = tmp;
    if (*p == '-') { out += '-'; ++p; }
    char digits[32];
    int nd = 0;
    for (; *p && *p != 'e'; ++p) if (*p != '.') digits[nd++] = *p;
    const int e10 = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    if (-5 < e10 && e10 < 6) {
        if (e10 >= 0) {
            for (int i = 0; i <= e10; ++i) out += (i < nd) ? digits[i] : '0';
            out += '.';
            if (nd > e10 + 1) out.append(digits + e10 + 1, nd - e10 - 1); else out += '0';
        } else {
            out += "0.";
            out.append((size_t)(-e10 - 1), '0');
            out.append(digits, nd);
        }
    } else {
        out += digits[0];
        out += '.';
        if (nd > 1) out.append(digits + 1, nd - 1); else out += '0';
        out += 'e';
        out += std::to_string(e10);
    }
}

void int_as_float(std::string &out, long long v)
{
    if (v > -1000000 && v < 1000000) { out += std::to_string(v); out += ".0"; }
    else jl_float(out, (double)v);
}

}  // namespace

extern "C" {

int pmdi_csv_open(const char *path, int32_t K, int64_t n, const char *const *data_names, pmdi_csv **out)
{
    if (!path || !out || K < 1 || K > PMDI_KMAX || n < 1) return pmdi_set_error(PMDI_E_ARG, "pmdi_csv_open: bad argument");
    *out = nullptr;
    pmdi_csv *w = new (std::nothrow) pmdi_csv();
    if (!w) return pmdi_set_error(PMDI_E_MEMORY, "out of host memory");
    w->f = fopen(path, "w");
    if (!w->f) { delete w; return pmdi_set_error(PMDI_E_ARG, "cannot open %s", path); }
    w->K = K; w->n = n; w->npairs = K > 1 ? K * (K - 1) / 2 : 1;
    std::string &b = w->buf;
    for (int k = 1; k <= K; ++k) { b += "MassParameter_"; b += std::to_string(k); b += ','; }          // :147
    if (K > 1) {
        for (int a = 1; a <= K - 1; ++a)
            for (int c = a + 1; c <= K; ++c) { b += "phi_"; b += std::to_string(a); b += '_'; b += std::to_string(c); b += ','; }
    } else {
        b += "phi_1_1,";                                                                               // calculate_Phi_lab(1) = [1 1]
    }
    b += "ll";
    for (int k = 0; k < K; ++k) {
        std::string name = data_names && data_names[k] ? data_names[k] : ("K" + std::to_string(k + 1));   // :46-48
        for (long long i = 1; i <= n; ++i) { b += ','; b += name; b += "_n"; b += std::to_string(i); }
    }
    b += '\n';
    fwrite(b.data(), 1, b.size(), w->f);
    *out = w;
    return PMDI_OK;
}

int pmdi_csv_write_row(pmdi_csv *w, const double *M, const double *Phi, double ll, const int64_t *s)
{
    if (!w || !w->f || !M || !Phi || !s) return pmdi_set_error(PMDI_E_ARG, "pmdi_csv_write_row: null argument");
    std::string &b = w->buf;
    b.clear();
    for (int k = 0; k < w->K; ++k) { jl_float(b, M[k]); b += ','; }
    for (int i = 0; i < w->npairs; ++i) { jl_float(b, Phi[i]); b += ','; }
    jl_float(b, ll);
    const long long tot = w->n * w->K;
    for (long long i = 0; i < tot; ++i) { b += ','; int_as_float(b, s[i]); }                          // s[1:(n_obs * K)]: column-major
    b += '\n';
    if (fwrite(b.data(), 1, b.size(), w->f) != b.size()) return pmdi_set_error(PMDI_E_STATE, "short write");
    return PMDI_OK;
}

int pmdi_csv_write_gibbs(pmdi_csv *w, pmdi_gibbs *g, int32_t chain, double ll)
{
    if (!w || !g) return pmdi_set_error(PMDI_E_ARG, "pmdi_csv_write_gibbs: null argument");
    std::vector<double> M((size_t)w->K), Phi((size_t)w->npairs);
    std::vector<int64_t> s((size_t)w->n * w->K);
    const int rc = pmdi_gibbs_get(g, chain, M.data(), nullptr, nullptr, Phi.data(), nullptr, s.data(), nullptr, nullptr);
    if (rc) return rc;
    return pmdi_csv_write_row(w, M.data(), Phi.data(), ll, s.data());
}

int pmdi_csv_open_features(const char *path, int32_t K, const int32_t *D, const char *const *data_names, pmdi_csv **out)
{
    if (!path || !out || !D || K < 1 || K > PMDI_KMAX) return pmdi_set_error(PMDI_E_ARG, "pmdi_csv_open_features: bad argument");
    *out = nullptr;
    pmdi_csv *w = new (std::nothrow) pmdi_csv();
    if (!w) return pmdi_set_error(PMDI_E_MEMORY, "out of host memory");
    w->f = fopen(path, "w");
    if (!w->f) { delete w; return pmdi_set_error(PMDI_E_ARG, "cannot open %s", path); }
    w->K = K;
    std::string &b = w->buf;
    bool first = true;
    for (int k = 0; k < K; ++k) {
        std::string name = data_names && data_names[k] ? data_names[k] : ("K" + std::to_string(k + 1));
        for (int d = 1; d <= D[k]; ++d) {                                                              // :111
            if (!first) b += ',';
            first = false;
            b += name; b += "_d"; b += std::to_string(d);
        }
        w->n += D[k];
    }
    b += '\n';
    fwrite(b.data(), 1, b.size(), w->f);
    *out = w;
    return PMDI_OK;
}

int pmdi_csv_write_flags(pmdi_csv *w, const uint8_t *flags)
{
    if (!w || !w->f || !flags) return pmdi_set_error(PMDI_E_ARG, "pmdi_csv_write_flags: null argument");
    std::string &b = w->buf;
    b.clear();
    for (long long i = 0; i < w->n; ++i) { if (i) b += ','; b += flags[i] ? "true" : "false"; }      // Bool prints as true/false
    b += '\n';
    if (fwrite(b.data(), 1, b.size(), w->f) != b.size()) return pmdi_set_error(PMDI_E_STATE, "short write");
    return PMDI_OK;
}

int pmdi_csv_close(pmdi_csv *w)
{
    if (!w) return PMDI_OK;
    if (w->f) fclose(w->f);
    delete w;
    return PMDI_OK;
}

int pmdi_format_float64(double x, char *out, int32_t cap)
{
    std::string b;
    jl_float(b, x);
    if (!out || cap <= (int32_t)b.size()) return pmdi_set_error(PMDI_E_ARG, "buffer too small");
    memcpy(out, b.c_str(), b.size() + 1);
    return (int)b.size();
}

}  // extern "C"
