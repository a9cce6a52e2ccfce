// pmdi_comm.cpp -- the ONE exchange step of the multi-GPU path: chains are independent, so the
// only collective is an all-gather of the retained allocation samples (uint8 labels, [sample][chain][K][n] per
// rank) over RCCL/xGMI, after which every GPU holds the pooled samples and builds its row block of the
// posterior-similarity matrix (consumer: generate_psm, src/output_analysis/consensus_map.jl:31-65) with no further
// exchange.  RCCL is bound at first use with dlopen (the process may already have torch's librccl.so mapped: the
// SONAME resolves to that same copy), so the library loads on machines without it.
#include "../../include/pmdi_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <vector>

int pmdi_set_error(int code, const char *fmt, ...);   // pmdi_api.cpp

struct pmdi_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, n_ranks = 1;
};

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *nm : names) {          // a copy the process has already mapped (e.g. the one PyTorch ships) is the one to share
        lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (lib) break;
    }
    for (int i = 0; i < 3 && !lib; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return pmdi_set_error(PMDI_E_DEVICE, "RCCL is not available: %s", dlerror());
    Rccl r;
    r.lib = lib;
#define SYM(field, name)                                                                      \
    *(void **)(&r.field) = dlsym(lib, name);                                                  \
    if (!r.field) return pmdi_set_error(PMDI_E_DEVICE, "RCCL symbol %s missing", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = r;
    return 0;
}

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r__ = (expr);                                                              \
        if (r__ != ncclSuccess) return pmdi_set_error(PMDI_E_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(r__)); \
    } while (0)

}  // namespace

extern "C" {

int pmdi_comm_unique_id(uint8_t id[PMDI_COMM_ID_BYTES])
{
    if (!id) return pmdi_set_error(PMDI_E_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) <= PMDI_COMM_ID_BYTES, "ncclUniqueId does not fit");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memset(id, 0, PMDI_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return PMDI_OK;
}

int pmdi_comm_init_rank(int32_t device, int32_t n_ranks, int32_t rank, const uint8_t id[PMDI_COMM_ID_BYTES], pmdi_comm **out)
{
    if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return pmdi_set_error(PMDI_E_ARG, "pmdi_comm_init_rank: bad argument");
    *out = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    if (hipSetDevice(device) != hipSuccess) return pmdi_set_error(PMDI_E_DEVICE, "hipSetDevice(%d) failed", device);
    pmdi_comm *c = new (std::nothrow) pmdi_comm();
    if (!c) return pmdi_set_error(PMDI_E_MEMORY, "out of host memory");
    c->device = device; c->rank = rank; c->n_ranks = n_ranks;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) { delete c; return pmdi_set_error(PMDI_E_DEVICE, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    *out = c;
    return PMDI_OK;
}

int pmdi_comm_init_all(int32_t n_devices, const int32_t *devices, pmdi_comm **out)
{
    if (!out || n_devices < 1) return pmdi_set_error(PMDI_E_ARG, "pmdi_comm_init_all: bad argument");
    for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    std::vector<ncclComm_t> comms((size_t)n_devices);
    std::vector<int> devs((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) devs[i] = devices ? devices[i] : i;
    NCCL_TRY(g_rccl.CommInitAll(comms.data(), n_devices, devs.data()));
    for (int i = 0; i < n_devices; ++i) {
        pmdi_comm *c = new (std::nothrow) pmdi_comm();
        if (!c) {
            // nothing half-made is left behind: the wrappers built so far (they own their communicators) and the
            // communicators that have no wrapper yet
            for (int j = 0; j < i; ++j) { (void)pmdi_comm_destroy(out[j]); out[j] = nullptr; }
            for (int j = i; j < n_devices; ++j)
                if (comms[j] && g_rccl.CommDestroy) { (void)hipSetDevice(devs[j]); (void)g_rccl.CommDestroy(comms[j]); }
            return pmdi_set_error(PMDI_E_MEMORY, "out of host memory");
        }
        c->comm = comms[i]; c->device = devs[i]; c->rank = i; c->n_ranks = n_devices;
        out[i] = c;
    }
    return PMDI_OK;
}

int pmdi_comm_destroy(pmdi_comm *c)
{
    if (!c) return PMDI_OK;
    if (c->comm && g_rccl.CommDestroy) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->comm); }
    delete c;
    return PMDI_OK;
}

int pmdi_allgather_samples(pmdi_comm *const *comms, int32_t n_local, const uint8_t *const *send, uint8_t *const *recv,
                           int64_t bytes_per_rank, void *const *streams)
{
    if (!comms || !send || !recv || n_local < 1 || bytes_per_rank < 0) return pmdi_set_error(PMDI_E_ARG, "pmdi_allgather_samples: bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    for (int i = 0; i < n_local; ++i)
        if (!comms[i] || !send[i] || !recv[i]) return pmdi_set_error(PMDI_E_ARG, "pmdi_allgather_samples: null entry %d", i);
    if (n_local > 1) NCCL_TRY(g_rccl.GroupStart());
    for (int i = 0; i < n_local; ++i) {
        if (hipSetDevice(comms[i]->device) != hipSuccess) return pmdi_set_error(PMDI_E_DEVICE, "hipSetDevice(%d) failed", comms[i]->device);
        hipStream_t st = streams ? (hipStream_t)streams[i] : nullptr;
        ncclResult_t r = g_rccl.AllGather(send[i], recv[i], (size_t)bytes_per_rank, ncclUint8, comms[i]->comm, st);
        if (r != ncclSuccess) {
            if (n_local > 1) (void)g_rccl.GroupEnd();
            return pmdi_set_error(PMDI_E_DEVICE, "ncclAllGather: %s", g_rccl.GetErrorString(r));
        }
    }
    if (n_local > 1) NCCL_TRY(g_rccl.GroupEnd());
    return PMDI_OK;
}

int pmdi_comm_rank(const pmdi_comm *c) { return c ? c->rank : -1; }
int pmdi_comm_size(const pmdi_comm *c) { return c ? c->n_ranks : 0; }

}  // extern "C"
