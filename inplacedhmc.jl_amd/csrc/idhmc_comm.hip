// idhmc_comm.hip -- the one collective of the path: SUM all-reduce of the fixed-point record {limb sums, chain count, error count}
// for the global dual-averaging stepsize (north_star).  RCCL is bound at run time (dlopen of
// librccl.so.1 -- the copy a host process such as PyTorch has already loaded is reused by SONAME), so the
// library has no link-time dependency on it and single-GPU users never load it.  The all-reduce is enqueued on
// the context's stream: ordered after k_accept_sum and before k_da_adapt_global, no host synchronisation.
#include "idhmc_internal.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace idhmc {

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl(char *err, size_t cap)
{
    if (g_rccl.handle) return 0;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { snprintf(err, cap, "cannot load RCCL: %s", dlerror()); return 1; }
    RcclApi a;
    a.handle = h;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(h, "ncclAllReduce"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!a.get_unique_id || !a.comm_init_rank || !a.all_reduce || !a.comm_destroy || !a.error_string) {
        snprintf(err, cap, "RCCL library lacks an expected symbol");
        dlclose(h);
        return 1;
    }
    g_rccl = a;
    return 0;
}
}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0;
    long long allreduces = 0;     // enqueued since creation (idhmc_comm_info)
};

static_assert(NCCL_UNIQUE_ID_BYTES == 128, "idhmc.h states IDHMC_COMM_ID_BYTES = 128");

int comm_unique_id(void *out128, char *err, size_t cap)
{
    if (load_rccl(err, cap)) return 1;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) { snprintf(err, cap, "ncclGetUniqueId: %s", g_rccl.error_string(r)); return 1; }
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

Comm *comm_create(int nranks, int rank, const void *id128, char *err, size_t cap)
{
    if (load_rccl(err, cap)) return nullptr;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    Comm *c = new Comm;
    c->nranks = nranks; c->rank = rank;
    const ncclResult_t r = g_rccl.comm_init_rank(&c->comm, nranks, id, rank);   // on the current device
    if (r != ncclSuccess) {
        snprintf(err, cap, "ncclCommInitRank(%d of %d): %s", rank, nranks, g_rccl.error_string(r));
        delete c;
        return nullptr;
    }
    return c;
}

int comm_allreduce_sum(Comm *c, double *dev_buf, int n, hipStream_t st, char *err, size_t cap)
{
    const ncclResult_t r = g_rccl.all_reduce(dev_buf, dev_buf, (size_t)n, ncclDouble, ncclSum, c->comm, st);
    if (r != ncclSuccess) { snprintf(err, cap, "ncclAllReduce: %s", g_rccl.error_string(r)); return 1; }
    ++c->allreduces;
    return 0;
}

void comm_destroy(Comm *c)
{
    if (!c) return;
    if (c->comm) (void)g_rccl.comm_destroy(c->comm);
    delete c;
}

void comm_info(const Comm *c, int *nranks, int *rank, long long *allreduces)
{
    *nranks = c ? c->nranks : 0;
    *rank = c ? c->rank : 0;
    *allreduces = c ? c->allreduces : 0;
}

}  // namespace idhmc
