// Multi-GPU: one process per GPU; the only data-path exchange of the site-sharded
// likelihood is the sum of three doubles (sum of log-likelihoods, number of
// zero-probability sites, number of sites), done with ncclAllReduce over
// RCCL/xGMI on the context's stream.  RCCL is loaded at run time (dlopen) so the
// single-GPU path has no dependency on it.
#include "common.h"

#include <dlfcn.h>

namespace {

typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm;
enum { NCCL_SUM = 0, NCCL_FLOAT64 = 8 };

struct rccl_api {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

rccl_api g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return RT_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        rt_set_error("cannot load librccl: %s", dlerror());
        return RT_ERR_RCCL;
    }
    g_rccl.GetUniqueId = (int (*)(nccl_uid *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank =
        (int (*)(nccl_comm *, int, nccl_uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(nccl_comm))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, nccl_comm,
                                hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy ||
        !g_rccl.AllReduce) {
        rt_set_error("librccl is missing a required symbol");
        dlclose(h);
        return RT_ERR_RCCL;
    }
    g_rccl.handle = h;
    return RT_OK;
}

int rccl_fail(const char *what, int code)
{
    rt_set_error("%s failed: %s", what,
                 g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "?");
    return RT_ERR_RCCL;
}

}  // namespace

extern "C" int rt_comm_unique_id(unsigned char id[128])
{
    RT_REQUIRE(id, "null id");
    RT_TRY(load_rccl());
    nccl_uid uid;
    const int rc = g_rccl.GetUniqueId(&uid);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, uid.internal, 128);
    return RT_OK;
}

extern "C" int rt_comm_init(rt_ctx *ctx, int nranks, int rank, const unsigned char id[128])
{
    RT_REQUIRE(ctx && id, "null pointer");
    RT_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank,
               nranks);
    RT_REQUIRE(!ctx->comm, "communicator already initialised");
    RT_TRY(load_rccl());
    RT_HIP(hipSetDevice(ctx->device));
    nccl_uid uid;
    memcpy(uid.internal, id, 128);
    nccl_comm comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, nranks, uid, rank);
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
    ctx->comm = comm;
    // collectives run on their own stream so the 24-byte all-reduce of step j
    // overlaps the kernels of step j+1 (it is pure latency)
    RT_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    return RT_OK;
}

extern "C" int rt_comm_destroy(rt_ctx *ctx)
{
    if (!ctx || !ctx->comm) return RT_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream) hipStreamSynchronize(ctx->comm_stream);
    g_rccl.CommDestroy((nccl_comm)ctx->comm);
    ctx->comm = nullptr;
    if (ctx->comm_stream) hipStreamDestroy(ctx->comm_stream);
    ctx->comm_stream = nullptr;
    return RT_OK;
}

extern "C" int rt_allreduce_totals(rt_ctx *ctx, rt_sites *s)
{
    RT_REQUIRE(ctx && s, "null pointer");
    RT_REQUIRE(ctx->comm, "rt_comm_init has not been called");
    RT_HIP(hipSetDevice(ctx->device));
    if (!s->ev_reduced) {
        RT_HIP(hipEventCreateWithFlags(&s->ev_reduced, hipEventDisableTiming));
        RT_HIP(hipEventCreateWithFlags(&s->ev_comm_done, hipEventDisableTiming));
    }
    // comm stream: wait for this batch's totals, reduce them in place, signal
    RT_HIP(hipEventRecord(s->ev_reduced, ctx->stream));
    RT_HIP(hipStreamWaitEvent(ctx->comm_stream, s->ev_reduced, 0));
    const int rc = g_rccl.AllReduce(s->d_totals, s->d_totals, 3, NCCL_FLOAT64, NCCL_SUM,
                                    (nccl_comm)ctx->comm, ctx->comm_stream);
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    RT_HIP(hipEventRecord(s->ev_comm_done, ctx->comm_stream));
    s->comm_pending = true;
    return RT_OK;
}
