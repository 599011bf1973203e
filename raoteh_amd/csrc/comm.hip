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

// RT_OK iff librccl can be loaded and exports what rt_comm_init needs.  Touches no
// GPU and no network: the ranks agree on this BEFORE any of them enters
// ncclCommInitRank (raoteh_amd/dist.py init_rccl), where a missing peer is a hang.
extern "C" int rt_comm_available(void)
{
    return load_rccl();
}

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
    if (ctx->ev_reduced) {
        hipEventDestroy(ctx->ev_reduced);
        ctx->ev_reduced = nullptr;
        for (auto &e : ctx->comm_events) { hipEventDestroy(e); e = nullptr; }
    }
    return RT_OK;
}

// The totals of `count` site batches in ONE collective.  The batches' slots in the
// context's totals arena must be consecutive and ascending (batches created one
// after the other are); otherwise every batch gets its own collective.  Cost of the
// stream bookkeeping (one event on the compute stream, a wait and an event on the
// comm stream): 10.7 us per call on an MI355X whatever the payload -- a quarter of a
// C2 step, which is why a ring of K rotating batches reduces once per K steps.
extern "C" int rt_allreduce_totals_group(rt_ctx *ctx, rt_sites **sites, int64_t count)
{
    RT_REQUIRE(ctx && sites && count >= 1, "bad arguments");
    RT_REQUIRE(ctx->comm, "rt_comm_init has not been called");
    for (int64_t k = 0; k < count; ++k) {
        RT_REQUIRE(sites[k] && sites[k]->model->ctx == ctx, "batch %lld: wrong context",
                   (long long)k);
    }
    RT_HIP(hipSetDevice(ctx->device));
    RT_TRY(rt_flush_reduce(ctx));        // the totals of the last step may still be partial sums
    if (!ctx->ev_reduced) {
        RT_HIP(hipEventCreateWithFlags(&ctx->ev_reduced, hipEventDisableTiming));
        for (auto &e : ctx->comm_events)
            RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    bool contiguous = sites[0]->totals_slot >= 0;
    for (int64_t k = 1; k < count && contiguous; ++k)
        contiguous = sites[k]->totals_slot == sites[0]->totals_slot + (int)k;
    // comm stream: wait for the totals (compute stream), reduce them in place, signal
    RT_HIP(hipEventRecord(ctx->ev_reduced, ctx->stream));
    RT_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_reduced, 0));
    if (contiguous) {
        const int rc = g_rccl.AllReduce(sites[0]->d_totals, sites[0]->d_totals,
                                        (size_t)(3 * count), NCCL_FLOAT64, NCCL_SUM,
                                        (nccl_comm)ctx->comm, ctx->comm_stream);
        if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    } else {
        for (int64_t k = 0; k < count; ++k) {
            const int rc = g_rccl.AllReduce(sites[k]->d_totals, sites[k]->d_totals, 3,
                                            NCCL_FLOAT64, NCCL_SUM, (nccl_comm)ctx->comm,
                                            ctx->comm_stream);
            if (rc != 0) return rccl_fail("ncclAllReduce", rc);
        }
    }
    hipEvent_t done = ctx->comm_events[ctx->comm_event_next];
    ctx->comm_event_next = (ctx->comm_event_next + 1) % 8;
    RT_HIP(hipEventRecord(done, ctx->comm_stream));
    for (int64_t k = 0; k < count; ++k) {
        // a later record of the same event only strengthens what it stands for (the
        // comm stream is in order)
        sites[k]->ev_comm_done = done;
        sites[k]->comm_pending = true;
    }
    return RT_OK;
}

extern "C" int rt_allreduce_totals(rt_ctx *ctx, rt_sites *s)
{
    RT_REQUIRE(ctx && s, "null pointer");
    return rt_allreduce_totals_group(ctx, &s, 1);
}
