// C ABI of libraoteh_hip.so: contexts, models (tree + transition matrices),
// site batches, and the host-pointer reference-shaped entry points.
// See include/raoteh_hip.h for the contract of every function.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <functional>
#include <cmath>
#include <cstdlib>

// ---- errors ----------------------------------------------------------------------

static thread_local char g_err[512] = "";

void rt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rt_last_error(void) { return g_err; }
extern "C" int rt_version(void) { return 100; }

extern "C" int rt_device_count(int *count)
{
    RT_REQUIRE(count, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        rt_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    *count = c;
    return RT_OK;
}

// ---- context ------------------------------------------------------------------------

extern "C" int rt_ctx_create(int device, rt_ctx **out)
{
    RT_REQUIRE(out, "null out pointer");
    *out = nullptr;
    int count = 0;
    RT_HIP(hipGetDeviceCount(&count));
    RT_REQUIRE(device >= 0 && device < count, "device %d not in [0,%d)", device, count);
    RT_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rt_set_error("device %d is %s; this library is built for gfx950 (MI355X) only",
                     device, prop.gcnArchName);
        return RT_ERR_UNSUPPORTED;
    }
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_ERR_NOMEM;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        rt_set_error("hipStreamCreate: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    e = hipMalloc((void **)&ctx->d_totals_arena, (size_t)RT_TOTALS_SLOTS * 3 * 8);
    if (e == hipSuccess) e = hipMemset(ctx->d_totals_arena, 0, (size_t)RT_TOTALS_SLOTS * 3 * 8);
    if (e == hipSuccess) e = hipDeviceSynchronize();    // the memset is asynchronous
    if (e != hipSuccess) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        rt_set_error("totals arena: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    // consecutive creations get consecutive slots (what rt_allreduce_totals_group needs)
    for (int k = RT_TOTALS_SLOTS - 1; k >= 0; --k) ctx->totals_free.push_back(k);
    *out = ctx;
    return RT_OK;
}

extern "C" int rt_ctx_sync(rt_ctx *ctx)
{
    RT_REQUIRE(ctx, "null context");
    RT_HIP(hipSetDevice(ctx->device));
    RT_TRY(rt_flush_reduce(ctx));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) RT_HIP(hipStreamSynchronize(ctx->comm_stream));
    return RT_OK;
}

static void drain_slot(rt_timing_slot &s, bool wait)
{
    size_t keep = 0;
    for (size_t i = 0; i < s.pending.size(); ++i) {
        auto &pr = s.pending[i];
        hipError_t q = wait ? hipEventSynchronize(pr.second) : hipEventQuery(pr.second);
        if (q == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                s.total_ms += ms;
                s.launches += 1;
            }
            s.pool.push_back(pr.first);
            s.pool.push_back(pr.second);
        } else {
            s.pending[keep++] = pr;
        }
    }
    s.pending.resize(keep);
}

extern "C" int rt_ctx_destroy(rt_ctx *ctx)
{
    if (!ctx) return RT_OK;
    hipSetDevice(ctx->device);
    rt_expect_cache_release(ctx);        // the library's own model + batch of the last call
    // the caller's models and chain batches hold this context: they go first
    if (ctx->live_models > 0 || ctx->live_chains > 0) {
        rt_set_error("rt_ctx_destroy: %d model(s) and %d chain batch(es) of this context are "
                     "still alive; destroy them first", ctx->live_models, ctx->live_chains);
        return RT_ERR_INVALID;
    }
    ctx->pending_reduce = nullptr;
    hipStreamSynchronize(ctx->stream);
    rt_comm_destroy(ctx);
    rt_jit_join_all(ctx);                // no compile thread of this context outlives it
    rt_jit_release(ctx);
    for (auto &s : ctx->slots) {
        drain_slot(s, true);
        for (auto ev : s.pool) hipEventDestroy(ev);
    }
    hipStreamDestroy(ctx->stream);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    hipFree(ctx->d_totals_arena);
    hipFree(ctx->d_scratch);
    hipFree(ctx->d_expm_scratch);
    delete ctx;
    return RT_OK;
}

extern "C" int rt_ctx_set_timing(rt_ctx *ctx, int enabled)
{
    RT_REQUIRE(ctx, "null context");
    // enabled = N > 0: time every N-th launch of each kernel (events cost a few
    // microseconds each, so a benchmark samples instead of timing every launch)
    ctx->timing = enabled != 0;
    ctx->timing_period = enabled > 1 ? enabled : 1;
    return RT_OK;
}

extern "C" int rt_ctx_reset_timing(rt_ctx *ctx)
{
    RT_REQUIRE(ctx, "null context");
    RT_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &s : ctx->slots) {
        drain_slot(s, true);
        s.total_ms = 0.0;
        s.launches = 0;
        s.seen = 0;
    }
    return RT_OK;
}

extern "C" int rt_ctx_kernel_time(rt_ctx *ctx, int kernel, double *total_ms,
                                  int64_t *launches, const char **name)
{
    RT_REQUIRE(ctx, "null context");
    RT_REQUIRE(kernel >= 0 && kernel < RT_K_COUNT, "bad kernel id %d", kernel);
    rt_timing_slot &s = ctx->slots[kernel];
    drain_slot(s, true);
    if (total_ms) *total_ms = s.total_ms;
    if (launches) *launches = s.launches;
    if (name) *name = s.name;
    return RT_OK;
}

static hipEvent_t slot_event(rt_timing_slot &s)
{
    hipEvent_t ev = nullptr;
    if (!s.pool.empty()) { ev = s.pool.back(); s.pool.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

// Sampled timing of ONE kernel launch: between begin and end the launch goes
// through RT_LAUNCH_TIMED / rt_launch_prune_jit, which hand ctx->ev_start/ev_stop to
// the runtime (hipExtLaunchKernelGGL / hipExtModuleLaunchKernel).
void rt_time_begin(rt_ctx *ctx, int kernel, const char *name, hipEvent_t *start)
{
    *start = nullptr;
    ctx->ev_start = ctx->ev_stop = nullptr;
    rt_timing_slot &s = ctx->slots[kernel];
    if (name && name[0]) snprintf(s.name, sizeof(s.name), "%s", name);
    if (!ctx->timing) return;
    if ((s.seen++ % ctx->timing_period) != 0) return;
    hipEvent_t a = slot_event(s), b = slot_event(s);
    if (!a || !b) {
        if (a) s.pool.push_back(a);
        if (b) s.pool.push_back(b);
        return;
    }
    ctx->ev_start = a;
    ctx->ev_stop = b;
    *start = a;
}

void rt_time_end(rt_ctx *ctx, int kernel, hipEvent_t start)
{
    if (!start) return;
    rt_timing_slot &s = ctx->slots[kernel];
    s.pending.emplace_back(ctx->ev_start, ctx->ev_stop);
    ctx->ev_start = ctx->ev_stop = nullptr;
    if (s.pending.size() > 4096) drain_slot(s, false);
}

bool rt_time_extra_begin(rt_ctx *ctx, int kernel, const char *name, hipEvent_t *a, hipEvent_t *b)
{
    *a = *b = nullptr;
    rt_timing_slot &s = ctx->slots[kernel];
    if (name && name[0]) snprintf(s.name, sizeof(s.name), "%s", name);
    if (!ctx->ev_start) return false;
    hipEvent_t x = slot_event(s), y = slot_event(s);
    if (!x || !y) {
        if (x) s.pool.push_back(x);
        if (y) s.pool.push_back(y);
        return false;
    }
    *a = x;
    *b = y;
    return true;
}

void rt_time_extra_end(rt_ctx *ctx, int kernel, hipEvent_t a, hipEvent_t b)
{
    if (!a) return;
    rt_timing_slot &s = ctx->slots[kernel];
    s.pending.emplace_back(a, b);
    if (s.pending.size() > 4096) drain_slot(s, false);
}

// ---- expm, host pointers ----------------------------------------------------------------

extern "C" int rt_expm(rt_ctx *ctx, int64_t n, int64_t count, const double *Q,
                       int64_t nq, const int64_t *q_index, const double *t, double *P,
                       int32_t *info)
{
    RT_REQUIRE(ctx, "null context");
    RT_REQUIRE(n >= 1 && count >= 0 && nq >= 1, "bad sizes");
    RT_REQUIRE(Q && t && P, "null array");
    if (n > RT_MAX_EXPM_STATES) {
        rt_set_error("expm: n=%lld > %d", (long long)n, RT_MAX_EXPM_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    RT_REQUIRE(q_index || nq == 1 || nq == count,
               "q_index is NULL but nq=%lld is neither 1 nor count", (long long)nq);
    if (count == 0) return RT_OK;
    std::vector<int32_t> qi((size_t)count);
    for (int64_t b = 0; b < count; ++b) {
        const int64_t v = q_index ? q_index[b] : (nq == 1 ? 0 : b);
        RT_REQUIRE(v >= 0 && v < nq, "q_index[%lld]=%lld out of range", (long long)b,
                   (long long)v);
        qi[(size_t)b] = (int32_t)v;
    }
    RT_HIP(hipSetDevice(ctx->device));
    // operands in the context's grow-only scratch: five hipMalloc / hipFree pairs per call cost
    // far more than the kernel (the call is synchronous, nothing else uses the scratch meanwhile)
    const size_t nn = (size_t)n * n;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_Q = 0, o_t = o_Q + up(nq * nn * 8), o_P = o_t + up(count * 8),
                 o_qi = o_P + up(count * nn * 8), o_info = o_qi + up(count * 4),
                 total = o_info + up(count * 8);
    RT_TRY(rt_scratch_reserve(ctx, total));
    unsigned char *base = ctx->d_scratch;
    double *dQ = (double *)(base + o_Q), *dt = (double *)(base + o_t), *dP = (double *)(base + o_P);
    int32_t *dqi = (int32_t *)(base + o_qi), *dinfo = (int32_t *)(base + o_info);
    int rc = RT_OK;
    hipError_t e = hipMemcpyAsync(dQ, Q, nq * nn * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dt, t, count * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dqi, qi.data(), count * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) rc = rt_launch_expm(ctx, n, count, dQ, dqi, dt, dP, dinfo, nullptr, 0, nullptr);
    if (e == hipSuccess && rc == RT_OK)
        e = hipMemcpyAsync(P, dP, count * nn * 8, hipMemcpyDeviceToHost, ctx->stream);
    std::vector<int32_t> hinfo((size_t)count * 2, 0);
    if (e == hipSuccess && rc == RT_OK)
        e = hipMemcpyAsync(hinfo.data(), dinfo, count * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == RT_OK) e = hipStreamSynchronize(ctx->stream);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) {
        rt_set_error("rt_expm: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    if (info) memcpy(info, hinfo.data(), (size_t)count * 8);
    for (int64_t b = 0; b < count; ++b)
        if (hinfo[2 * b] < 0) {
            rt_set_error("expm: matrix %lld has a non-finite entry or norm (Taylor scheme), or its Pade "
                         "denominator is singular (RAOTEH_EXPM=pade)", (long long)b);
            return RT_ERR_SINGULAR;
        }
    return RT_OK;
}

// ---- model -----------------------------------------------------------------------------------

// Post-order schedule with a Sethi-Ullman child order (the child whose subtree
// needs the most accumulator slots goes first), so the register stack of the
// fast kernels stays at <= floor(log2(#leaves)) + 1 slots for any tree shape.
static void build_schedule(rt_model *m)
{
    const int64_t N = m->nnodes;
    std::vector<std::vector<int32_t>> kids((size_t)N);
    m->parent.assign((size_t)N, -1);
    for (int64_t v = 0; v < N; ++v)
        for (int64_t e = m->indptr[v]; e < m->indptr[v + 1]; ++e) {
            kids[(size_t)v].push_back((int32_t)m->indices[e]);
            m->parent[(size_t)m->indices[e]] = (int32_t)v;
        }
    std::vector<int> need((size_t)N, 0);
    for (int64_t v = N - 1; v >= 0; --v) {
        auto &k = kids[(size_t)v];
        if (k.empty()) continue;
        std::stable_sort(k.begin(), k.end(),
                         [&](int32_t a, int32_t b) { return need[a] > need[b]; });
        int nd = std::max(1, need[k[0]]);
        for (size_t i = 1; i < k.size(); ++i) nd = std::max(nd, 1 + need[k[i]]);
        need[(size_t)v] = nd;
    }
    m->ops.clear();
    m->ops.reserve((size_t)N);
    std::vector<int32_t> acc_slot((size_t)N, -1);
    std::vector<std::pair<int32_t, size_t>> stack;   // (node, next child)
    stack.emplace_back(0, 0);
    int sp = 0, maxsp = 0;
    while (!stack.empty()) {
        const int32_t v = stack.back().first;
        const size_t ci = stack.back().second;
        auto &k = kids[(size_t)v];
        if (ci < k.size()) {
            stack.back().second = ci + 1;
            stack.emplace_back(k[ci], 0);
            continue;
        }
        stack.pop_back();
        rt_op op;
        op.node = v;
        op.obs = -1;
        op.pop = k.empty() ? -1 : acc_slot[(size_t)v];
        if (!k.empty()) sp -= 1;
        if (stack.empty()) {
            op.dst = -1;                       // root
        } else {
            const int32_t p = stack.back().first;
            if (acc_slot[(size_t)p] < 0) {
                acc_slot[(size_t)p] = sp;
                op.dst = sp | 256;
                sp += 1;
                maxsp = std::max(maxsp, sp);
            } else {
                op.dst = acc_slot[(size_t)p];
            }
        }
        m->ops.push_back(op);
    }
    m->max_depth = maxsp;
}

extern "C" int rt_build_schedule(int64_t nnodes, const int64_t *idx, const int64_t *ptr,
                                 int32_t *ops, int32_t *depth)
{
    RT_REQUIRE(nnodes >= 1 && ptr && (idx || nnodes == 1) && ops, "bad arguments");
    RT_REQUIRE(ptr[0] == 0 && ptr[nnodes] == nnodes - 1,
               "tree_csr_indptr does not describe a tree");
    for (int64_t v = 0; v < nnodes; ++v)
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e)
            RT_REQUIRE(idx[e] > v && idx[e] < nnodes, "children not in preorder");
    rt_model m;
    m.nnodes = nnodes;
    if (nnodes > 1) m.indices.assign(idx, idx + (nnodes - 1));
    m.indptr.assign(ptr, ptr + nnodes + 1);
    build_schedule(&m);
    memcpy(ops, m.ops.data(), m.ops.size() * sizeof(rt_op));
    if (depth) *depth = m.max_depth;
    return RT_OK;
}

static int64_t pfrag_doubles(const rt_model *m)
{
    const int64_t nops = (int64_t)m->ops.size();
    if (m->n <= 4) return (nops + 1) * m->n * m->n;   // + one zero record (fetch-ahead)
    const int64_t nt = (m->n + 15) / 16, kp = ((m->n + 3) / 4 + 1) / 2;
    return nops * nt * kp * 128;
}

extern "C" int rt_model_destroy(rt_model *m)
{
    if (!m) return RT_OK;
    // a site batch refers to its model (and through it to the context) until it is destroyed
    if (m->live_batches > 0) {
        rt_set_error("rt_model_destroy: %d site batch(es) of this model are still alive; "
                     "destroy them first", m->live_batches);
        return RT_ERR_INVALID;
    }
    m->ctx->live_models -= 1;
    hipSetDevice(m->ctx->device);
    rt_expect_state_release(m);
    rt_expect_lane_release(m);
    hipStreamSynchronize(m->ctx->stream);
    hipFree(m->d_indices); hipFree(m->d_indptr); hipFree(m->d_ops); hipFree(m->d_P);
    hipFree(m->d_Pfrag); hipFree(m->d_Pquad); hipFree(m->d_Pcol); hipFree(m->d_root); hipFree(m->d_Q); hipFree(m->d_spec); hipFree(m->d_qidx);
    hipFree(m->d_t); hipFree(m->d_info); hipFree(m->d_step_of_node);
    hipFree(m->d_qidx_step); hipFree(m->d_t_step);
    delete m;
    return RT_OK;
}

extern "C" int rt_model_create(rt_ctx *ctx, int64_t nnodes, int64_t n,
                               const int64_t *idx, const int64_t *ptr, rt_model **out)
{
    RT_REQUIRE(ctx && out, "null pointer");
    *out = nullptr;
    RT_REQUIRE(nnodes >= 1 && n >= 1, "bad sizes");
    RT_REQUIRE(n <= RT_MAX_STATES, "n=%lld > %d", (long long)n, RT_MAX_STATES);
    RT_REQUIRE(nnodes < (1ll << 30), "tree too large");
    RT_REQUIRE(ptr && (idx || nnodes == 1), "null array");
    RT_REQUIRE(ptr[0] == 0 && ptr[nnodes] == nnodes - 1,
               "tree_csr_indptr does not describe a tree");
    std::vector<char> seen((size_t)nnodes, 0);
    for (int64_t v = 0; v < nnodes; ++v) {
        RT_REQUIRE(ptr[v + 1] >= ptr[v], "tree_csr_indptr not monotone");
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) {
            RT_REQUIRE(idx[e] > v && idx[e] < nnodes,
                       "child index %lld of node %lld is not in preorder",
                       (long long)idx[e], (long long)v);
            RT_REQUIRE(!seen[(size_t)idx[e]], "node %lld has two parents",
                       (long long)idx[e]);
            seen[(size_t)idx[e]] = 1;
        }
    }
    RT_HIP(hipSetDevice(ctx->device));
    rt_model *m = new (std::nothrow) rt_model();
    if (!m) return RT_ERR_NOMEM;
    m->ctx = ctx;
    ctx->live_models += 1;            // (every failure path below goes through rt_model_destroy)
    m->nnodes = nnodes;
    m->n = n;
    if (nnodes > 1) m->indices.assign(idx, idx + (nnodes - 1));
    m->indptr.assign(ptr, ptr + nnodes + 1);
    build_schedule(m);
    const size_t nn = (size_t)n * n;
    hipError_t e = hipMalloc((void **)&m->d_indices, std::max<size_t>(8, (nnodes - 1) * 8));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_indptr, (nnodes + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_ops, m->ops.size() * sizeof(rt_op));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_P, nnodes * nn * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_Pfrag, pfrag_doubles(m) * 8);
    if (e == hipSuccess) e = hipMemset(m->d_Pfrag, 0, pfrag_doubles(m) * 8);
    if (e == hipSuccess && n > 4 && n <= 32) {
        const size_t bytes = m->ops.size() * (size_t)rt_quad_stride((int)n) * 8;
        e = hipMalloc((void **)&m->d_Pquad, bytes);
        if (e == hipSuccess) e = hipMemset(m->d_Pquad, 0, bytes);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_root, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_qidx, nnodes * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_t, nnodes * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_info, nnodes * 8);
    if (e == hipSuccess && n <= 4) e = hipMalloc((void **)&m->d_qidx_step, nnodes * 4);
    if (e == hipSuccess && n <= 4) e = hipMalloc((void **)&m->d_t_step, nnodes * 8);
    if (e == hipSuccess && nnodes > 1)
        e = hipMemcpy(m->d_indices, idx, (nnodes - 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_indptr, ptr, (nnodes + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(m->d_ops, m->ops.data(), m->ops.size() * sizeof(rt_op),
                      hipMemcpyHostToDevice);
    std::vector<int32_t> step_of((size_t)nnodes, -1);
    for (size_t k = 0; k < m->ops.size(); ++k) step_of[(size_t)m->ops[k].node] = (int32_t)k;
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_step_of_node, nnodes * 4);
    if (e == hipSuccess)
        e = hipMemcpy(m->d_step_of_node, step_of.data(), nnodes * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(m->d_P, 0, nnodes * nn * 8);
    if (e == hipSuccess) e = hipMemset(m->d_info, 0, nnodes * 8);
    std::vector<double> ones((size_t)n, 1.0);
    if (e == hipSuccess) e = hipMemcpy(m->d_root, ones.data(), n * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rt_set_error("rt_model_create: %s", hipGetErrorString(e));
        rt_model_destroy(m);
        return e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP;
    }
    *out = m;
    return RT_OK;
}

// rate-matrix index and branch length of every edge, by node and (n <= 4) by schedule step:
// an optimiser changes the rates far more often than the assignment of matrices to edges or
// the branch lengths, so these go up only when they differ from what the device holds
static int model_upload_edge_parameters(rt_model *m, const std::vector<int32_t> &qi,
                                        const std::vector<double> &tt)
{
    if (qi != m->h_qidx) {
        RT_HIP(hipMemcpy(m->d_qidx, qi.data(), m->nnodes * 4, hipMemcpyHostToDevice));
        m->h_qidx = qi;
        if (m->d_qidx_step) {
            std::vector<int32_t> by_step(m->ops.size());
            for (size_t k = 0; k < m->ops.size(); ++k) by_step[k] = qi[(size_t)m->ops[k].node];
            RT_HIP(hipMemcpy(m->d_qidx_step, by_step.data(), by_step.size() * 4, hipMemcpyHostToDevice));
        }
    }
    if (tt != m->h_t) {
        RT_HIP(hipMemcpy(m->d_t, tt.data(), m->nnodes * 8, hipMemcpyHostToDevice));
        m->h_t = tt;
        if (m->d_t_step) {
            std::vector<double> by_step(m->ops.size());
            for (size_t k = 0; k < m->ops.size(); ++k) by_step[k] = tt[(size_t)m->ops[k].node];
            RT_HIP(hipMemcpy(m->d_t_step, by_step.data(), by_step.size() * 8, hipMemcpyHostToDevice));
        }
    }
    return RT_OK;
}

static int model_run_expm(rt_model *m)
{
    // the expm epilogue also writes the step-ordered layout the pruning kernel reads
    // a reduction deferred by the previous rt_step rides on this launch
    rt_reduce_args red;
    const bool carry = rt_take_pending_reduce(m->ctx, &red);
    if (m->spectral) {
        const size_t nn = (size_t)m->n * m->n;
        RT_TRY(rt_launch_spectral(m->ctx, m->n, m->nnodes, m->d_spec, m->d_spec + 2 * nn,
                                  m->d_spec + nn, m->spectral_has_D ? m->d_spec + 2 * nn + m->n
                                                                    : nullptr,
                                  m->d_qidx, m->d_t, m->d_P, m->d_info, m->d_step_of_node,
                                  m->n <= 4 ? 0 : 1, m->d_Pfrag, carry ? &red : nullptr,
                                  m->d_Pquad));
        m->have_P = true;
        m->frag_dirty = false;
        return rt_model_pack_pcol(m);
    }
    RT_TRY(rt_launch_expm(m->ctx, m->n, m->nnodes, m->d_Q, m->d_qidx, m->d_t, m->d_P,
                          m->d_info, m->d_step_of_node, m->n <= 4 ? 0 : 1, m->d_Pfrag,
                          carry ? &red : nullptr, m->d_Pquad));
    m->have_P = true;
    m->frag_dirty = false;
    return rt_model_pack_pcol(m);
}

extern "C" int rt_model_set_rates(rt_model *m, const double *Q, int64_t nq,
                                  const int64_t *node_q, const double *t)
{
    RT_REQUIRE(m && Q && t, "null pointer");
    RT_REQUIRE(nq >= 1, "nq must be >= 1");
    if (m->n > RT_MAX_EXPM_STATES) {
        rt_set_error("expm: n=%lld > %d", (long long)m->n, RT_MAX_EXPM_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    RT_HIP(hipSetDevice(m->ctx->device));
    const size_t nn = (size_t)m->n * m->n;
    std::vector<int32_t> qi((size_t)m->nnodes);
    std::vector<double> tt((size_t)m->nnodes);
    qi[0] = -1;
    tt[0] = 0.0;
    for (int64_t v = 1; v < m->nnodes; ++v) {
        const int64_t q = node_q ? node_q[v] : 0;
        RT_REQUIRE(q >= 0 && q < nq, "node_q[%lld]=%lld out of range", (long long)v,
                   (long long)q);
        RT_REQUIRE(std::isfinite(t[v]), "branch length of node %lld is not finite",
                   (long long)v);
        qi[(size_t)v] = (int32_t)q;
        tt[(size_t)v] = t[v];
    }
    if (nq > m->q_capacity) {
        hipFree(m->d_Q);
        m->d_Q = nullptr;
        m->q_capacity = 0;
        RT_HIP(hipMalloc((void **)&m->d_Q, nq * nn * 8));
        m->q_capacity = nq;
    }
    hipStream_t st = m->ctx->stream;
    // pageable sources: hipMemcpy (synchronous) so the vectors above may go away
    RT_HIP(hipStreamSynchronize(st));
    RT_HIP(hipMemcpy(m->d_Q, Q, nq * nn * 8, hipMemcpyHostToDevice));
    // an optimiser changes the rates far more often than the assignment of matrices
    // to edges or the branch lengths: those go up only when they differ
    RT_TRY(model_upload_edge_parameters(m, qi, tt));
    m->spectral = false;
    return model_run_expm(m);
}

// One reversible rate matrix given by its decomposition (examples/p53/qtop.py:128-152):
// every edge's transition matrix is rebuilt from it (spectral.hip) now and at every later
// rt_model_recompute_transitions / rt_step, until rt_model_set_rates is called again.
extern "C" int rt_model_set_rates_spectral(rt_model *m, const double *A, const double *lam,
                                           const double *B, const double *D, const double *t)
{
    RT_REQUIRE(m && A && lam && B && t, "null pointer");
    if (m->n > 64) {
        rt_set_error("spectral reconstruction: n=%lld > 64", (long long)m->n);
        return RT_ERR_UNSUPPORTED;
    }
    RT_HIP(hipSetDevice(m->ctx->device));
    const size_t n = (size_t)m->n, nn = n * n;
    std::vector<int32_t> qi((size_t)m->nnodes, 0);
    std::vector<double> tt((size_t)m->nnodes);
    qi[0] = -1;
    tt[0] = 0.0;
    for (int64_t v = 1; v < m->nnodes; ++v) {
        RT_REQUIRE(std::isfinite(t[v]), "branch length of node %lld is not finite", (long long)v);
        tt[(size_t)v] = t[v];
    }
    for (size_t i = 0; i < n; ++i)
        RT_REQUIRE(std::isfinite(lam[i]) && (!D || D[i] >= 0.0), "eigenvalue / weight %zu", i);
    if (!m->d_spec) RT_HIP(hipMalloc((void **)&m->d_spec, (2 * nn + 2 * n) * 8));
    RT_HIP(hipStreamSynchronize(m->ctx->stream));
    RT_HIP(hipMemcpy(m->d_spec, A, nn * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(m->d_spec + nn, B, nn * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(m->d_spec + 2 * nn, lam, n * 8, hipMemcpyHostToDevice));
    if (D) RT_HIP(hipMemcpy(m->d_spec + 2 * nn + n, D, n * 8, hipMemcpyHostToDevice));
    m->spectral_has_D = D != nullptr;
    RT_TRY(model_upload_edge_parameters(m, qi, tt));
    m->spectral = true;
    return model_run_expm(m);
}

// getp_spectral_v2 (qtop.py:76-88) for `count` branch lengths in one launch: host in, host out
extern "C" int rt_expm_spectral(rt_ctx *ctx, int64_t n, int64_t count, const double *A,
                                const double *lam, const double *B, const double *D,
                                const double *t, double *P)
{
    RT_REQUIRE(ctx, "null context");
    RT_REQUIRE(n >= 1 && count >= 0, "bad sizes");
    RT_REQUIRE(A && lam && B && t && P, "null array");
    if (n > 64) {
        rt_set_error("spectral reconstruction: n=%lld > 64", (long long)n);
        return RT_ERR_UNSUPPORTED;
    }
    if (count == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_A = 0, o_B = o_A + up(nn * 8), o_lam = o_B + up(nn * 8), o_D = o_lam + up(n * 8),
                 o_t = o_D + up(n * 8), o_P = o_t + up(count * 8), total = o_P + up(count * nn * 8);
    RT_TRY(rt_scratch_reserve(ctx, total));
    unsigned char *base = ctx->d_scratch;
    double *dA = (double *)(base + o_A), *dB = (double *)(base + o_B), *dl = (double *)(base + o_lam),
           *dD = (double *)(base + o_D), *dt = (double *)(base + o_t), *dP = (double *)(base + o_P);
    int rc = RT_OK;
    hipError_t e = hipMemcpyAsync(dA, A, nn * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dB, B, nn * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dl, lam, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && D) e = hipMemcpyAsync(dD, D, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dt, t, count * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        rc = rt_launch_spectral(ctx, n, count, dA, dl, dB, D ? dD : nullptr, nullptr, dt, dP, nullptr,
                                nullptr, 0, nullptr);
    if (e == hipSuccess && rc == RT_OK)
        e = hipMemcpyAsync(P, dP, count * nn * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == RT_OK) e = hipStreamSynchronize(ctx->stream);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) {
        rt_set_error("rt_expm_spectral: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return RT_OK;
}

extern "C" int rt_model_recompute_transitions(rt_model *m)
{
    RT_REQUIRE(m, "null model");
    RT_REQUIRE(m->d_Q || m->spectral, "rt_model_set_rates has not been called");
    RT_HIP(hipSetDevice(m->ctx->device));
    return model_run_expm(m);
}

extern "C" int rt_model_set_transitions(rt_model *m, const double *esd)
{
    RT_REQUIRE(m && esd, "null pointer");
    RT_HIP(hipSetDevice(m->ctx->device));
    RT_HIP(hipStreamSynchronize(m->ctx->stream));
    RT_HIP(hipMemcpy(m->d_P, esd, (size_t)m->nnodes * m->n * m->n * 8,
                     hipMemcpyHostToDevice));
    m->have_P = true;
    m->frag_dirty = true;
    return rt_launch_pfrag(m);
}

extern "C" int rt_model_get_transitions(rt_model *m, double *esd)
{
    RT_REQUIRE(m && esd, "null pointer");
    RT_REQUIRE(m->have_P, "the model has no transition matrices yet");
    RT_HIP(hipSetDevice(m->ctx->device));
    RT_HIP(hipStreamSynchronize(m->ctx->stream));
    RT_HIP(hipMemcpy(esd, m->d_P, (size_t)m->nnodes * m->n * m->n * 8,
                     hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_model_get_expm_info(rt_model *m, int32_t *info)
{
    RT_REQUIRE(m && info, "null pointer");
    RT_HIP(hipSetDevice(m->ctx->device));
    RT_HIP(hipStreamSynchronize(m->ctx->stream));
    RT_HIP(hipMemcpy(info, m->d_info, (size_t)m->nnodes * 8, hipMemcpyDeviceToHost));
    for (int64_t v = 1; v < m->nnodes; ++v)
        if (info[2 * v] < 0) {
            rt_set_error("expm: the matrix of the edge above node %lld has a non-finite entry or "
                         "norm (Taylor scheme), or its Pade denominator is singular "
                         "(RAOTEH_EXPM=pade)", (long long)v);
            return RT_ERR_SINGULAR;
        }
    return RT_OK;
}

extern "C" int rt_model_set_root_distn(rt_model *m, const double *root_distn)
{
    RT_REQUIRE(m, "null model");
    RT_HIP(hipSetDevice(m->ctx->device));
    std::vector<double> w((size_t)m->n, 1.0);
    if (root_distn) w.assign(root_distn, root_distn + m->n);
    RT_HIP(hipStreamSynchronize(m->ctx->stream));
    RT_HIP(hipMemcpy(m->d_root, w.data(), m->n * 8, hipMemcpyHostToDevice));
    return RT_OK;
}

extern "C" int rt_model_schedule_depth(const rt_model *m)
{
    return m ? m->max_depth : -1;
}

// ---- sites ------------------------------------------------------------------------------------

// Process-wide defaults (rt_set_option); a context's own value (rt_ctx_set_option)
// takes precedence.  Atomics: two threads with a context each may create batches
// while a third changes a default.
static std::atomic<int> g_force_generic{0};
// tree-specialised kernels (jit.hip): -1 = automatic (batches of at least
// RT_JIT_MIN_WORK site-states), 0 = never, 1 = always
static std::atomic<int> g_jit{-1};
static const int64_t RT_JIT_MIN_WORK = 65536;
static std::atomic<int> g_jit_block_sites{0};    // 0 = automatic (see sites_jit)
// 1: rt_sites_create does not wait for hiprtc (MFMA family): the batch runs the interpreter
// kernel until the background job is done; 0: compile inside rt_sites_create
static std::atomic<int> g_jit_async{1};
// 1: batches created from now on rescale their messages by powers of two (interpreter kernels
// only; prune.hip, rescale_exponent): trees of a thousand leaves whose likelihood underflows f64
static std::atomic<int> g_rescale{0};
// 1: a batch of observed STATES at the leaves (split-M family) may get a kernel whose leaf steps
// gather columns of P instead of multiplying; 0: always the dense resident layout's products
static std::atomic<int> g_leaf_state_kernels{1};

static int parse_option(const char *key, int64_t value, int *which, int *out)
{
    RT_REQUIRE(key, "null key");
    if (strcmp(key, "force_generic") == 0) { *which = 0; *out = value != 0; return RT_OK; }
    if (strcmp(key, "jit") == 0) { *which = 1; *out = value < 0 ? -1 : value != 0; return RT_OK; }
    if (strcmp(key, "jit_block_sites") == 0) {
        RT_REQUIRE(value >= 0 && value <= 64, "jit_block_sites must be 0 (automatic) .. 64");
        *which = 2;
        *out = (int)value;
        return RT_OK;
    }
    if (strcmp(key, "jit_async") == 0) { *which = 3; *out = value != 0; return RT_OK; }
    if (strcmp(key, "rescale") == 0) { *which = 4; *out = value != 0; return RT_OK; }
    if (strcmp(key, "leaf_state_kernels") == 0) { *which = 5; *out = value != 0; return RT_OK; }
    rt_set_error("unknown option %s", key);
    return RT_ERR_INVALID;
}

extern "C" int rt_set_option(const char *key, int64_t value)
{
    int which = 0, v = 0;
    RT_TRY(parse_option(key, value, &which, &v));
    (which == 0 ? g_force_generic : which == 1 ? g_jit : which == 2 ? g_jit_block_sites
     : which == 3 ? g_jit_async : which == 4 ? g_rescale : g_leaf_state_kernels).store(v);
    return RT_OK;
}

extern "C" int rt_ctx_set_option(rt_ctx *ctx, const char *key, int64_t value)
{
    RT_REQUIRE(ctx, "null context");
    int which = 0, v = 0;
    if (key && value == RT_OPT_UNSET) {          // back to the process-wide default
        RT_TRY(parse_option(key, 0, &which, &v));
        v = RT_OPT_UNSET;
    } else {
        RT_TRY(parse_option(key, value, &which, &v));
    }
    (which == 0 ? ctx->opt_force_generic : which == 1 ? ctx->opt_jit
     : which == 2 ? ctx->opt_jit_block_sites : which == 3 ? ctx->opt_jit_async
     : which == 4 ? ctx->opt_rescale : ctx->opt_leaf_state_kernels) = v;
    return RT_OK;
}

static int opt_force_generic(const rt_ctx *c)
{
    return c->opt_force_generic != RT_OPT_UNSET ? c->opt_force_generic : g_force_generic.load();
}
static int opt_jit(const rt_ctx *c) { return c->opt_jit != RT_OPT_UNSET ? c->opt_jit : g_jit.load(); }
static int opt_jit_async(const rt_ctx *c)
{
    if (const char *v = getenv("RAOTEH_JIT_ASYNC")) return atoi(v) != 0;
    return c->opt_jit_async != RT_OPT_UNSET ? c->opt_jit_async : g_jit_async.load();
}
static int opt_rescale(const rt_ctx *c)
{
    if (const char *v = getenv("RAOTEH_RESCALE")) return atoi(v) != 0;
    return c->opt_rescale != RT_OPT_UNSET ? c->opt_rescale : g_rescale.load();
}
static int opt_leaf_state_kernels(const rt_ctx *c)
{
    return c->opt_leaf_state_kernels != RT_OPT_UNSET ? c->opt_leaf_state_kernels
                                                     : g_leaf_state_kernels.load();
}
static int opt_jit_block_sites(const rt_ctx *c)
{
    return c->opt_jit_block_sites != RT_OPT_UNSET ? c->opt_jit_block_sites : g_jit_block_sites.load();
}

extern "C" int rt_sites_destroy(rt_sites *s)
{
    if (!s) return RT_OK;
    if (s->counted) s->model->live_batches -= 1;
    hipSetDevice(s->model->ctx->device);
    if (s->model->ctx->pending_reduce == s) s->model->ctx->pending_reduce = nullptr;
    hipStreamSynchronize(s->model->ctx->stream);
    // an all-reduce of this batch's totals may still be in flight on the comm stream
    rt_jit_ref(s->model->ctx, s->jit_fn, -1);
    if (s->jit_fn2) rt_jit_ref(s->model->ctx, s->jit_fn2, -1);
    if (s->expect_twin) rt_sites_destroy(s->expect_twin);
    hipFree(s->d_weights);
    hipFree(s->d_sets);
    hipFree(s->d_raw);
    hipFree(s->d_raw_src);
    if (!s->obs_borrowed) hipFree(s->d_obs);
    hipFree(s->d_ops); hipFree(s->d_lane_ops); hipFree(s->d_lane_ops_a); hipFree(s->d_lane_ops_b); hipFree(s->d_loglik); hipFree(s->d_status);
    hipFree(s->d_down_meta);
    if (s->model->ctx->comm_stream) hipStreamSynchronize(s->model->ctx->comm_stream);
    hipFree(s->d_partial); hipFree(s->d_partial_alt); hipFree(s->d_scratch); hipFree(s->d_half);
    hipFree(s->d_half_count);
    hipFree(s->d_leafw);
    if (s->totals_slot >= 0) {
        // keep the free list sorted (descending) so that batches created one after
        // the other keep getting neighbouring slots
        auto &fl = s->model->ctx->totals_free;
        fl.insert(std::lower_bound(fl.begin(), fl.end(), s->totals_slot, std::greater<int>()),
                  s->totals_slot);
    } else {
        hipFree(s->d_totals);
    }
    delete s;
    return RT_OK;
}

// Lane-kernel program: simulate the register cache of the top accumulator and
// turn slots into LDS byte offsets (see LOP_* in prune.hip).
static std::vector<int32_t> lane_program(const std::vector<rt_op> &ops, int64_t slot_bytes, bool fuse)
{
    enum { INTERNAL = 1, X_CUR = 2, FIRST = 4, ROOT = 8, SPILL = 16, DST_CUR = 32, OBS = 64,
           FAST = 128, CHERRY = 256 };
    std::vector<int32_t> prog;
    prog.reserve(ops.size() * 4);
    int cur = -1;                              // slot cached in registers
    auto leaf_obs = [&](size_t k) {
        return k < ops.size() && ops[k].pop < 0 && ops[k].obs >= 0 && ops[k].dst >= 0;
    };
    for (size_t k = 0; k < ops.size(); ++k) {
        const rt_op &op = ops[k];
        int flags = 0, pop_off = 0, dst_off = 0, spill_off = 0;
        // the parent's own step follows immediately <=> this is its last child
        auto parent_next = [&](size_t kk, int d) {
            return kk + 1 < ops.size() && ops[kk + 1].pop == d;
        };
        // ---- fused cherry: leaf a (first child), leaf b, then their parent ----
        if (fuse && leaf_obs(k) && (op.dst >> 8) && leaf_obs(k + 1) &&
            !(ops[k + 1].dst >> 8) && (ops[k + 1].dst & 255) == (op.dst & 255) &&
            k + 2 < ops.size() && ops[k + 2].pop == (op.dst & 255) &&
            ops[k + 2].obs < 0 && ops[k + 2].dst >= 0) {
            const rt_op &par = ops[k + 2];
            flags = CHERRY | OBS;
            // a, b and p pass through temporaries: the register cache is only
            // touched by the parent's result
            const int d = par.dst & 255;
            if (par.dst >> 8) {                // p is a first child: cur = t
                flags |= FIRST;
                if (cur >= 0) { flags |= SPILL; spill_off = (int)(cur * slot_bytes); }
                cur = d;
            } else if (cur == d) {             // the grandparent's accumulator is cached
                flags |= DST_CUR;
            } else {
                dst_off = (int)(d * slot_bytes);
                if (cur < 0 && parent_next(k + 2, d)) { flags |= FAST; cur = d; }  // un-spill
            }
            prog.insert(prog.end(), {flags, 0, dst_off, spill_off});
            k += 2;
            continue;
        }
        bool x_ok;                             // x is the observation or the cache
        if (op.obs >= 0) flags |= OBS;
        if (op.pop >= 0) {
            flags |= INTERNAL;
            if (cur == op.pop) { flags |= X_CUR; cur = -1; x_ok = op.obs < 0; }
            else { pop_off = (int)(op.pop * slot_bytes); x_ok = false; }
        } else {
            x_ok = op.obs >= 0;
        }
        if (op.dst < 0) flags |= ROOT;
        else {
            const int d = op.dst & 255;
            if (op.dst >> 8) {
                flags |= FIRST;
                if (cur >= 0) { flags |= SPILL; spill_off = (int)(cur * slot_bytes); }
                cur = d;
                if (x_ok) flags |= FAST;
            } else if (cur == d) {
                flags |= DST_CUR;
                if (x_ok) flags |= FAST;
            } else {
                dst_off = (int)(d * slot_bytes);
                if (x_ok && cur < 0 && parent_next(k, d)) {
                    // un-spill: cur = lds[d] * t; the LDS copy is dead afterwards
                    flags |= FAST;
                    cur = d;
                }
            }
        }
        prog.insert(prog.end(), {flags, pop_off, dst_off, spill_off});
    }
    prog.insert(prog.end(), {512 /* LOP_STOP */, 0, 0, 0});     // sentinel
    return prog;
}

// number of LDS stack slots a program touches (offsets are slot * slot_bytes)
static int program_stack_slots(const std::vector<int32_t> &prog, int64_t slot_bytes)
{
    int64_t top = 0;
    for (size_t k = 0; k + 3 < prog.size(); k += 4)
        top = std::max<int64_t>(top, std::max(prog[k + 1], std::max(prog[k + 2], prog[k + 3])));
    return (int)(top / slot_bytes) + 1;
}

static bool want_root_halves(const rt_sites *s, int64_t ntiles);

// sizes of the resident layout (blocks, bytes, partial sums) -> *padded = sites incl. padding
static void sites_layout_sizes(rt_sites *s, int64_t *padded_out)
{
    rt_model *m = s->model;
    const int64_t n = m->n;
    const int64_t K = s->nobs;
    int64_t padded;
    if (s->layout == RT_LAYOUT_LANE) {
        const int64_t S = s->block_sites;
        s->nblocks = std::max<int64_t>(1, (s->nsites + S - 1) / S);
        // an even number of blocks: the LDS-DMA kernel may give a wave two
        if (S == 64) s->nblocks = (s->nblocks + 1) & ~1ll;
        const int64_t np = (n + 1) & ~1ll;
        s->obs_bytes = s->compact_states ? s->nblocks * ((K + 3) / 4) * S * 4
                                         : s->nblocks * K * S * np * 8;
        padded = s->nblocks * S;
        s->npartials = s->nblocks;
    } else {
        s->nblocks = std::max<int64_t>(1, (s->nsites + 15) / 16);
        const int64_t kp = ((n + 3) / 4 + 1) / 2;
        s->obs_bytes = s->nblocks * K * kp * 128 * 8;
        padded = s->nblocks * 16;
        const int64_t nt = (n + 15) / 16;
        const int64_t waves = nt == 3 ? 3 : std::max<int64_t>(4, nt);
        const int64_t tiles = waves / nt;
        s->npartials = s->mfma_solo ? (s->nblocks + 3) / 4 * 4
                                    : (s->nblocks + tiles - 1) / tiles * waves;
    }
    *padded_out = padded;
}

static int sites_alloc(rt_sites *s, bool generic)
{
    rt_model *m = s->model;
    const int64_t n = m->n;
    int64_t padded = 0;
    sites_layout_sizes(s, &padded);
    hipError_t e = hipMalloc((void **)&s->d_ops, s->ops.size() * sizeof(rt_op));
    if (e == hipSuccess && !s->obs_borrowed)
        e = hipMalloc((void **)&s->d_obs, std::max<int64_t>(s->obs_bytes, 1024));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_loglik, padded * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_status, padded * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_partial, s->npartials * 16);
    // entries no kernel writes (padding of the last workgroup) must read as zero
    if (e == hipSuccess) e = hipMemset(s->d_partial, 0, s->npartials * 16);
    if (e == hipSuccess && s->jit_fused) {
        e = hipMalloc((void **)&s->d_partial_alt, s->npartials * 16);
        if (e == hipSuccess) e = hipMemset(s->d_partial_alt, 0, s->npartials * 16);
    }
    if (e == hipSuccess) {
        rt_ctx *ctx = m->ctx;
        if (!ctx->totals_free.empty()) {
            s->totals_slot = ctx->totals_free.back();
            ctx->totals_free.pop_back();
            s->d_totals = ctx->d_totals_arena + 3 * (size_t)s->totals_slot;
        } else {
            e = hipMalloc((void **)&s->d_totals, 3 * 8);
        }
    }
    if (e == hipSuccess) e = hipMemset(s->d_totals, 0, 3 * 8);
    if (e == hipSuccess && generic) {
        s->scratch_bytes = std::max<int64_t>(1, m->max_depth) * n * padded * 8;
        e = hipMalloc((void **)&s->d_scratch, s->scratch_bytes);
    }
    if (e == hipSuccess && s->jit_halves)      // [tile][half][k-step][lane] (jit.hip)
        e = hipMalloc((void **)&s->d_half,       // (padded to whole groups of <= 8 tiles)
                      (size_t)(s->nblocks + 8) * 2 * ((n + 15) / 16) * 4 * 64 * 8);
    if (e == hipSuccess && s->sparse_ok)
        e = hipMalloc((void **)&s->d_leafw,
                      (size_t)s->nblocks * (s->sparse_pairs ? (s->nobs + 1) / 2 : (s->nobs + 3) / 4) * 16 * 4 + 64);
    if (e == hipSuccess && s->jit_halves) {
        e = hipMalloc((void **)&s->d_half_count, (size_t)(s->nblocks + 8) * 4);
        if (e == hipSuccess) e = hipMemset(s->d_half_count, 0, (size_t)(s->nblocks + 8) * 4);
    }
    if (e == hipSuccess)
        e = hipMemcpy(s->d_ops, s->ops.data(), s->ops.size() * sizeof(rt_op),
                      hipMemcpyHostToDevice);
    if (e == hipSuccess && !generic) {
        // lane family: N doubles per slot and lane; MFMA family: 4 (own rows)
        // cherries are fused only for the LDS-DMA lane kernel (RAOTEH_LANE_NO_FUSE
        // turns it off for A/B runs)
        const bool fuse = s->layout == RT_LAYOUT_LANE && s->lane_dma &&
                          (s->lane_ring == 0 || s->lane_ring >= 2) &&
                          !getenv("RAOTEH_LANE_NO_FUSE");
        // bytes of one accumulator slot of one wave: lane family n doubles per lane,
        // MFMA split-M 4 (own rows), MFMA solo the whole message (NT*4)
        const int64_t slot_bytes = s->layout == RT_LAYOUT_LANE ? n * 512
                                 : s->mfma_solo ? ((n + 15) / 16) * 4 * 512 : 2048;
        const std::vector<int32_t> prog = lane_program(s->ops, slot_bytes, fuse);
        s->lane_nprog = (int64_t)(prog.size() / 4) - 1;     // without the sentinel
        s->lane_stack_slots = program_stack_slots(prog, slot_bytes);
        e = hipMalloc((void **)&s->d_lane_ops, prog.size() * 4);
        if (e == hipSuccess)
            e = hipMemcpy(s->d_lane_ops, prog.data(), prog.size() * 4, hipMemcpyHostToDevice);
        // Split-M interpreter kernel, root halves (prune.hip): what a batch of a few tiles
        // per CU runs while it has no tree-specialised kernel (a fresh topology) -- the same
        // cut and the same policy as the specialised kernel's (want_root_halves).
        // RAOTEH_INTERP_HALVES=0 / 1 overrides.
        std::vector<rt_op> opsA, opsB;
        bool ih = e == hipSuccess && s->layout == RT_LAYOUT_MFMA && !s->mfma_solo && !s->rescale &&
                  !s->obs_borrowed && rt_split_at_root(s->ops, &opsA, &opsB);
        if (ih) {
            if (const char *v = getenv("RAOTEH_INTERP_HALVES")) ih = atoi(v) != 0;
            else ih = want_root_halves(s, s->nblocks);
        }
        if (ih) {
            // the root's own observation is the combine kernel's; B's only child of the root
            // is a first child there
            s->half_kroot = opsA.back().obs;
            opsA.back().obs = -1;
            opsB.back().obs = -1;
            rt_op &bc = opsB[opsB.size() - 2];
            bc.dst = (bc.dst & 255) | 256;
            const std::vector<int32_t> pa = lane_program(opsA, slot_bytes, false);
            const std::vector<int32_t> pb = lane_program(opsB, slot_bytes, false);
            s->lane_stack_slots = std::max(s->lane_stack_slots,
                                           std::max(program_stack_slots(pa, slot_bytes),
                                                    program_stack_slots(pb, slot_bytes)));
            s->half_nops[0] = (int)opsA.size();
            s->half_nops[1] = (int)opsB.size();
            s->half_rec1 = (int)opsA.size() - 1;
            s->half_kobs1 = 0;
            for (size_t k = 0; k + 1 < opsA.size(); ++k) s->half_kobs1 += opsA[k].obs >= 0;
            e = hipMalloc((void **)&s->d_lane_ops_a, pa.size() * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&s->d_lane_ops_b, pb.size() * 4);
            if (e == hipSuccess)
                e = hipMemcpy(s->d_lane_ops_a, pa.data(), pa.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess)
                e = hipMemcpy(s->d_lane_ops_b, pb.data(), pb.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess && !s->d_half)
                e = hipMalloc((void **)&s->d_half,
                              (size_t)(s->nblocks + 8) * 2 * ((n + 15) / 16) * 4 * 64 * 8);
            s->interp_halves = e == hipSuccess;
        }
    }
    if (e != hipSuccess) {
        rt_set_error("rt_sites: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_HIP;
    }
    return RT_OK;
}

// Tree-specialised kernel for this batch (lane family): source from the
// schedule, compiled once per distinct (tree, observed nodes) and device.
// How rt_sites_create picks the pruning kernel of a batch: the automatic policy, the
// interpreter kernels only, or exactly the tree-specialised kernel of another batch
// (the probe batches of verify_jit_kernel).
// the split-M generator in use: pipelined unless RAOTEH_JIT_SPLIT=serial (A/B runs)
static std::string split_source(const std::vector<rt_op> &ops, int n, int K, int T, int D, int LA,
                                bool halves = false)
{
    const char *v = getenv("RAOTEH_JIT_SPLIT");
    if (v && strcmp(v, "serial") == 0 && !halves) return rt_jit_mfma_split_source(ops, n, K, T, D, LA);
    return rt_jit_mfma_split_pipelined_source(ops, n, K, T, D, LA, halves);
}

// Root halves (jit.hip) for a split-M batch of `ntiles` tiles at one tile per workgroup:
// a workgroup's time is its chain on the matrix pipe, a CU runs three at a time and the
// kernel ends with the busiest CU, so what counts is ceil(tiles / CUs) against
// ceil(2 tiles / CUs) / 2.  Worth it when that is >= 7 % less (the second kernel and the
// half buffer cost a few us) and the two programs are of comparable length (the longer
// one bounds the gain).  RAOTEH_JIT_HALVES=0 / 1 overrides.
static bool want_root_halves(const rt_sites *s, int64_t ntiles)
{
    int a = 0, b = 0;
    rt_jit_root_halves(s->ops, &a, &b);
    if (a == 0 || b == 0) return false;
    if (const char *v = getenv("RAOTEH_JIT_HALVES")) return atoi(v) != 0;
    const double cus = std::max(1, s->model->ctx->num_cus);
    const double whole = std::ceil((double)ntiles / cus) * (a + b);
    // 2 ntiles workgroups, alternating A / B: a CU's share in the worst case
    const double halves = std::ceil(2.0 * (double)ntiles / cus) * 0.5 * 2.0 * std::max(a, b);
    return halves <= 0.93 * whole;
}

// the halves form's second kernel
static int sites_halves_setup(rt_sites *s)
{
    RT_TRY(rt_jit_companion(s->model->ctx, s->jit_fn, "rt_jit_combine", &s->jit_combine));
    s->jit_halves = true;          // sites_alloc allocates d_half
    // (the pipelined generator folds the combine step into the pruning kernel; the serial
    // generator, RAOTEH_JIT_SPLIT=serial, has no halves form)
    // (... and the leaf-state form keeps the two kernels)
    s->jit_fold = rt_jit_fold_enabled() && !s->jit_sparse;
    return RT_OK;
}

struct jit_override {
    int mode = 0;             // 0 automatic, 1 interpreter only, 2 exactly these parameters
    int T = 1, S = 64, WG = 1, D = 1, LA = 1, compact = 0;
    bool quad = false;
    bool halves = false;      // split-M family: the two root programs as separate workgroups
    bool fuse = false;        // lane family: the kernel can compute its own transitions
    bool no_solo = false;     // n <= 32: the split-M interpreter kernel, not the one-wave one
    bool sparse = false;      // split-M family: leaf steps gather columns of P (leaf states)
    bool pipe = false;        // ... from the pipelined generator (leaves as factors)
};

// At most this many background compiles at a time (RAOTEH_JIT_MAX_JOBS, default 2): a caller
// that creates batches over ever new trees (a search over topologies with "jit" left on
// automatic) would otherwise start a host thread inside hiprtc per tree.  A batch created
// while the limit is reached stays on the interpreter kernel (same numbers).
static bool jit_jobs_full()
{
    int limit = 2;
    if (const char *v = getenv("RAOTEH_JIT_MAX_JOBS")) limit = std::max(0, atoi(v));
    return rt_jit_jobs_pending() >= limit;
}

// Background compile for an MFMA-family batch: candidates the cache already knows as
// rejected are skipped; if the first one left is usable the caller's synchronous path takes
// it from the cache (fast: -> false), else a job compiles the remaining ones in order, or
// none may start now and the batch keeps the interpreter kernel (-> true: nothing more to do
// in rt_sites_create).
template <class MakeSource>
static bool sites_jit_start_async(rt_sites *s, const std::vector<rt_sites::jit_cand> &cands,
                                  MakeSource make)
{
    if (jit_jobs_full()) {
        // without generating every candidate's text: is the preferred usable form cached?
        for (const auto &c : cands) {
            const std::string src = make(c);
            if (src.empty()) continue;
            const int known = rt_jit_cached(s->model->ctx, src);
            if (known < 0) continue;
            if (known > 0) return false;
            break;
        }
        return true;
    }
    std::vector<rt_sites::jit_cand> todo;
    std::vector<std::string> srcs;
    for (const auto &c : cands) {
        std::string src = make(c);
        if (src.empty()) continue;                    // (this form does not exist for this tree)
        const int known = rt_jit_cached(s->model->ctx, src);
        if (known < 0 && todo.empty()) continue;      // spilled or failed verification earlier
        if (known > 0 && todo.empty()) return false;  // in the cache: no job needed
        todo.push_back(c);
        srcs.push_back(std::move(src));
    }
    if (todo.empty()) return false;
    s->jit_cands = todo;
    s->jit_srcs = srcs;
    s->jit_job = rt_jit_start(s->model->ctx, std::move(srcs), true);
    return true;
}

static int sites_jit(rt_sites *s, bool generic, int kind, const jit_override *ov)
{
    if (ov && ov->mode == 1) return RT_OK;
    if (ov && ov->mode == 2) {
        const int n = (int)s->model->n, K = (int)s->nobs;
        const bool mfma = s->layout == RT_LAYOUT_MFMA;
        const bool split = mfma && (n > 32 || !s->mfma_solo);
        const std::string src =
            !mfma ? rt_jit_lane_source(s->ops, n, K, ov->D, ov->LA, ov->S, ov->WG, ov->compact,
                                       ov->fuse)
            : split && ov->sparse && ov->pipe
                ? rt_jit_mfma_split_pipelined_source(s->ops, n, K, ov->T, ov->D, ov->LA, ov->halves,
                                                     s->sparse_pairs ? 2 : 1)
            : split && ov->sparse ? rt_jit_mfma_split_source(s->ops, n, K, ov->T, ov->D, ov->LA,
                                                             s->sparse_pairs ? 2 : 1)
            : split ? split_source(s->ops, n, K, ov->T, ov->D, ov->LA, ov->halves)
                    : rt_jit_mfma_source(s->ops, n, K, ov->T, ov->D, ov->LA, ov->quad,
                                         ov->sparse ? (s->sparse_pairs ? 2 : 1) : 0);
        RT_TRY(rt_jit_get(s->model->ctx, src, &s->jit_fn, mfma));
        s->jit_sparse = mfma && ov->sparse;
        s->jit_pipe = split && ov->sparse && ov->pipe;
        if (split && ov->halves && (!ov->sparse || ov->pipe)) RT_TRY(sites_halves_setup(s));
        s->jit_quad = mfma && !split && ov->quad;
        s->jit_prefetch = ov->D;
        s->jit_lookahead = ov->LA;
        s->jit_tiles = ov->T;
        if (!mfma) {
            s->block_sites = ov->S;
            s->jit_waves = ov->WG;
            s->compact_states = ov->compact;
            s->jit_fused = ov->fuse;
        } else if (split) {
            s->jit_waves = (n + 15) / 16;
        }
        return RT_OK;
    }
    int want = opt_jit(s->model->ctx);
    if (const char *v = getenv("RAOTEH_JIT")) want = atoi(v);
    const bool forced = want > 0;
    // automatic: enough work per batch to be worth a second or two of compilation
    // (16 384 sites at 4 states, 1 075 at 61)
    if (want < 0) want = s->nsites * s->model->n >= RT_JIT_MIN_WORK;
    if (!want || generic) return RT_OK;
    // straight-line code, one block of arithmetic per step and tile: bound what
    // hiprtc is asked to compile (64 leaves x 4 states: 1.7 s; 512 leaves: 11 s)
    if (!forced && s->ops.size() > 600) return RT_OK;
    if (s->ops.size() > 2048) return RT_OK;
    if (s->layout == RT_LAYOUT_MFMA) {
        // one wave = T site tiles (jit.hip, MFMA family): n <= 32 only (row tiles of
        // larger matrices do not fit one wave's registers)
        if (s->model->n > 32 || !s->mfma_solo) {
            // split-M family: NT waves share T tiles.  T = 2 halves the A-fragment
            // traffic and the barriers per MFMA but needs the whole register file
            // (one workgroup per CU): worth it once the batch is several rounds deep
            const bool wide = s->model->n > 64;     // NT = 5..8 waves: one tile per workgroup
            const int64_t ntiles = (s->nsites + 15) / 16;
            // tiles per workgroup: T = 2 (one A fetch and one barrier per two chains, two
            // workgroups per CU) once the batch is several rounds deep; below that one tile
            // per workgroup, three workgroups per CU.  Measured with the pipelined
            // generator: config 3 (625 tiles) 205 us at T = 1, 211 us at T = 3 (209
            // workgroups, one per CU), 253 us at T = 2; a config-4 shard (7 813 tiles)
            // 1 984 us at T = 2, 2 040 us at T = 1, 2 303 us at T = 3.
            int T = (ntiles >= 2048 && s->ops.size() <= 300 && !wide) ? 2 : 1;
            bool halves = T == 1 && want_root_halves(s, ntiles);
            // Root halves with as many half-tiles per workgroup as make the launch ONE round
            // of at most one workgroup per CU (3..5 independent chains per SIMD keep the pipe
            // full, the whole register file): no dependence on where the dispatcher puts a
            // second round.  Config 3: 1 250 half-tiles = 250 workgroups of 5 on 256 CUs.
            if (halves && !getenv("RAOTEH_JIT_HALVES_T1")) {
                const int64_t cus = std::max(1, s->model->ctx->num_cus);
                const int64_t th = (2 * ntiles + cus - 1) / cus;
                if (th >= 3 && th <= 5 && s->ops.size() <= 300 && !wide) T = (int)th;
            }
            if (const char *v = getenv("RAOTEH_JIT_TILES"))
                T = std::min(wide ? 2 : halves ? 5 : 3, std::max(1, atoi(v)));
            int D = 2, LA = 1;     // leaves fetched ahead (the pipelined generator needs >= 2)
            if (const char *v = getenv("RAOTEH_JIT_PREFETCH")) D = std::max(1, atoi(v));
            if (const char *v = getenv("RAOTEH_JIT_LOOKAHEAD")) LA = std::max(1, atoi(v));
            s->jit_prefetch = D;
            s->jit_lookahead = LA;
            int rc = RT_ERR_UNSUPPORTED;
            // Observed states at the leaves (type x): the kernel whose leaf steps gather
            // columns of P instead of multiplying (half of the steps of a binary tree); one or
            // two tiles per workgroup, the serial generator (few steps consume their
            // predecessor once the leaves are out of the chain of matrix steps)
            if (s->sparse_ok) {
                int Ts = (ntiles >= 2048 && !wide) ? 2 : 1;
                if (const char *v = getenv("RAOTEH_JIT_TILES")) Ts = std::min(2, std::max(1, atoi(v)));
                std::vector<rt_sites::jit_cand> cands;
                // first the pipelined generator with the leaves as factors of their parents'
                // expressions (n <= 64; the dense kernels' tiling: root halves at T, at one tile,
                // the whole tree), then the serial generator; RAOTEH_JIT_SPARSE=serial: only that
                const char *sv = getenv("RAOTEH_JIT_SPARSE");
                if (!(sv && strcmp(sv, "serial") == 0)) {
                    for (int t = T, h = halves;;) {
                        cands.push_back({t, h != 0, false, true, true});
                        if (h && t > 1) t = 1;
                        else if (h) h = 0;
                        else if (t > 1) --t;
                        else break;
                    }
                }
                for (int t = Ts; t >= 1; --t) cands.push_back({t, false, false, true, false});
                const int smode = s->sparse_pairs ? 2 : 1;
                auto make = [&](const rt_sites::jit_cand &c) {
                    return c.pipe ? rt_jit_mfma_split_pipelined_source(s->ops, (int)s->model->n, (int)s->nobs,
                                                                       c.T, D, LA, c.halves, smode)
                                  : rt_jit_mfma_split_source(s->ops, (int)s->model->n, (int)s->nobs, c.T, D,
                                                             LA, smode);
                };
                if (!forced && opt_jit_async(s->model->ctx) && sites_jit_start_async(s, cands, make))
                    return RT_OK;
                for (const auto &c : cands) {
                    const std::string src = make(c);
                    if (src.empty()) continue;
                    rc = rt_jit_get(s->model->ctx, src, &s->jit_fn, true, &s->jit_compile_s);
                    if (rc == RT_OK) {
                        s->jit_tiles = c.T;
                        s->jit_waves = (int)((s->model->n + 15) / 16);
                        s->jit_sparse = true;
                        s->jit_pipe = c.pipe;
                        if (c.halves) RT_TRY(sites_halves_setup(s));
                        return RT_OK;
                    }
                    if (rc != RT_ERR_UNSUPPORTED) break;
                }
                rc = RT_ERR_UNSUPPORTED;       // spilled: the dense kernels below
            }
            // fewer tiles if it spills: halves at T, halves at one tile, then the whole tree.
            // Unless the kernel is already in this context's cache, a background job works
            // through that list while the batch runs the interpreter kernel.
            if (!forced && opt_jit_async(s->model->ctx)) {
                std::vector<rt_sites::jit_cand> cands;
                for (int t = T, h = halves;;) {
                    cands.push_back({t, h != 0, false});
                    if (h && t > 1) t = 1;
                    else if (h) h = 0;
                    else if (t > 1) --t;
                    else break;
                }
                if (sites_jit_start_async(s, cands, [&](const rt_sites::jit_cand &c) {
                        return split_source(s->ops, (int)s->model->n, (int)s->nobs, c.T, D, LA, c.halves);
                    }))
                    return RT_OK;
            }
            while (rc == RT_ERR_UNSUPPORTED) {
                const std::string src =
                    split_source(s->ops, (int)s->model->n, (int)s->nobs, T, D, LA, halves);
                rc = rt_jit_get(s->model->ctx, src, &s->jit_fn, true, &s->jit_compile_s);
                if (rc != RT_ERR_UNSUPPORTED) break;
                if (halves && T > 1) T = 1;
                else if (halves) halves = false;
                else if (T > 1) --T;
                else break;
            }
            if (rc != RT_OK && (!forced || rc == RT_ERR_UNSUPPORTED)) {
                s->jit_fn = nullptr;           // the interpreter kernel runs
                return RT_OK;
            }
            if (rc == RT_OK) {
                s->jit_tiles = T;
                s->jit_waves = (int)((s->model->n + 15) / 16);
                if (halves) rc = sites_halves_setup(s);
            }
            return rc;
        }
        const int64_t ntiles = (s->nsites + 15) / 16;
        const int KS = (int)((s->model->n + 3) / 4), NT = (int)((s->model->n + 15) / 16);
        // tiles per wave: as few waves as SIMDs (1 024) when the batch allows it,
        // within the register file (pending accumulators: KS doubles per lane, tile
        // and level)
        int T = (int)std::min<int64_t>(4, std::max<int64_t>(1, (ntiles + 1023) / 1024));
        auto regs = [&](int t) {
            return 2 * (s->model->max_depth * t * KS + 2 * NT * KS + 3 * t * KS + 4 * t * NT) + 48;
        };
        // up to two tiles: two waves per SIMD, 256 registers each; more: the whole file
        while (T > 1 && (regs(T) > (T <= 2 ? 240 : 480) || (int64_t)s->ops.size() * T > 600)) --T;
        if (const char *v = getenv("RAOTEH_JIT_TILES")) T = std::min(4, std::max(1, atoi(v)));
        int D = 2, LA = 1;
        if (const char *v = getenv("RAOTEH_JIT_PREFETCH")) D = std::max(1, atoi(v));
        if (const char *v = getenv("RAOTEH_JIT_LOOKAHEAD")) LA = std::max(1, atoi(v));
        s->jit_prefetch = D;
        s->jit_lookahead = LA;
        // built on v_mfma_f64_4x4x4_4b (4 rows x 16 sites per instruction: no padding of
        // n to a multiple of 16 rows) unless RAOTEH_JIT_QUAD=0
        // (RAOTEH_JIT_QUAD=0 keeps the 16x16x4 form.  Config 5: 71 us against 84 us at T = 4
        // with the blocks of P_e passing through LDS; with every block fetched replicated
        // from global memory the 4x4x4 form gained nothing)
        // ... where it saves matrix-pipe time: ceil(n/4)^2 x 16.5 cycles against
        // ceil(n/16) ceil(n/4) x 67 (none at n = 16 or 29..32; RAOTEH_JIT_QUAD=1 forces it)
        bool quad = s->model->d_Pquad != nullptr && KS * KS * 16.5 <= 0.8 * NT * KS * 67.0;
        if (const char *v = getenv("RAOTEH_JIT_QUAD")) quad = s->model->d_Pquad && atoi(v) != 0;
        int rc = RT_ERR_UNSUPPORTED;
        // A batch a little over a whole number q of tiles per SIMD (C5: 3 125 tiles on 1 024
        // SIMDs = 3.05) leaves, with q + 1 tiles per wave, a quarter of the SIMDs without a wave
        // and the others with a tile too many.  Then: q tiles per wave on every SIMD and the few
        // tiles over as one-tile waves of a second kernel on a side stream, next to them.
        // MEASURED, AND NOT THE DEFAULT (RAOTEH_JIT_SPLIT2=1 turns it on): the two kernels do
        // run side by side (rocprofv3: main 0..64 us, tail 7..65 us, against 67 us for the
        // single T = 4 kernel), but the one-tile waves crawl next to the three-tile ones (58 us
        // instead of 26 alone), the main kernel slows from 57 to 64 us, and the fork / join
        // events between the streams cost more than the 2 us gained: the step goes from 80 to
        // 95 us.  Only without an explicit RAOTEH_JIT_TILES.
        {
            const int64_t nsimd = 4 * (int64_t)s->model->ctx->num_cus;
            const int64_t q = nsimd > 0 ? ntiles / nsimd : 0, r = nsimd > 0 ? ntiles - q * nsimd : 0;
            const char *sp = getenv("RAOTEH_JIT_SPLIT2");
            if (!getenv("RAOTEH_JIT_TILES") && sp && atoi(sp) != 0 && q >= 1 && q <= 3 && r > 0 &&
                r <= nsimd / 4 && T == (int)q + 1) {
                void *fn_main = nullptr, *fn_tail = nullptr;
                double cs1 = 0.0, cs2 = 0.0;
                const std::string src_main =
                    rt_jit_mfma_source(s->ops, (int)s->model->n, (int)s->nobs, (int)q, D, LA, quad);
                int rc2 = rt_jit_get(s->model->ctx, src_main, &fn_main, true, &cs1);
                if (rc2 == RT_OK) {
                    const std::string src_tail =
                        rt_jit_mfma_source(s->ops, (int)s->model->n, (int)s->nobs, 1, D, LA, quad);
                    rc2 = rt_jit_get(s->model->ctx, src_tail, &fn_tail, true, &cs2);
                    if (rc2 != RT_OK) rt_jit_ref(s->model->ctx, fn_main, -1);
                }
                if (rc2 == RT_OK) {
                    s->jit_fn = fn_main;
                    s->jit_fn2 = fn_tail;
                    s->jit_tiles = (int)q;
                    s->jit_tiles2 = 1;
                    s->jit_split_tiles = q * nsimd;
                    s->jit_compile_s = cs1 + cs2;
                    s->jit_quad = quad;
                    return RT_OK;
                }
            }
        }
        // observed states (or allowed sets of one or two) at every leaf, 4x4x4 form: the leaf
        // steps read their columns from the parked blocks (first in the list; then the dense forms)
        const int smode = s->sparse_ok && quad ? (s->sparse_pairs ? 2 : 1) : 0;
        if (!forced && opt_jit_async(s->model->ctx)) {
            std::vector<rt_sites::jit_cand> cands;
            if (smode)
                for (int t = T; t >= 1; --t) cands.push_back({t, false, quad, true});
            for (int t = T; t >= 1; --t) cands.push_back({t, false, quad});
            if (sites_jit_start_async(s, cands, [&](const rt_sites::jit_cand &c) {
                    return rt_jit_mfma_source(s->ops, (int)s->model->n, (int)s->nobs, c.T, D, LA, c.quad,
                                              c.sparse ? smode : 0);
                }))
                return RT_OK;
        }
        if (smode) {
            for (int t = T; t >= 1 && rc == RT_ERR_UNSUPPORTED; --t) {
                const std::string src = rt_jit_mfma_source(s->ops, (int)s->model->n, (int)s->nobs, t, D, LA,
                                                           quad, smode);
                rc = rt_jit_get(s->model->ctx, src, &s->jit_fn, true, &s->jit_compile_s);
                if (rc == RT_OK) {
                    s->jit_quad = quad;
                    s->jit_tiles = t;
                    s->jit_sparse = true;
                    return RT_OK;
                }
            }
            rc = RT_ERR_UNSUPPORTED;           // spilled at every T: the dense forms
        }
        for (; T >= 1 && rc == RT_ERR_UNSUPPORTED; --T) {         // fewer tiles if it spills
            const std::string src =
                rt_jit_mfma_source(s->ops, (int)s->model->n, (int)s->nobs, T, D, LA, quad);
            rc = rt_jit_get(s->model->ctx, src, &s->jit_fn, true, &s->jit_compile_s);
        }
        ++T;
        if (rc == RT_OK) s->jit_quad = quad;
        if (rc != RT_OK && (!forced || rc == RT_ERR_UNSUPPORTED)) {
            s->jit_fn = nullptr;               // the interpreter kernel runs
            return RT_OK;
        }
        if (rc == RT_OK) s->jit_tiles = T;
        return rc;
    }
    // measured on C2: 6 stream positions and 2 P records ahead (tools/ab_jit.sh)
    int D = 6;
    if (const char *v = getenv("RAOTEH_JIT_PREFETCH")) D = std::max(1, atoi(v));
    int LA = 2;
    if (const char *v = getenv("RAOTEH_JIT_LOOKAHEAD")) LA = std::max(1, atoi(v));
    // Sites per wave S (at most 8 waves per CU: VGPRs).  A batch that fits the chip
    // in one round (<= 2 048 blocks of 64) is cut into a multiple of 256 waves of
    // S <= 64 active lanes, so that every CU streams the same number of bytes (C2:
    // 1 563 blocks of 64 would be 6 or 7 waves per CU; 1 792 waves of 56 sites are
    // 7 everywhere: 41 -> 39 us).  Waves per workgroup: 1 (each wave stages its own
    // copy of the P table; sharing one per CU, RAOTEH_JIT_WAVES=7, measured the same).
    {
        // registers: pending accumulators + P records in flight + leaf vectors in flight
        const int64_t n = s->model->n, np = (n + 1) & ~1ll;
        const int64_t regs = 2 * (s->model->max_depth * n + (1 + LA) * n * n + (D + 1) * np) + 40;
        if (!forced && regs > 250) return RT_OK;    // would spill: the interpreter is faster
    }
    // uint8 states and (n <= 4) allowed-set masks stay one byte per leaf on the device
    // (64 B per site instead of 2 KB) when the specialised kernel runs them;
    // RAOTEH_JIT_DENSE_STATES=1 expands them as before
    const int states = (s->nobs > 0 && s->nobs <= 1024 && !getenv("RAOTEH_JIT_DENSE_STATES"))
                           ? (kind == RT_OBS_STATE ? 1 : kind == RT_OBS_MASK ? 2 : 0) : 0;
    // LDS tables of a workgroup: the step-ordered P table and, for state batches, the
    // column table of the observed leaves (jit.hip).  Each wave keeps its own copy
    // while the (up to 8) waves of a CU fit the 160 KB that way; else the waves of a
    // CU form one workgroup (balanced batches) or workgroups of 2, 4, 8 waves share.
    int S = 64, WG = 1;
    int64_t tables = (int64_t)s->ops.size() * s->model->n * s->model->n * 8;
    if (states)
        for (const rt_op &op : s->ops)
            if (op.pop < 0 && op.obs >= 0 && op.dst >= 0)
                tables += s->model->n * (states == 2 ? (1 << s->model->n) : s->model->n + 1) * 8;
    if (tables > 150 * 1024) return RT_OK;
    const int64_t nb64 = (s->nsites + 63) / 64;
    if (nb64 >= 256 && nb64 <= 2048) {
        const int64_t nw = (nb64 + 255) / 256 * 256;
        S = (int)((s->nsites + nw - 1) / nw);
        const int64_t per_cu = nw / 256;
        if (per_cu * tables > 150 * 1024) WG = (int)per_cu;     // one workgroup per CU
    } else {
        while (WG < 8 && (8 / WG) * tables > 150 * 1024) WG *= 2;
    }
    if (opt_jit_block_sites(s->model->ctx) > 0) S = opt_jit_block_sites(s->model->ctx);
    if (const char *v = getenv("RAOTEH_JIT_BLOCK_SITES")) S = atoi(v);
    if (const char *v = getenv("RAOTEH_JIT_WAVES")) WG = atoi(v);
    S = std::min(64, std::max(1, S));
    WG = std::min(8, std::max(1, WG));
    // RAOTEH_JIT_FUSE_EXPM=1: the kernel that can run a whole step in ONE launch (jit.hip).
    // Correct (test_single_launch_step_is_the_two_launch_step) and NOT the default: measured on
    // config 2 the fused kernel takes 39.3 us where the plain one takes 33.0 and the expm
    // launch 4.8 -- every workgroup repeats the ~6 us of load + dependent f64 latency of the
    // exponentials at its start, with its HBM stream idle -- and the step 41.8 us against 41.1.
    const bool fuse = getenv("RAOTEH_JIT_FUSE_EXPM") && atoi(getenv("RAOTEH_JIT_FUSE_EXPM")) != 0;
    const std::string src =
        rt_jit_lane_source(s->ops, (int)s->model->n, (int)s->nobs, D, LA, S, WG, states, fuse);
    s->jit_prefetch = D;
    s->jit_lookahead = LA;
    // not in this context's cache yet: compile in the background; the batch is created in the
    // interpreter's layout, keeps the caller's observations on the device and is packed again
    // for the kernel when it arrives (rt_sites_jit_poll)
    const bool background = !forced && opt_jit_async(s->model->ctx) &&
                            rt_jit_cached(s->model->ctx, src) == 0;
    if (background && jit_jobs_full()) return RT_OK;    // the interpreter kernel, no new thread
    if (background) {
        s->jit_lane.S = S;
        s->jit_lane.WG = WG;
        s->jit_lane.D = D;
        s->jit_lane.LA = LA;
        s->jit_lane.compact = states;
        s->jit_lane.fuse = fuse;
        s->jit_srcs.assign(1, src);
        s->jit_cands.assign(1, rt_sites::jit_cand{1, false, false});
        s->jit_job = rt_jit_start(s->model->ctx, std::vector<std::string>(1, src), false);
        s->keep_raw = true;
        return RT_OK;
    }
    const int rc = rt_jit_get(s->model->ctx, src, &s->jit_fn, false, &s->jit_compile_s);
    if (rc != RT_OK && (!forced || rc == RT_ERR_UNSUPPORTED)) {
        // the interpreter kernel (prune.hip) computes the same numbers;
        // rt_last_error() keeps the compiler's message / the rejection
        s->jit_fn = nullptr;
        return RT_OK;
    }
    if (rc == RT_OK) {
        s->block_sites = S;
        s->jit_waves = WG;
        s->compact_states = states;
        s->jit_fused = fuse;
    }
    return rc;
}

extern "C" int rt_jit_source(int64_t nnodes, const int64_t *idx, const int64_t *ptr,
                             int64_t n, int64_t nobs, const int64_t *obs_nodes,
                             int64_t prefetch, char *buf, int64_t capacity)
{
    RT_REQUIRE(nnodes >= 1 && ptr && (idx || nnodes == 1) && buf && capacity > 0,
               "bad arguments");
    RT_REQUIRE(n >= 1 && n <= RT_MAX_STATES, "tree-specialised kernels exist for n <= %d",
               RT_MAX_STATES);
    rt_model m;
    m.nnodes = nnodes;
    if (nnodes > 1) m.indices.assign(idx, idx + (nnodes - 1));
    m.indptr.assign(ptr, ptr + nnodes + 1);
    build_schedule(&m);
    std::vector<int32_t> node_obs((size_t)nnodes, -1);
    for (int64_t j = 0; j < nobs; ++j) {
        RT_REQUIRE(obs_nodes[j] >= 0 && obs_nodes[j] < nnodes, "obs_nodes out of range");
        node_obs[(size_t)obs_nodes[j]] = (int32_t)j;
    }
    int32_t k = 0;
    for (auto &op : m.ops)
        if (node_obs[(size_t)op.node] >= 0) op.obs = k++;
    const int LA = getenv("RAOTEH_JIT_LOOKAHEAD") ? std::max(1, atoi(getenv("RAOTEH_JIT_LOOKAHEAD"))) : 2;
    const int T = getenv("RAOTEH_JIT_TILES") ? std::min(5, std::max(1, atoi(getenv("RAOTEH_JIT_TILES")))) : 2;
    const std::string src = n <= 4
        ? rt_jit_lane_source(m.ops, (int)n, (int)nobs, (int)prefetch, LA, 64, 4,
                             getenv("RAOTEH_JIT_SOURCE_STATES") != nullptr,
                             getenv("RAOTEH_JIT_FUSE_EXPM") && atoi(getenv("RAOTEH_JIT_FUSE_EXPM")) != 0)
        : n <= 32 ? rt_jit_mfma_source(m.ops, (int)n, (int)nobs, std::min(T, 4), (int)prefetch, 1,
                                       !(getenv("RAOTEH_JIT_QUAD") && atoi(getenv("RAOTEH_JIT_QUAD")) == 0),
                                       !getenv("RAOTEH_JIT_SOURCE_SPARSE") ? 0
                                       : strcmp(getenv("RAOTEH_JIT_SOURCE_SPARSE"), "2") == 0 ? 2 : 1)
                  : getenv("RAOTEH_JIT_SOURCE_SPARSE") && strncmp(getenv("RAOTEH_JIT_SOURCE_SPARSE"), "pipe", 4) == 0
                        ? rt_jit_mfma_split_pipelined_source(
                              m.ops, (int)n, (int)nobs, T, 2, 1,
                              getenv("RAOTEH_JIT_HALVES") && atoi(getenv("RAOTEH_JIT_HALVES")),
                              strcmp(getenv("RAOTEH_JIT_SOURCE_SPARSE"), "pipe2") == 0 ? 2 : 1)
                  : getenv("RAOTEH_JIT_SOURCE_SPARSE")
                        ? rt_jit_mfma_split_source(m.ops, (int)n, (int)nobs, std::min(T, 2), 2, 1, 1)
                  : split_source(m.ops, (int)n, (int)nobs,
                                 getenv("RAOTEH_JIT_HALVES") && atoi(getenv("RAOTEH_JIT_HALVES"))
                                     ? T : std::min(T, 3), (int)prefetch, 1,
                                 getenv("RAOTEH_JIT_HALVES") && atoi(getenv("RAOTEH_JIT_HALVES")));
    RT_REQUIRE((int64_t)src.size() + 1 <= capacity, "buffer too small: %lld bytes needed",
               (long long)src.size() + 1);
    memcpy(buf, src.c_str(), src.size() + 1);
    return RT_OK;
}

static int verify_jit_kernel(rt_sites *s, int kind);

static int sites_create_impl(rt_model *m, int64_t nsites, int kind, int64_t nobs,
                             const int64_t *obs_nodes, const void *data,
                             const jit_override *ov, rt_sites **out)
{
    RT_REQUIRE(m && out, "null pointer");
    *out = nullptr;
    RT_REQUIRE(nsites >= 1, "nsites must be >= 1");
    RT_REQUIRE(kind == RT_OBS_DENSE || kind == RT_OBS_STATE || kind == RT_OBS_MASK,
               "unknown observation kind %d", kind);
    RT_REQUIRE(nobs >= 0 && nobs <= m->nnodes, "bad nobs");
    RT_REQUIRE(nobs == 0 || (obs_nodes && data), "null observation arrays");
    RT_REQUIRE(kind != RT_OBS_STATE || m->n <= 255, "uint8 states need n <= 255");
    RT_HIP(hipSetDevice(m->ctx->device));
    rt_sites *s = new (std::nothrow) rt_sites();
    if (!s) return RT_ERR_NOMEM;
    s->model = m;
    s->nsites = nsites;
    s->nobs = nobs;
    s->jit_kind = kind;
    s->node_obs.assign((size_t)m->nnodes, -1);
    for (int64_t j = 0; j < nobs; ++j) {
        const int64_t v = obs_nodes[j];
        if (v < 0 || v >= m->nnodes || s->node_obs[(size_t)v] >= 0) {
            rt_set_error("obs_nodes[%lld]=%lld out of range or repeated", (long long)j,
                         (long long)v);
            delete s;
            return RT_ERR_INVALID;
        }
        s->node_obs[(size_t)v] = (int32_t)j;
    }
    // observation stream = observed nodes in schedule order
    s->ops = m->ops;
    std::vector<int64_t> src_of_k;
    for (auto &op : s->ops) {
        const int32_t j = s->node_obs[(size_t)op.node];
        if (j >= 0) {
            op.obs = (int32_t)src_of_k.size();
            src_of_k.push_back(j);
        }
    }
    // the split-M kernels keep NT * 2 KB of LDS per accumulator slot: above 64 states a tree
    // that needs more slots than the CU's 160 KB hold runs the generic kernel
    const int64_t nt_waves = std::max<int64_t>(4, (m->n + 15) / 16);
    const bool generic = opt_force_generic(m->ctx) || m->max_depth > RT_FAST_MAX_DEPTH ||
                         (m->n > 64 && nt_waves * m->max_depth * 2048 + 20 * 1024 > 160 * 1024);
    s->layout = (generic || m->n <= 4) ? RT_LAYOUT_LANE : RT_LAYOUT_MFMA;
    // rescaling lives in the interpreter kernels: no tree-specialised kernel for such a batch,
    // and the lane family's VGPR-ring variant (the LDS-DMA variant fuses cherries)
    s->rescale = opt_rescale(m->ctx) != 0 && !ov;
    // tuning knobs of the lane family (A/B measurements)
    // Default: leaf vectors through the LDS-DMA ring (3 slots) when two 4-wave
    // workgroups (shared P table + rings + accumulator stacks) fit on a CU;
    // otherwise the VGPR-ring variant, whose LDS need is only the stacks.
    if (s->layout == RT_LAYOUT_LANE && !generic) {
        const int64_t np = (m->n + 1) & ~1ll;
        const int64_t stack = std::max<int64_t>(1, m->max_depth - 1) * m->n * 512;
        const int64_t ptab = ((int64_t)(m->ops.size() + 1) * m->n * m->n * 8 + 15) & ~15ll;
        s->lane_dma = ptab + 4 * (3 * 64 * np * 8 + stack) <= 80 * 1024;
    }
    s->mfma_solo = s->layout == RT_LAYOUT_MFMA && m->n <= 32 && !getenv("RAOTEH_MFMA_NO_SOLO");
    if (ov && ov->no_solo) s->mfma_solo = false;
    if (const char *v = getenv("RAOTEH_LANE_VARIANT")) s->lane_dma = strcmp(v, "dma") == 0;
    if (s->rescale) s->lane_dma = false;
    if (const char *r = getenv("RAOTEH_LANE_RING")) s->lane_ring = atoi(r);
    else s->lane_ring = s->lane_dma ? 0 : 8;      // 0: rt_launch_prune picks what fits
    // observed STATES, all at leaves, none unobserved, split-M family: the kernel's leaf steps
    // may gather columns of P (jit.hip, `sparse`); the dense image stays (interpreter kernel)
    if (kind == RT_OBS_STATE && s->layout == RT_LAYOUT_MFMA && (!s->mfma_solo || m->d_Pquad) && !generic &&
        nobs > 0 &&
        !s->rescale && opt_leaf_state_kernels(m->ctx) && !getenv("RAOTEH_JIT_NO_SPARSE") &&
        (!ov || ov->sparse)) {
        bool ok = true;
        for (const rt_op &op : s->ops)
            if (op.obs >= 0 && !(op.pop < 0 && op.dst >= 0)) ok = false;
        const unsigned char *bytes = (const unsigned char *)data;
        const size_t count = (size_t)nsites * (size_t)nobs;
        for (size_t i = 0; ok && i < count; ++i) ok = bytes[i] < m->n;
        s->sparse_ok = ok;
    }
    // ... and allowed SETS of one or two states at every leaf (the compound models: a codon in
    // either class of the switching model, liwen.py:682): the leaf's message is one column of P
    // or the sum of two
    if (kind == RT_OBS_MASK && s->layout == RT_LAYOUT_MFMA && (!s->mfma_solo || m->d_Pquad) && !generic &&
        nobs > 0 &&
        !s->rescale && opt_leaf_state_kernels(m->ctx) && !getenv("RAOTEH_JIT_NO_SPARSE") &&
        (!ov || ov->sparse)) {
        bool ok = true;
        for (const rt_op &op : s->ops)
            if (op.obs >= 0 && !(op.pop < 0 && op.dst >= 0)) ok = false;
        const uint64_t *words = (const uint64_t *)data;
        const size_t nw = (size_t)((m->n + 63) / 64);
        const size_t count = (size_t)nsites * (size_t)nobs;
        const uint64_t top = (m->n & 63) ? ((1ull << (m->n & 63)) - 1) : ~0ull;
        for (size_t i = 0; ok && i < count; ++i) {
            int bits = 0;
            for (size_t w = 0; w < nw; ++w) {
                const uint64_t v = words[i * nw + w] & (w + 1 == nw ? top : ~0ull);
                bits += __builtin_popcountll(v);
            }
            ok = bits == 1 || bits == 2;
        }
        s->sparse_ok = ok;
        s->sparse_pairs = ok;
    }
    // (the split-M leaf-state kernels read the columns from the model's leaf-column table)
    if (s->sparse_ok && !s->mfma_solo) RT_TRY(rt_model_need_pcol(m));
    int rc = sites_jit(s, generic || s->rescale, kind, ov);    // before the layout is fixed: block_sites
    // a freshly compiled kernel is checked against the interpreter kernel on a probe batch
    // before any user batch may launch it; if it fails this batch runs the interpreter
    // (RAOTEH_JIT_NO_VERIFY: diagnostics only, tests/soak/spill_probe.py)
    if (rc == RT_OK && s->jit_fn2 && !ov && !getenv("RAOTEH_JIT_NO_VERIFY")) {
        // two-kernel batch: each kernel is checked as the single kernel of a probe batch
        void *fn_main = s->jit_fn, *fn_tail = s->jit_fn2;
        const int t_main = s->jit_tiles;
        bool good = true;
        for (int which = 0; which < 2 && good; ++which) {
            s->jit_fn = which ? fn_tail : fn_main;
            s->jit_tiles = which ? s->jit_tiles2 : t_main;
            s->jit_fn2 = nullptr;
            if (!rt_jit_verified(m->ctx, s->jit_fn)) {
                const int vrc = verify_jit_kernel(s, kind);
                rt_jit_set_verified(m->ctx, s->jit_fn, vrc == RT_OK);
                good = vrc == RT_OK;
            }
        }
        s->jit_fn = fn_main;
        s->jit_fn2 = fn_tail;
        s->jit_tiles = t_main;
        if (!good) {
            rt_jit_ref(m->ctx, fn_main, -1);
            rt_jit_ref(m->ctx, fn_tail, -1);
            s->jit_fn = s->jit_fn2 = nullptr;
            s->jit_split_tiles = 0;
            s->jit_tiles = 1;
            s->jit_quad = false;
        }
    } else if (rc == RT_OK && s->jit_fn && !ov && !rt_jit_verified(m->ctx, s->jit_fn) &&
        !getenv("RAOTEH_JIT_NO_VERIFY")) {
        const int vrc = verify_jit_kernel(s, kind);
        rt_jit_set_verified(m->ctx, s->jit_fn, vrc == RT_OK);
        // (RAOTEH_JIT_VERIFY_STRICT=1, soak runs: a rejected kernel is a generator or compiler
        // defect to look at, not something to fall back from silently)
        if (vrc != RT_OK && getenv("RAOTEH_JIT_VERIFY_STRICT")) {
            fprintf(stderr, "[raoteh_amd] probe verification rejected %s\n", s->kernel_name);
            rc = RT_ERR_INVALID;
        }
        if (vrc != RT_OK) {
            rt_jit_ref(m->ctx, s->jit_fn, -1);
            s->jit_fn = nullptr;
            s->block_sites = 64;
            s->jit_waves = 1;
            s->jit_tiles = 1;
            s->compact_states = 0;
            s->jit_quad = false;
            s->jit_sparse = false;
            s->jit_pipe = false;
            s->jit_halves = false;
            s->jit_fold = false;
            s->jit_combine = nullptr;
            s->jit_fused = false;
        }
    }
    if (rc == RT_OK) rc = sites_alloc(s, generic);
    if (rc == RT_OK) rc = rt_sites_pack(s, kind, src_of_k.data(), data);
    if (rc != RT_OK) {
        rt_sites_destroy(s);
        return rc;
    }
    s->counted = true;
    m->live_batches += 1;
    *out = s;
    return RT_OK;
}

int rt_sites_create_interpreter(rt_model *m, int64_t nsites, int kind, int64_t nobs,
                                const int64_t *obs_nodes, const void *data, rt_sites **out)
{
    jit_override interp;
    interp.mode = 1;
    interp.no_solo = true;        // the kernel that can leave L and M of every step behind
    return sites_create_impl(m, nsites, kind, nobs, obs_nodes, data, &interp, out);
}

extern "C" int rt_sites_create(rt_model *m, int64_t nsites, int kind, int64_t nobs,
                               const int64_t *obs_nodes, const void *data,
                               rt_sites **out)
{
    return sites_create_impl(m, nsites, kind, nobs, obs_nodes, data, nullptr, out);
}

// A tree-specialised kernel is compiled at run time by a compiler this library does not
// control, with the whole register file in use.  Round 1's randomised soak found one that
// returned wrong log-likelihoods (31 / 32 states, 4 site tiles per wave); the cause, found
// in round 2 from the ISA of that kernel (tools/jit_offline.py, tests/soak/spill_probe.py),
// is a register-allocation defect of ROCm 7.2's AMDGPU backend: a 64-bit value living in an
// AGPR pair (a root weight, live through the whole walk) is split when the file is full --
// low half to a scratch slot, high half copied to a VGPR that is then treated as dead -- and
// the reload restores the low half only.  Nothing in the source is wrong and a kernel that
// spills far more (256 registers) is right, so "it has scratch" is a proxy, not the defect.
// Hence this check: every freshly compiled kernel runs once on a probe batch (random
// transition matrices, root weights and observations on the batch's own tree) next to the
// interpreter kernel, and is used only if all log-likelihoods and statuses agree bit for
// bit -- which is what correct code guarantees (same arithmetic order).  ~1 ms per compile.
static int verify_jit_kernel(rt_sites *s, int kind)
{
    rt_model *m = s->model;
    const int64_t n = m->n, N = m->nnodes, K = s->nobs;
    const bool mfma = s->layout == RT_LAYOUT_MFMA;
    // three full blocks / tile groups of the kernel and a ragged tail
    const int64_t per = mfma ? 16 * s->jit_tiles : s->block_sites * s->jit_waves;
    const int64_t np = 3 * per + 5;
    uint64_t state = 0x9E3779B97F4A7C15ull ^ (uint64_t)(N * 1315423911u + n);
    auto next = [&]() {            // splitmix64 -> (0, 1)
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    };
    std::vector<double> esd((size_t)N * n * n), root((size_t)n);
    for (double &v : esd) v = 0.02 + next();
    for (double &v : root) v = 0.1 + next();
    std::vector<int64_t> obs_nodes((size_t)K);
    for (int64_t v = 0; v < N; ++v)
        if (s->node_obs[(size_t)v] >= 0) obs_nodes[(size_t)s->node_obs[(size_t)v]] = v;
    // observations of the kind the kernel was specialised for (compact kernels read bytes)
    const int pkind = s->jit_sparse && s->sparse_pairs ? RT_OBS_MASK
                    : (s->compact_states == 1 || s->jit_sparse) ? RT_OBS_STATE
                    : s->compact_states == 2 ? RT_OBS_MASK : RT_OBS_DENSE;
    (void)kind;
    std::vector<double> dense;
    std::vector<unsigned char> states;
    std::vector<uint64_t> masks;
    const void *data = nullptr;
    if (pkind == RT_OBS_DENSE) {
        dense.resize((size_t)np * K * n);
        for (double &v : dense) {
            const double u = next();
            v = u < 0.15 ? 0.0 : u;
        }
        data = dense.data();
    } else if (pkind == RT_OBS_STATE) {
        states.resize((size_t)np * K);
        for (unsigned char &v : states) {
            const double u = next();
            // (the column-gathering kernels take batches without unobserved leaves only)
            v = (u < 0.15 && !s->jit_sparse) ? 255 : (unsigned char)((int)(next() * n) % (int)n);
        }
        data = states.data();
    } else if (s->jit_sparse && s->sparse_pairs) {
        // one or two allowed states per leaf (what the pair-gathering kernels take)
        const size_t nw = (size_t)((n + 63) / 64);
        masks.assign((size_t)np * K * nw, 0);
        for (size_t i = 0; i < (size_t)np * K; ++i) {
            const int a1 = (int)(next() * n) % (int)n, b1 = (int)(next() * n) % (int)n;
            masks[i * nw + (size_t)(a1 >> 6)] |= 1ull << (a1 & 63);
            if (next() < 0.8) masks[i * nw + (size_t)(b1 >> 6)] |= 1ull << (b1 & 63);
        }
        data = masks.data();
    } else {
        masks.resize((size_t)np * K);
        for (uint64_t &v : masks) v = 1 + (uint64_t)(next() * ((1ull << n) - 1));
        data = masks.data();
    }
    rt_model *tm = nullptr;
    rt_sites *si = nullptr, *sj = nullptr;
    int rc = rt_model_create(m->ctx, N, n, m->indices.data(), m->indptr.data(), &tm);
    if (rc == RT_OK) rc = rt_model_set_transitions(tm, esd.data());
    if (rc == RT_OK) rc = rt_model_set_root_distn(tm, root.data());
    jit_override interp, same;
    interp.mode = 1;
    same.mode = 2;
    same.T = s->jit_tiles;
    same.S = s->block_sites;
    same.WG = s->jit_waves;
    same.D = s->jit_prefetch;
    same.LA = s->jit_lookahead;
    same.compact = s->compact_states;
    same.quad = s->jit_quad;
    same.halves = s->jit_halves;
    same.fuse = s->jit_fused;
    same.sparse = s->jit_sparse;
    same.pipe = s->jit_pipe;
    if (rc == RT_OK)
        rc = sites_create_impl(tm, np, pkind, K, obs_nodes.data(), data, &interp, &si);
    if (rc == RT_OK)
        rc = sites_create_impl(tm, np, pkind, K, obs_nodes.data(), data, &same, &sj);
    if (rc == RT_OK && sj->jit_fn != s->jit_fn) {
        rt_set_error("probe batch did not get the kernel under test");
        rc = RT_ERR_UNSUPPORTED;
    }
    std::vector<double> li((size_t)np), lj((size_t)np);
    std::vector<int32_t> sti((size_t)np), stj((size_t)np);
    // no event timing for the probe launches (they are not the caller's)
    const bool timing = m->ctx->timing;
    m->ctx->timing = false;
    if (rc == RT_OK) rc = rt_prune(tm, si);
    if (rc == RT_OK) rc = rt_prune(tm, sj);
    m->ctx->timing = timing;
    if (rc == RT_OK) rc = rt_sites_get_logliks(si, li.data(), sti.data());
    if (rc == RT_OK) rc = rt_sites_get_logliks(sj, lj.data(), stj.data());
    if (rc == RT_OK) {
        int64_t bad = 0;
        for (int64_t i = 0; i < np; ++i)
            bad += memcmp(&li[(size_t)i], &lj[(size_t)i], 8) != 0 || sti[(size_t)i] != stj[(size_t)i];
        if (bad) {
            rt_set_error("tree-specialised kernel rejected: %lld of %lld probe sites differ from "
                         "the interpreter kernel (miscompiled)", (long long)bad, (long long)np);
            rc = RT_ERR_UNSUPPORTED;
        }
    }
    rt_sites_destroy(si);
    rt_sites_destroy(sj);
    rt_model_destroy(tm);
    return rc;
}

static void sites_drop_raw(rt_sites *s)
{
    hipFree(s->d_raw);
    hipFree(s->d_raw_src);
    s->d_raw = nullptr;
    s->d_raw_src = nullptr;
    s->keep_raw = false;
}

// Lane family: the kernel is there.  Verify it (the probe batches take the kernel's parameters
// from the batch), then move the batch to the kernel's resident layout: new block size /
// compact encoding, the observations packed again from the retained device copy.
static int sites_lane_switch(rt_sites *s, const std::string &src)
{
    rt_model *m = s->model;
    rt_ctx *ctx = m->ctx;
    void *fn = nullptr;
    if (rt_jit_get(ctx, src, &fn, false, nullptr) != RT_OK) {
        sites_drop_raw(s);
        return RT_OK;
    }
    const int old_S = s->block_sites, old_waves = s->jit_waves, old_compact = s->compact_states;
    const int old_D = s->jit_prefetch, old_LA = s->jit_lookahead;
    s->jit_fn = fn;
    s->block_sites = s->jit_lane.S;
    s->jit_waves = s->jit_lane.WG;
    s->compact_states = s->jit_lane.compact;
    s->jit_fused = s->jit_lane.fuse;
    s->jit_prefetch = s->jit_lane.D;
    s->jit_lookahead = s->jit_lane.LA;
    int rc = RT_OK;
    if (!rt_jit_verified(ctx, fn) && !getenv("RAOTEH_JIT_NO_VERIFY")) {
        rc = verify_jit_kernel(s, s->jit_kind);
        rt_jit_set_verified(ctx, fn, rc == RT_OK);
    }
    if (rc == RT_OK && ctx->pending_reduce == s) rc = rt_flush_reduce(ctx);
    if (rc == RT_OK) rc = hipStreamSynchronize(ctx->stream) == hipSuccess ? RT_OK : RT_ERR_HIP;
    double *n_obs = nullptr, *n_ll = nullptr, *n_part = nullptr, *n_alt = nullptr;
    int32_t *n_st = nullptr;
    int64_t padded = 0;
    const int64_t old_nblocks = s->nblocks, old_bytes = s->obs_bytes, old_np = s->npartials;
    if (rc == RT_OK) {
        sites_layout_sizes(s, &padded);
        hipError_t e = hipMalloc((void **)&n_obs, std::max<int64_t>(s->obs_bytes, 1024));
        if (e == hipSuccess) e = hipMalloc((void **)&n_ll, padded * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&n_st, padded * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&n_part, s->npartials * 16);
        if (e == hipSuccess) e = hipMemset(n_part, 0, s->npartials * 16);
        if (e == hipSuccess && s->jit_fused) {
            e = hipMalloc((void **)&n_alt, s->npartials * 16);
            if (e == hipSuccess) e = hipMemset(n_alt, 0, s->npartials * 16);
        }
        if (e != hipSuccess) rc = RT_ERR_NOMEM;
    }
    if (rc == RT_OK) {
        std::swap(s->d_obs, n_obs);
        rc = rt_sites_pack_device(s, s->jit_kind, s->d_raw, s->d_raw_src);
        if (rc == RT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = RT_ERR_HIP;
        if (rc != RT_OK) std::swap(s->d_obs, n_obs);
    }
    if (rc != RT_OK) {
        // stay on the interpreter kernel and its layout
        hipFree(n_obs); hipFree(n_ll); hipFree(n_st); hipFree(n_part); hipFree(n_alt);
        rt_jit_ref(ctx, fn, -1);
        s->jit_fn = nullptr;
        s->block_sites = old_S;
        s->jit_waves = old_waves;
        s->compact_states = old_compact;
        s->jit_fused = false;
        s->jit_prefetch = old_D;
        s->jit_lookahead = old_LA;
        s->nblocks = old_nblocks;
        s->obs_bytes = old_bytes;
        s->npartials = old_np;
        sites_drop_raw(s);
        return RT_OK;
    }
    hipFree(n_obs);                          // (the interpreter's image, after the swap above)
    hipFree(s->d_loglik); hipFree(s->d_status); hipFree(s->d_partial); hipFree(s->d_partial_alt);
    s->d_loglik = n_ll;
    s->d_status = n_st;
    s->d_partial = n_part;
    s->d_partial_alt = n_alt;
    sites_drop_raw(s);
    return RT_OK;
}

// The background job of this batch is done (or `wait`: join it): take the kernel it left in
// the context's cache, verify it on a probe batch if nobody has yet, and from the next launch
// on the batch runs it.  Any failure leaves the batch on the interpreter kernel.
int rt_sites_jit_poll(rt_sites *s, bool wait)
{
    if (!s->jit_job) return RT_OK;
    if (!rt_jit_job_done(s->jit_job.get(), wait)) return RT_OK;
    std::shared_ptr<rt_jit_job> job;
    job.swap(s->jit_job);
    int rc = RT_OK, chosen = -1;
    double seconds = 0.0;
    std::string err;
    rt_jit_job_result(job.get(), &rc, &chosen, &seconds, &err);
    s->jit_compile_s = seconds;
    std::vector<rt_sites::jit_cand> cands;
    std::vector<std::string> srcs;
    cands.swap(s->jit_cands);
    srcs.swap(s->jit_srcs);
    if (rc != RT_OK || chosen < 0 || chosen >= (int)cands.size()) {
        rt_set_error("background compile: %s", err.c_str());
        sites_drop_raw(s);
        return RT_OK;
    }
    rt_model *m = s->model;
    RT_HIP(hipSetDevice(m->ctx->device));
    if (s->layout == RT_LAYOUT_LANE) return sites_lane_switch(s, srcs[(size_t)chosen]);
    void *fn = nullptr;
    if (rt_jit_get(m->ctx, srcs[(size_t)chosen], &fn, true, nullptr) != RT_OK) return RT_OK;
    const rt_sites::jit_cand c = cands[(size_t)chosen];
    const bool split = m->n > 32 || !s->mfma_solo;
    s->jit_fn = fn;
    s->jit_tiles = c.T;
    s->jit_sparse = c.sparse;
    s->jit_pipe = split && c.sparse && c.pipe;
    s->jit_quad = !split && c.quad;
    if (split) s->jit_waves = (int)((m->n + 15) / 16);
    int src = RT_OK;
    if (split && c.halves) src = sites_halves_setup(s);
    if (src == RT_OK && s->jit_halves && !s->d_half &&
        hipMalloc((void **)&s->d_half,
                  (size_t)(s->nblocks + 8) * 2 * ((m->n + 15) / 16) * 4 * 64 * 8) != hipSuccess)
        src = RT_ERR_NOMEM;
    if (src == RT_OK && s->jit_halves && !s->d_half_count) {
        if (hipMalloc((void **)&s->d_half_count, (size_t)(s->nblocks + 8) * 4) != hipSuccess ||
            hipMemset(s->d_half_count, 0, (size_t)(s->nblocks + 8) * 4) != hipSuccess)
            src = RT_ERR_NOMEM;
    }
    if (src == RT_OK && !rt_jit_verified(m->ctx, fn) && !getenv("RAOTEH_JIT_NO_VERIFY")) {
        src = verify_jit_kernel(s, s->jit_kind);
        rt_jit_set_verified(m->ctx, fn, src == RT_OK);
    }
    if (src != RT_OK) {
        rt_jit_ref(m->ctx, fn, -1);
        s->jit_fn = nullptr;
        s->jit_tiles = 1;
        s->jit_quad = false;
        s->jit_sparse = false;
        s->jit_pipe = false;
        s->jit_halves = false;
        s->jit_fold = false;
        s->jit_combine = nullptr;
        return RT_OK;
    }
    // the interpreter kernel and the specialised one leave their per-wave partial sums in
    // different places of d_partial: what the other one wrote must read as zero
    if (m->ctx->pending_reduce == s) RT_TRY(rt_flush_reduce(m->ctx));
    RT_HIP(hipMemsetAsync(s->d_partial, 0, (size_t)s->npartials * 16, m->ctx->stream));
    return RT_OK;
}

extern "C" int rt_sites_jit_wait(rt_sites *s)
{
    RT_REQUIRE(s, "null pointer");
    return rt_sites_jit_poll(s, true);
}

extern "C" int rt_sites_clone(rt_sites *src, rt_sites **out)
{
    RT_REQUIRE(src && out, "null pointer");
    *out = nullptr;
    RT_HIP(hipSetDevice(src->model->ctx->device));
    // (a lane-family batch changes its resident layout when its kernel arrives: wait for it
    // rather than cloning the image it is about to leave)
    if (src->jit_job && src->layout == RT_LAYOUT_LANE) RT_TRY(rt_sites_jit_poll(src, true));
    rt_sites *s = new (std::nothrow) rt_sites();
    if (!s) return RT_ERR_NOMEM;
    s->model = src->model;
    s->nsites = src->nsites;
    s->nobs = src->nobs;
    s->layout = src->layout;
    s->lane_dma = src->lane_dma;
    s->rescale = src->rescale;
    s->mfma_solo = src->mfma_solo;
    s->lane_ring = src->lane_ring;
    s->jit_fn = src->jit_fn;
    rt_jit_ref(src->model->ctx, s->jit_fn, +1);
    s->jit_fn2 = src->jit_fn2;
    if (s->jit_fn2) rt_jit_ref(src->model->ctx, s->jit_fn2, +1);
    s->jit_tiles2 = src->jit_tiles2;
    s->jit_split_tiles = src->jit_split_tiles;
    s->jit_prefetch = src->jit_prefetch;
    s->jit_lookahead = src->jit_lookahead;
    s->block_sites = src->block_sites;
    s->jit_waves = src->jit_waves;
    s->jit_tiles = src->jit_tiles;
    s->jit_quad = src->jit_quad;
    s->jit_halves = src->jit_halves;
    s->jit_fold = src->jit_fold;
    s->jit_combine = src->jit_combine;
    s->jit_fused = src->jit_fused;
    s->compact_states = src->compact_states;
    s->node_obs = src->node_obs;
    s->ops = src->ops;
    s->jit_job = src->jit_job;              // a pending background compile serves both
    s->jit_cands = src->jit_cands;
    s->jit_srcs = src->jit_srcs;
    s->jit_kind = src->jit_kind;
    s->jit_compile_s = src->jit_compile_s;
    s->sparse_ok = src->sparse_ok;
    s->sparse_pairs = src->sparse_pairs;
    s->jit_sparse = src->jit_sparse;
    s->jit_pipe = src->jit_pipe;
    int rc = sites_alloc(s, src->d_scratch != nullptr);
    if (rc == RT_OK && s->d_leafw && src->d_leafw &&
        hipMemcpyAsync(s->d_leafw, src->d_leafw,
                       (size_t)s->nblocks * (s->sparse_pairs ? (s->nobs + 1) / 2 : (s->nobs + 3) / 4) * 16 * 4,
                       hipMemcpyDeviceToDevice, src->model->ctx->stream) != hipSuccess)
        rc = RT_ERR_HIP;
    if (rc == RT_OK && s->obs_bytes > 0) {
        // on the library's stream: a device-to-device hipMemcpy returns before the copy
        // has run and the (non-blocking) stream of the kernels does not wait for the
        // null stream -- a launch right after the clone would read a half-copied batch
        hipError_t e = hipMemcpyAsync(s->d_obs, src->d_obs, s->obs_bytes, hipMemcpyDeviceToDevice,
                                      src->model->ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(src->model->ctx->stream);
        if (e != hipSuccess) {
            rt_set_error("rt_sites_clone: %s", hipGetErrorString(e));
            rc = RT_ERR_HIP;
        }
    }
    if (rc != RT_OK) {
        rt_sites_destroy(s);
        return rc;
    }
    s->counted = true;
    s->model->live_batches += 1;
    *out = s;
    return RT_OK;
}

// The batch rt_expect_step runs its passes on: the split-M interpreter kernel's program and
// outputs over the resident observations of `src` (MFMA layout: the same [tile][slot][k-pair]
// image whichever pruning kernel the batch itself runs).  Owned by `src`.
int rt_sites_twin_interpreter(rt_sites *src, rt_sites **out)
{
    *out = nullptr;
    RT_REQUIRE(src->layout == RT_LAYOUT_MFMA && !src->d_scratch,
               "the batch is not resident in the matrix-pipe layout");
    rt_sites *s = new (std::nothrow) rt_sites();
    if (!s) return RT_ERR_NOMEM;
    s->model = src->model;
    s->nsites = src->nsites;
    s->nobs = src->nobs;
    s->layout = RT_LAYOUT_MFMA;
    s->mfma_solo = false;
    s->node_obs = src->node_obs;
    s->ops = src->ops;
    s->jit_kind = src->jit_kind;
    s->obs_borrowed = true;
    s->d_obs = src->d_obs;
    const int rc = sites_alloc(s, false);
    if (rc != RT_OK) {
        rt_sites_destroy(s);
        return rc;
    }
    // (the expectation launch stores L and M of the whole tree: no root halves)
    s->interp_halves = false;
    *out = s;
    return RT_OK;
}

// per-site multiplicities of a resident batch (site patterns): f64[nsites], NULL = ones
extern "C" int rt_sites_set_weights(rt_sites *s, const double *weights)
{
    RT_REQUIRE(s, "null pointer");
    RT_HIP(hipSetDevice(s->model->ctx->device));
    RT_HIP(hipStreamSynchronize(s->model->ctx->stream));
    if (!weights) {
        hipFree(s->d_weights);
        s->d_weights = nullptr;
        return RT_OK;
    }
    if (!s->d_weights) RT_HIP(hipMalloc((void **)&s->d_weights, (size_t)s->nsites * 8));
    RT_HIP(hipMemcpy(s->d_weights, weights, (size_t)s->nsites * 8, hipMemcpyHostToDevice));
    return RT_OK;
}

extern "C" int64_t rt_sites_device_bytes(const rt_sites *s)
{
    return s ? s->obs_bytes : 0;
}

extern "C" double rt_sites_jit_compile_seconds(const rt_sites *s)
{
    return s ? s->jit_compile_s : 0.0;
}

// diagnostics (tools/trace_c3.py): a __device__ variable of the batch's compiled kernel
extern "C" int rt_debug_jit_global(rt_sites *s, const char *name, void *dst, int64_t bytes)
{
    RT_REQUIRE(s && name && dst && bytes > 0, "bad arguments");
    RT_REQUIRE(s->jit_fn, "the batch has no tree-specialised kernel");
    RT_HIP(hipSetDevice(s->model->ctx->device));
    RT_HIP(hipStreamSynchronize(s->model->ctx->stream));
    return rt_jit_read_global(s->model->ctx, s->jit_fn, name, dst, (size_t)bytes);
}

extern "C" const char *rt_sites_kernel_name(const rt_sites *s)
{
    return s ? s->kernel_name : "";
}

extern "C" int rt_prune(rt_model *m, rt_sites *s)
{
    RT_REQUIRE(m && s, "null pointer");
    RT_REQUIRE(s->model == m, "the site batch belongs to another model");
    RT_REQUIRE(m->have_P, "the model has no transition matrices yet");
    RT_HIP(hipSetDevice(m->ctx->device));
    if (s->jit_job) RT_TRY(rt_sites_jit_poll(s, false));
    return rt_launch_prune(m, s);
}

// One step of the repeated-evaluation loop in one call: (expm of every edge from
// the resident rates) + pruning + reduce.  (Recording the three launches into a
// hipGraph and replaying it was measured and is slower here: 45.9 us per C2 step
// against 42.4 us with ordinary launches, ROCm 7.2.)
extern "C" int rt_step(rt_model *m, rt_sites *s, int recompute_transitions)
{
    RT_REQUIRE(m && s, "null pointer");
    RT_REQUIRE(s->model == m, "the site batch belongs to another model");
    RT_REQUIRE(!recompute_transitions || m->d_Q || m->spectral,
               "rt_model_set_rates has not been called");
    RT_REQUIRE(recompute_transitions || m->have_P,
               "the model has no transition matrices yet");
    RT_HIP(hipSetDevice(m->ctx->device));
    if (s->jit_job) RT_TRY(rt_sites_jit_poll(s, false));
    // n <= 4 and a tree-specialised kernel: the pruning launch computes the transitions from
    // the resident rates itself (and carries the previous step's batch sum): ONE launch
    const bool fuse = recompute_transitions && s->jit_fused && s->jit_fn && !m->spectral &&
                      m->d_Q && !getenv("RAOTEH_NO_DEFER_REDUCE");
    if (recompute_transitions && !fuse) RT_TRY(model_run_expm(m));
    // the batch sum is reduced by the next step's first launch (or by whoever reads the
    // totals first)
    return rt_launch_prune(m, s, true, fuse);
}

extern "C" int rt_sites_get_logliks(rt_sites *s, double *loglik, int32_t *status)
{
    RT_REQUIRE(s, "null pointer");
    RT_HIP(hipSetDevice(s->model->ctx->device));
    RT_HIP(hipStreamSynchronize(s->model->ctx->stream));
    if (loglik)
        RT_HIP(hipMemcpy(loglik, s->d_loglik, s->nsites * 8, hipMemcpyDeviceToHost));
    if (status)
        RT_HIP(hipMemcpy(status, s->d_status, s->nsites * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_sites_get_totals(rt_sites *s, double totals[3])
{
    RT_REQUIRE(s && totals, "null pointer");
    RT_HIP(hipSetDevice(s->model->ctx->device));
    RT_TRY(rt_flush_reduce(s->model->ctx));
    RT_HIP(hipStreamSynchronize(s->model->ctx->stream));
    if (s->model->ctx->comm_stream)
        RT_HIP(hipStreamSynchronize(s->model->ctx->comm_stream));
    RT_HIP(hipMemcpy(totals, s->d_totals, 3 * 8, hipMemcpyDeviceToHost));
    return RT_OK;
}

// Debug / test hook: copy the post-order schedule out (rt_op as int32[4]).
extern "C" int rt_model_get_schedule(const rt_model *m, int32_t *ops, int64_t capacity,
                                     int64_t *nops)
{
    RT_REQUIRE(m && nops, "null pointer");
    *nops = (int64_t)m->ops.size();
    if (ops) {
        RT_REQUIRE(capacity >= *nops, "capacity too small");
        memcpy(ops, m->ops.data(), m->ops.size() * sizeof(rt_op));
    }
    return RT_OK;
}
