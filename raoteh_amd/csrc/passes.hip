// The three pyfelscore passes on the reference's own array formats, batched
// over sites (one workgroup of 64 lanes per site, lane = state).
//
//   mcy_esd_get_node_to_pset   backward boolean pass   (_mcy_dense.py:270)
//   esd_get_node_to_set        forward boolean pass    (_mcy_dense.py:277)
//   mcy_esd_get_node_to_pmap   Felsenstein upward pass (_mcy_dense.py:286)
//
// Arrays: tree CSR int64 (children, DFS-preorder index space, so every child
// index is larger than its parent's), esd_transitions f64[N][n][n] keyed by the
// child index, state_mask int64[nsites][N][n], pmap f64[nsites][N][n].
// These kernels serve the drop-in API (per-node pmaps for every site); the
// log-likelihood hot path is prune.hip.
#include "common.h"

#include <algorithm>
#include <memory>

namespace {

// One 64-lane workgroup holds 64 / n sites (n <= 32; one site above, in a workgroup of
// 128 lanes for 64 < n <= 128): lane = (site slot, state).  The sites of a workgroup never exchange anything, so the
// arithmetic of a site is the same in every packing; at n = 4 a wave carries 16
// sites instead of one with 60 idle lanes.
struct site_lane {
    int s;          // state
    long site;      // site of this lane, -1 = idle lane
    int slot;
};

__device__ inline site_lane lane_site(int n, long nsites)
{
    const int per = n <= 32 ? 64 / n : 1;
    site_lane L;
    L.slot = (int)threadIdx.x / n;
    L.s = (int)threadIdx.x % n;
    const long site = (long)blockIdx.x * per + L.slot;
    L.site = (L.slot < per && site < nsites) ? site : -1;
    return L;
}

inline unsigned pass_grid(int64_t nsites, int64_t n)
{
    const int64_t per = n <= 32 ? 64 / n : 1;
    return (unsigned)((nsites + per - 1) / per);
}
inline unsigned pass_block(int64_t n) { return n > 64 ? 128u : 64u; }

__global__ void __launch_bounds__(128)
pset_kernel(int nnodes, int n, long nsites, const long *__restrict__ idx,
            const long *__restrict__ ptr, const double *__restrict__ esd,
            long *__restrict__ mask)
{
    const site_lane L = lane_site(n, nsites);
    const int s = L.s;
    long *mk = mask + (size_t)(L.site < 0 ? 0 : L.site) * nnodes * n;
    for (int v = nnodes - 1; v >= 0; --v) {
        if (L.site >= 0) {
            long keep = mk[(size_t)v * n + s] != 0;
            for (long e = ptr[v]; e < ptr[v + 1]; ++e) {
                const long c = idx[e];
                const double *Pc = esd + ((size_t)c * n + s) * n;
                const long *mc = mk + (size_t)c * n;
                long any = 0;
                for (int sp = 0; sp < n; ++sp)
                    any |= (Pc[sp] > 0.0) && (mc[sp] != 0);
                keep &= any;
            }
            mk[(size_t)v * n + s] = keep;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(128)
set_kernel(int nnodes, int n, long nsites, const long *__restrict__ idx,
           const long *__restrict__ ptr, const double *__restrict__ esd,
           long *__restrict__ mask)
{
    const site_lane L = lane_site(n, nsites);
    const int sp = L.s;             // child state
    long *mk = mask + (size_t)(L.site < 0 ? 0 : L.site) * nnodes * n;
    for (int v = 0; v < nnodes; ++v) {
        for (long e = ptr[v]; e < ptr[v + 1]; ++e) {
            const long c = idx[e];
            if (L.site >= 0) {
                const double *Pc = esd + (size_t)c * n * n;
                const long *mv = mk + (size_t)v * n;
                long any = 0;
                for (int s = 0; s < n; ++s)
                    any |= (mv[s] != 0) && (Pc[(size_t)s * n + sp] > 0.0);
                mk[(size_t)c * n + sp] = (mk[(size_t)c * n + sp] != 0) & any;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(128)
pmap_kernel(int nnodes, int n, long nsites, const long *__restrict__ idx,
            const long *__restrict__ ptr, const double *__restrict__ esd,
            const long *__restrict__ mask, const double *__restrict__ obs,
            double *__restrict__ out)
{
    const site_lane L = lane_site(n, nsites);
    const int s = L.s;
    const size_t base = (size_t)(L.site < 0 ? 0 : L.site) * nnodes * n;
    const long *mk = mask + base;
    double *o = out + base;
    for (int v = nnodes - 1; v >= 0; --v) {
        if (L.site >= 0) {
            double acc = 1.0;
            for (long e = ptr[v]; e < ptr[v + 1]; ++e) {
                const long c = idx[e];
                const double *Pc = esd + ((size_t)c * n + s) * n;
                const double *Lc = o + (size_t)c * n;
                double sum = 0.0;
                for (int sp = 0; sp < n; ++sp) sum = fma(Pc[sp], Lc[sp], sum);
                acc *= sum;
            }
            if (obs) acc *= obs[base + (size_t)v * n + s];
            o[(size_t)v * n + s] = mk[(size_t)v * n + s] != 0 ? acc : 0.0;
        }
        __syncthreads();
    }
}

// bump allocation out of the context's grow-only scratch (valid until the next call)
struct scratch_plan {
    size_t total = 0;
    size_t take(size_t bytes)
    {
        const size_t off = total;
        total += (bytes + 255) & ~(size_t)255;
        return off;
    }
};

int scratch_reserve(rt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return RT_OK;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
    RT_HIP(hipMalloc((void **)&ctx->d_scratch, want));
    ctx->scratch_bytes = want;
    return RT_OK;
}


// the tree of a reference-format call, in the context's scratch (nothing to free: a
// hipMalloc / hipFree pair per array was most of the 0.8-3.5 ms of a single-site call)
struct dev_tree {
    long *idx = nullptr, *ptr = nullptr;
    double *esd = nullptr;
    unsigned char *rest = nullptr;         // scratch behind the tree, `extra` bytes of it
};

int check_tree(int64_t nnodes, int64_t n, int64_t nsites, const int64_t *idx,
               const int64_t *ptr, const double *esd)
{
    RT_REQUIRE(nnodes >= 1 && n >= 1 && nsites >= 0, "bad sizes");
    RT_REQUIRE(n <= RT_MAX_STATES, "n=%lld > %d", (long long)n, RT_MAX_STATES);
    RT_REQUIRE(ptr && esd && (idx || nnodes == 1), "null array");
    RT_REQUIRE(ptr[0] == 0 && ptr[nnodes] == nnodes - 1,
               "tree_csr_indptr does not describe a tree");
    for (int64_t v = 0; v < nnodes; ++v) {
        RT_REQUIRE(ptr[v + 1] >= ptr[v], "tree_csr_indptr not monotone");
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e)
            RT_REQUIRE(idx[e] > v && idx[e] < nnodes,
                       "child index %lld of node %lld is not in preorder",
                       (long long)idx[e], (long long)v);
    }
    return RT_OK;
}

int upload_tree(rt_ctx *ctx, dev_tree &d, int64_t nnodes, int64_t n,
                const int64_t *idx, const int64_t *ptr, const double *esd, size_t extra = 0)
{
    const size_t ni = (size_t)(nnodes > 1 ? nnodes - 1 : 1);
    scratch_plan plan;
    const size_t o_idx = plan.take(ni * 8), o_ptr = plan.take((size_t)(nnodes + 1) * 8);
    const size_t o_esd = plan.take((size_t)nnodes * n * n * 8), o_rest = plan.take(extra);
    RT_TRY(scratch_reserve(ctx, plan.total));
    d.idx = (long *)(ctx->d_scratch + o_idx);
    d.ptr = (long *)(ctx->d_scratch + o_ptr);
    d.esd = (double *)(ctx->d_scratch + o_esd);
    d.rest = ctx->d_scratch + o_rest;
    if (nnodes > 1)
        RT_HIP(hipMemcpyAsync(d.idx, idx, (size_t)(nnodes - 1) * 8,
                              hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(hipMemcpyAsync(d.ptr, ptr, (size_t)(nnodes + 1) * 8, hipMemcpyHostToDevice,
                          ctx->stream));
    RT_HIP(hipMemcpyAsync(d.esd, esd, (size_t)nnodes * n * n * 8,
                          hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

int mask_pass(rt_ctx *ctx, bool forward, int64_t nnodes, int64_t n, int64_t nsites,
              const int64_t *idx, const int64_t *ptr, const double *esd,
              int64_t *state_mask)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(check_tree(nnodes, n, nsites, idx, ptr, esd));
    RT_REQUIRE(state_mask || nsites == 0, "null state_mask");
    if (nsites == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    dev_tree d;
    const size_t bytes = (size_t)nsites * nnodes * n * 8;
    RT_TRY(upload_tree(ctx, d, nnodes, n, idx, ptr, esd, bytes));
    long *dm = (long *)d.rest;
    hipError_t e = hipMemcpyAsync(dm, state_mask, bytes, hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e == hipSuccess) {
        if (forward)
            hipLaunchKernelGGL(set_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, ctx->stream, (int)nnodes, (int)n,
                       (long)nsites, d.idx, d.ptr, d.esd, dm);
        else
            hipLaunchKernelGGL(pset_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, ctx->stream, (int)nnodes, (int)n,
                       (long)nsites, d.idx, d.ptr, d.esd, dm);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(state_mask, dm, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        rt_set_error("mask pass failed: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return RT_OK;
}

}  // namespace

int rt_scratch_reserve(rt_ctx *ctx, size_t bytes) { return scratch_reserve(ctx, bytes); }

// The pass sequence of _mcy_dense.py:261-291 / _mcz.py:128-163 in ONE call: the tree,
// the transition matrices and the masks go to the device once, the three kernels
// run back to back, masks and pmaps come back once.  (Through the three separate
// entry points a single 61-state site uploads its 3.75 MB of matrices three times
// and pays eleven hipMalloc / hipFree pairs: 11.7 ms per site.)
extern "C" int rt_mcy_esd_passes(rt_ctx *ctx, int64_t nnodes, int64_t n, int64_t nsites,
        const int64_t *idx, const int64_t *ptr, const double *esd, int64_t *state_mask,
        const double *obs_likelihood, double *subtree_probability)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(check_tree(nnodes, n, nsites, idx, ptr, esd));
    RT_REQUIRE((state_mask && subtree_probability) || nsites == 0, "null array");
    if (nsites == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)nsites * nnodes * n * 8;
    const size_t ni = (size_t)(nnodes > 1 ? nnodes - 1 : 1);
    scratch_plan plan;
    const size_t o_idx = plan.take(ni * 8), o_ptr = plan.take((size_t)(nnodes + 1) * 8);
    const size_t o_esd = plan.take((size_t)nnodes * n * n * 8);
    const size_t o_mask = plan.take(bytes), o_out = plan.take(bytes);
    const size_t o_obs = obs_likelihood ? plan.take(bytes) : 0;
    RT_TRY(scratch_reserve(ctx, plan.total));
    unsigned char *base = ctx->d_scratch;
    long *d_idx = (long *)(base + o_idx), *d_ptr = (long *)(base + o_ptr);
    double *d_esd = (double *)(base + o_esd), *d_out = (double *)(base + o_out);
    long *d_mask = (long *)(base + o_mask);
    double *d_obs = obs_likelihood ? (double *)(base + o_obs) : nullptr;
    hipStream_t st = ctx->stream;
    if (nnodes > 1)
        RT_HIP(hipMemcpyAsync(d_idx, idx, (size_t)(nnodes - 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_ptr, ptr, (size_t)(nnodes + 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_esd, esd, (size_t)nnodes * n * n * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_mask, state_mask, bytes, hipMemcpyHostToDevice, st));
    if (d_obs) RT_HIP(hipMemcpyAsync(d_obs, obs_likelihood, bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pset_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_mask);
    hipLaunchKernelGGL(set_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_mask);
    hipLaunchKernelGGL(pmap_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_mask, d_obs, d_out);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(state_mask, d_mask, bytes, hipMemcpyDeviceToHost, st));
    RT_HIP(hipMemcpyAsync(subtree_probability, d_out, bytes, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

extern "C" int rt_mcy_esd_get_node_to_pset(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        int64_t *state_mask)
{
    return mask_pass(ctx, false, nnodes, n, nsites, idx, ptr, esd, state_mask);
}

extern "C" int rt_esd_get_node_to_set(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        int64_t *state_mask)
{
    return mask_pass(ctx, true, nnodes, n, nsites, idx, ptr, esd, state_mask);
}

extern "C" int rt_mcy_esd_get_node_to_pmap(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        const int64_t *state_mask, const double *obs_likelihood,
        double *subtree_probability)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(check_tree(nnodes, n, nsites, idx, ptr, esd));
    RT_REQUIRE((state_mask && subtree_probability) || nsites == 0, "null array");
    if (nsites == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    dev_tree d;
    const size_t bytes = (size_t)nsites * nnodes * n * 8;
    const size_t slot = (bytes + 255) & ~(size_t)255;
    RT_TRY(upload_tree(ctx, d, nnodes, n, idx, ptr, esd, 3 * slot));
    long *dm = (long *)d.rest;
    double *dout = (double *)(d.rest + slot);
    double *dobs = obs_likelihood ? (double *)(d.rest + 2 * slot) : nullptr;
    hipError_t e = hipMemcpyAsync(dm, state_mask, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && obs_likelihood)
        e = hipMemcpyAsync(dobs, obs_likelihood, bytes, hipMemcpyHostToDevice,
                           ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pmap_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, ctx->stream, (int)nnodes, (int)n,
                       (long)nsites, d.idx, d.ptr, d.esd, dm, dobs, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(subtree_probability, dout, bytes, hipMemcpyDeviceToHost,
                           ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        rt_set_error("pmap pass failed: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return RT_OK;
}

// ---------------------------------------------------------------------------
// Downward pass: posterior marginal state distribution at every node, and the
// joint (parent state, child state) posterior on every edge.
//   pyfelscore.mc0_esd_get_node_to_distn        (_mc0_dense.py:381, _mcy_dense.py:195;
//                                                pure-Python twin _mc0_dense.py:446-486)
//   pyfelscore.mc0_esd_get_joint_endpoint_distn (_mcy_dense.py:205; twin
//                                                _mc0_dense.py:246-267)
// For an edge (a -> b) with upward messages pmap:
//   den[sa]    = sum_s' P_b[sa,s'] pmap[b,s']
//   J[sa,sb]   = distn[a,sa] * P_b[sa,sb] * pmap[b,sb] / den[sa]     (0 where distn[a,sa] = 0)
//   distn[b,sb] = sum_sa J[sa,sb]
// One 64-lane workgroup per site; lane = state.  status (optional): 2 where a
// normalising denominator is zero (the reference raises NumericalZeroProb,
// _util.py:164-165).
// ---------------------------------------------------------------------------

namespace {

template <bool JOINT>
__global__ void __launch_bounds__(128)
distn_kernel(int nnodes, int n, long nsites, const long *__restrict__ idx,
             const long *__restrict__ ptr, const double *__restrict__ esd,
             const double *__restrict__ root_distn, const double *__restrict__ pmap_all,
             double *__restrict__ distn_all, double *__restrict__ joint_all,
             int *__restrict__ status)
{
    __shared__ double wbuf[128];
    __shared__ int bad[64];
    const site_lane L = lane_site(n, nsites);
    const int s = L.s, lane0 = L.slot * n;         // first lane of this site
    const bool on = L.site >= 0;
    const size_t base = (size_t)(on ? L.site : 0) * nnodes * n;
    const double *pm = pmap_all + base;
    double *dn = distn_all + base;
    double *jt = JOINT ? joint_all + base * n : nullptr;
    if (threadIdx.x < 64) bad[threadIdx.x] = 0;
    __syncthreads();
    if (!JOINT) {
        // root: normalised pmap * prior (_mc0_dense.py:458-459); the sum runs over the
        // site's states in lane order
        double w = 0.0;
        if (on) w = pm[s] * (root_distn ? root_distn[s] : 1.0);
        wbuf[threadIdx.x] = w;
        __syncthreads();
        if (on) {
            double tot = 0.0;
            for (int k = 0; k < n; ++k) tot += wbuf[lane0 + k];
            if (!(tot > 0.0)) bad[L.slot] = 1;
            dn[s] = tot > 0.0 ? w / tot : 0.0;
        }
    } else if (on) {
        for (int sa = 0; sa < n; ++sa) jt[(size_t)sa * n + s] = 0.0;   // root slot
    }
    __syncthreads();
    for (int v = 0; v < nnodes; ++v) {
        for (long e = ptr[v]; e < ptr[v + 1]; ++e) {
            const long c = idx[e];
            const double *Pc = esd + (size_t)c * n * n;
            const double *Lc = pm + (size_t)c * n;
            if (on) {
                const double pa = dn[(size_t)v * n + s];
                double den = 0.0;
                for (int sp = 0; sp < n; ++sp) den = fma(Pc[(size_t)s * n + sp], Lc[sp], den);
                double w = 0.0;
                if (pa != 0.0) {
                    if (den > 0.0) w = pa / den;
                    else bad[L.slot] = 1;
                }
                wbuf[threadIdx.x] = w;
            }
            __syncthreads();
            if (on) {
                const double lb = Lc[s];
                if (JOINT) {
                    for (int sa = 0; sa < n; ++sa)
                        jt[((size_t)c * n + sa) * n + s] =
                            wbuf[lane0 + sa] * Pc[(size_t)sa * n + s] * lb;
                } else {
                    double acc = 0.0;
                    for (int sa = 0; sa < n; ++sa)
                        acc = fma(wbuf[lane0 + sa], Pc[(size_t)sa * n + s], acc);
                    dn[(size_t)c * n + s] = acc * lb;
                }
            }
            __syncthreads();
        }
    }
    if (status && on && s == 0) status[L.site] = bad[L.slot] ? 2 : 0;
}

int distn_pass(rt_ctx *ctx, bool joint, int64_t nnodes, int64_t n, int64_t nsites,
               const int64_t *idx, const int64_t *ptr, const double *esd,
               const double *root_distn, const double *pmap, double *distn,
               double *joint_out, int32_t *status)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(check_tree(nnodes, n, nsites, idx, ptr, esd));
    RT_REQUIRE((pmap && distn && (!joint || joint_out)) || nsites == 0, "null array");
    if (nsites == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    dev_tree d;
    const size_t bytes = (size_t)nsites * nnodes * n * 8;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_p = 0, o_d = o_p + up(bytes), o_j = o_d + up(bytes),
                 o_r = o_j + (joint ? up(bytes * n) : 0), o_s = o_r + up((size_t)n * 8),
                 extra = o_s + up((size_t)nsites * 4);
    RT_TRY(upload_tree(ctx, d, nnodes, n, idx, ptr, esd, extra));
    double *dp = (double *)(d.rest + o_p), *dd = (double *)(d.rest + o_d);
    double *dj = joint ? (double *)(d.rest + o_j) : nullptr;
    double *dr = root_distn ? (double *)(d.rest + o_r) : nullptr;
    int *ds = (int *)(d.rest + o_s);
    hipError_t e = hipMemcpyAsync(dp, pmap, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && joint)
        e = hipMemcpyAsync(dd, distn, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && root_distn)
        e = hipMemcpyAsync(dr, root_distn, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        if (joint)
            hipLaunchKernelGGL(distn_kernel<true>, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, ctx->stream, (int)nnodes, (int)n,
                       (long)nsites, d.idx, d.ptr, d.esd, dr, dp,
                               dd, dj, ds);
        else
            hipLaunchKernelGGL(distn_kernel<false>, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, ctx->stream, (int)nnodes, (int)n,
                       (long)nsites, d.idx, d.ptr, d.esd, dr, dp,
                               dd, dj, ds);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !joint)
        e = hipMemcpyAsync(distn, dd, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && joint)
        e = hipMemcpyAsync(joint_out, dj, bytes * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && status)
        e = hipMemcpyAsync(status, ds, nsites * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        rt_set_error("distn pass failed: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return RT_OK;
}

}  // namespace

extern "C" int rt_mc0_esd_get_node_to_distn(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        const double *root_distn, const double *subtree_probability,
        double *node_to_distn_array, int32_t *status)
{
    return distn_pass(ctx, false, nnodes, n, nsites, idx, ptr, esd, root_distn,
                      subtree_probability, node_to_distn_array, nullptr, status);
}

extern "C" int rt_mc0_esd_get_joint_endpoint_distn(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        const double *subtree_probability, const double *node_to_distn_array,
        double *joint_distns)
{
    return distn_pass(ctx, true, nnodes, n, nsites, idx, ptr, esd, nullptr,
                      subtree_probability, const_cast<double *>(node_to_distn_array),
                      joint_distns, nullptr);
}

// ---------------------------------------------------------------------------
// Site sums for the expected history statistics (_mjp_dense.py:410-539).
// Per edge (a -> b) the reference contracts J / P (J the joint endpoint posterior,
// over its nonzero entries) with Frechet derivatives; with the quantities of the
// downward pass
//   J[sa,sb] / P[sa,sb] = (distn[a,sa] / den[sa]) * pmap[b,sb]        where P[sa,sb] != 0
// an outer product per site, so the site sum W_b = sum_s w_s u_s (x) p_s (masked by
// the pattern of P_b) is all that has to leave the device: n*n numbers per edge
// instead of n*n per edge AND site.  upward passes -> downward pass -> these sums
// run back to back on the arrays of one upload.
// ---------------------------------------------------------------------------

namespace {

// grid (nnodes, G): block (c, g) adds the sites g, g + G, ... of edge (parent[c] -> c)
// in that order; slot c = 0 carries the summed root posterior in its column 0
__global__ void __launch_bounds__(256)
ratio_sum_kernel(int nnodes, int n, int nsites, const int *__restrict__ parent,
                 const double *__restrict__ esd, const double *__restrict__ pmap_all,
                 const double *__restrict__ distn_all, const double *__restrict__ weights,
                 double *__restrict__ part)
{
    __shared__ double Ps[RT_MAX_EXPECT_STATES * RT_MAX_EXPECT_STATES];
    __shared__ double p[RT_MAX_EXPECT_STATES], u[RT_MAX_EXPECT_STATES];
    const int c = blockIdx.x, g = blockIdx.y, G = gridDim.y, t = threadIdx.x;
    const int nn = n * n;
    constexpr int PER = RT_MAX_EXPECT_STATES * RT_MAX_EXPECT_STATES / 256;
    double acc[PER];
    int ea[PER], eb[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        acc[k] = 0.0;
        const int e = t + 256 * k;
        ea[k] = e < nn ? e / n : 0;
        eb[k] = e < nn ? e % n : 0;
    }
    double *out = part + ((size_t)g * nnodes + c) * nn;
    if (c == 0) {
        double r = 0.0;
        for (int s = g; s < nsites; s += G)
            if (t < n) r = fma(weights ? weights[s] : 1.0, distn_all[(size_t)s * nnodes * n + t], r);
        for (int e = t; e < nn; e += 256) out[e] = 0.0;
        __syncthreads();
        if (t < n) out[(size_t)t * n] = r;
        return;
    }
    for (int e = t; e < nn; e += 256) Ps[e] = esd[(size_t)c * nn + e];
    const int par = parent[c];
    __syncthreads();
    for (int s = g; s < nsites; s += G) {
        const size_t base = (size_t)s * nnodes * n;
        if (t < n) p[t] = pmap_all[base + (size_t)c * n + t];
        __syncthreads();
        if (t < n) {
            const double pa = distn_all[base + (size_t)par * n + t];
            double den = 0.0;
            for (int b = 0; b < n; ++b) den = fma(Ps[t * n + b], p[b], den);
            u[t] = (pa != 0.0 && den > 0.0) ? (weights ? weights[s] : 1.0) * pa / den : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) acc[k] = fma(u[ea[k]], p[eb[k]], acc[k]);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = t + 256 * k;
        if (e < nn) out[e] = Ps[e] != 0.0 ? acc[k] : 0.0;
    }
}

// n*n <= 128: 256 / (n*n) sites per iteration, one (site slot, a, b) per thread -- the
// loop above is bound by the latency of its two dependent global loads and three
// barriers per site, which this divides by the number of slots (n = 4: 16)
__global__ void __launch_bounds__(256)
ratio_sum_small_kernel(int nnodes, int n, int nsites, const int *__restrict__ parent,
                       const double *__restrict__ esd, const double *__restrict__ pmap_all,
                       const double *__restrict__ distn_all, const double *__restrict__ weights,
                       double *__restrict__ part)
{
    __shared__ double Ps[128];
    __shared__ double p[256], u[256], red[256];
    const int c = blockIdx.x, g = blockIdx.y, G = gridDim.y, t = threadIdx.x;
    const int nn = n * n, slots = 256 / nn;
    double *out = part + ((size_t)g * nnodes + c) * nn;
    if (c == 0) {
        double r = 0.0;
        for (int s = g; s < nsites; s += G)
            if (t < n) r = fma(weights ? weights[s] : 1.0, distn_all[(size_t)s * nnodes * n + t], r);
        for (int e = t; e < nn; e += 256) out[e] = 0.0;
        __syncthreads();
        if (t < n) out[(size_t)t * n] = r;
        return;
    }
    if (t < nn) Ps[t] = esd[(size_t)c * nn + t];
    const int par = parent[c];
    const int q1 = t / n, j1 = t % n;               // loader role: slot, state
    const int q = t / nn, e = t % nn, a = e / n, b = e % n;
    const bool loader = t < slots * n, worker = t < slots * nn;
    double acc = 0.0;
    __syncthreads();
    for (long s0 = (long)g * slots; s0 < nsites; s0 += (long)G * slots) {
        const long site = s0 + q1;
        const bool live = loader && site < nsites;
        const size_t base = (size_t)site * nnodes * n;
        if (loader) p[t] = live ? pmap_all[base + (size_t)c * n + j1] : 0.0;
        const double pa = live ? distn_all[base + (size_t)par * n + j1] : 0.0;
        __syncthreads();
        if (loader) {
            double den = 0.0;
            for (int k = 0; k < n; ++k) den = fma(Ps[j1 * n + k], p[q1 * n + k], den);
            u[t] = (pa != 0.0 && den > 0.0) ? (weights ? weights[site] : 1.0) * pa / den : 0.0;
        }
        __syncthreads();
        if (worker) acc = fma(u[q * n + a], p[q * n + b], acc);
        __syncthreads();
    }
    red[t] = worker ? acc : 0.0;
    __syncthreads();
    if (t < nn) {
        double sum = 0.0;
        for (int k = 0; k < slots; ++k) sum += red[k * nn + t];
        out[t] = Ps[t] != 0.0 ? sum : 0.0;
    }
}

// the same with one workgroup per 4 outputs and 64 threads per output: thread t of an
// output adds partials t, t + 64, ... in that order, then a fixed tree over the 64 sums
// (1 563 partials per output at 100 000 sites: one thread per output is a serial chain)
__global__ void __launch_bounds__(256)
sum_parts_wide_kernel(int G, long count, const double *__restrict__ part, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= count) return;
    double s = 0.0;
    for (int g = lane; g < G; g += 64) s += part[(size_t)g * count + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[i] = s;
}

__global__ void __launch_bounds__(256)
sum_parts_kernel(int G, long count, const double *__restrict__ part, double *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * count + i];
    out[i] = s;
}

}  // namespace

namespace {

// state_mask[site][node][state] from one byte / one 64-bit set per (site, observed node):
// every other node is unrestricted (RT_OBS_STATE: a value >= n is "unobserved")
__global__ void __launch_bounds__(256)
expand_obs_kernel(int nnodes, int n, long nsites, int nobs, const int *__restrict__ obs_nodes,
                  int kind, const void *__restrict__ data, long *__restrict__ mask)
{
    const long total = nsites * nnodes * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        mask[i] = 1;            // the observed rows are written by the next launch
}

__global__ void __launch_bounds__(256)
apply_obs_kernel(int nnodes, int n, long nsites, int nobs, const int *__restrict__ obs_nodes,
                 int kind, const void *__restrict__ data, long *__restrict__ mask)
{
    const long total = nsites * nobs * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int s = (int)(i % n);
        const long so = i / n;                  // site * nobs + j
        const int j = (int)(so % nobs);
        const long site = so / nobs;
        long allowed;
        if (kind == RT_OBS_STATE) {
            const unsigned v = ((const unsigned char *)data)[so];
            allowed = (v >= (unsigned)n || v == (unsigned)s) ? 1 : 0;
        } else {
            allowed = (long)((((const unsigned long long *)data)[so] >> s) & 1ull);
        }
        mask[(site * nnodes + obs_nodes[j]) * n + s] = allowed;
    }
}

// ---------------------------------------------------------------------------
// n <= 8: the whole chain -- both boolean passes, the upward pass, the downward pass
// and the per-edge site sums of J / P -- in ONE kernel with a LANE PER SITE, on the
// resident layout [node][state][site] (a wave touches 64 consecutive sites of one
// (node, state) row: every access is a coalesced 512-byte row).  Allowed sets are one
// byte per node and site; the matrices are wave-uniform (scalar loads).  The kernels
// above put (site slot, state) on the lanes of the reference's [site][node][state]
// arrays and synchronise per node: 1.2-1.45 ms per pass and 100 000 4-state sites
// against the ~0.1 ms their bytes cost, plus 6.7 ms for the site sums.  Here: three
// arrays (message to the parent M, subtree likelihood L, posterior D) are written once
// and read once or twice, and a site's arithmetic is the same fma chains in the same
// order as in pmap_kernel / distn_kernel, so the numbers are those of the old path.
// The site sums: per edge and (a, b) one wave reduction over the 64 sites, one partial
// per wave, then sum_parts_kernel adds the waves in index order (fixed rounding).
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
sets_from_mask_kernel(int nnodes, int n, long nsites, long S, const long *__restrict__ mask,
                      unsigned char *__restrict__ sets)
{
    const long total = (long)nnodes * S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long v = i / S, site = i - v * S;
        unsigned m = 0;
        if (site < nsites)
            for (int s = 0; s < n; ++s)
                m |= (mask[((size_t)site * nnodes + v) * n + s] != 0 ? 1u : 0u) << s;
        else
            m = (1u << n) - 1u;
        sets[i] = (unsigned char)m;
    }
}

__global__ void __launch_bounds__(256)
sets_fill_kernel(long total, int n, unsigned char *__restrict__ sets)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        sets[i] = (unsigned char)((1u << n) - 1u);
}

__global__ void __launch_bounds__(256)
sets_apply_obs_kernel(int n, long nsites, long S, int nobs, const int *__restrict__ obs_nodes,
                      int kind, const void *__restrict__ data, unsigned char *__restrict__ sets)
{
    const long total = nsites * nobs;
    const unsigned full = (1u << n) - 1u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % nobs);
        const long site = i / nobs;
        unsigned m;
        if (kind == RT_OBS_STATE) {
            const unsigned v = ((const unsigned char *)data)[i];
            m = v >= (unsigned)n ? full : 1u << v;
        } else {
            m = (unsigned)(((const unsigned long long *)data)[i]) & full;
        }
        sets[(size_t)obs_nodes[j] * S + site] = (unsigned char)m;
    }
}

// Sum P = 2^k values per lane over the 64 lanes of a wave: in each of the first k steps a
// lane hands half of its values to its partner (lane ^ 32, ^ 16, ...) and keeps the sums of
// the other half -- P - 1 exchanged values in all instead of 6 P for P separate butterfly
// reductions; the remaining 6 - k steps finish the one value left.  Afterwards x[0] of
// lane l is the total of entry (l >> (6 - k)) & (P - 1); the lanes whose low 6 - k bits
// are zero own distinct entries.  Fixed order: deterministic sums.
template <int CNT, int H, int P>
__device__ __forceinline__ void wave_sum_many_step(double (&x)[P], int lane)
{
    // (a template recursion: written as one loop over cnt and h the compiler does not
    // unroll it and indexes the register array with selects, 1 300 instructions per call)
    if constexpr (CNT > 1) {
        const bool up = (lane & H) != 0;
#pragma unroll
        for (int i = 0; i < CNT / 2; ++i) {
            const double send = up ? x[i] : x[i + CNT / 2];
            const double keep = up ? x[i + CNT / 2] : x[i];
            x[i] = keep + __shfl_xor(send, H, 64);
        }
        wave_sum_many_step<CNT / 2, H / 2, P>(x, lane);
    } else if constexpr (H > 0) {
        x[0] += __shfl_xor(x[0], H, 64);
        wave_sum_many_step<1, H / 2, P>(x, lane);
    }
}

template <int P>
__device__ __forceinline__ void wave_sum_many(double (&x)[P], int lane)
{
    wave_sum_many_step<P, 32, P>(x, lane);
}

template <int N>
__global__ void __launch_bounds__(64)
expect_lane_kernel(int nnodes, long nsites, long S, const int *__restrict__ parent,
                   const long *__restrict__ cidx, const long *__restrict__ cptr,
                   const double *__restrict__ esd, const unsigned char *__restrict__ rowbits,
                   const unsigned char *__restrict__ colbits, const double *__restrict__ root_distn,
                   const double *__restrict__ weights, unsigned char *__restrict__ sets,
                   double *__restrict__ M, double *__restrict__ Lb, double *__restrict__ Dn,
                   double *__restrict__ part, int *__restrict__ status)
{
    const int lane = threadIdx.x;
    const long site = (long)blockIdx.x * 64 + lane;
    const bool live = site < nsites;
    const size_t row = (size_t)S;                     // doubles per (node, state) row
    // ---- backward boolean pass (pyfelscore.mcy_esd_get_node_to_pset) ----
    for (int v = nnodes - 1; v >= 1; --v) {
        const unsigned sv = sets[(size_t)v * S + site];
        unsigned keep = 0;
#pragma unroll
        for (int a = 0; a < N; ++a) keep |= (rowbits[v * N + a] & sv) ? 1u << a : 0u;
        sets[(size_t)parent[v] * S + site] &= (unsigned char)keep;
    }
    // ---- forward boolean pass (pyfelscore.esd_get_node_to_set) ----
    for (int v = 1; v < nnodes; ++v) {
        const unsigned pv = sets[(size_t)parent[v] * S + site];
        unsigned reach = 0;
#pragma unroll
        for (int b = 0; b < N; ++b) reach |= (colbits[v * N + b] & pv) ? 1u << b : 0u;
        sets[(size_t)v * S + site] &= (unsigned char)reach;
    }
    // ---- upward pass (pyfelscore.mcy_esd_get_node_to_pmap) ----
    for (int v = nnodes - 1; v >= 0; --v) {
        double acc[N];
#pragma unroll
        for (int a = 0; a < N; ++a) acc[a] = 1.0;
        for (long e = cptr[v]; e < cptr[v + 1]; ++e) {
            const size_t c = (size_t)cidx[e];
#pragma unroll
            for (int a = 0; a < N; ++a) acc[a] *= M[(c * N + a) * row + site];
        }
        const unsigned sv = sets[(size_t)v * S + site];
        double l[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            l[a] = (sv >> a) & 1u ? acc[a] : 0.0;
            Lb[((size_t)v * N + a) * row + site] = l[a];
        }
        if (v > 0) {
            const double *Pv = esd + (size_t)v * N * N;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                double sum = 0.0;
#pragma unroll
                for (int b = 0; b < N; ++b) sum = fma(Pv[a * N + b], l[b], sum);
                M[((size_t)v * N + a) * row + site] = sum;
            }
        }
    }
    // ---- downward pass (pyfelscore.mc0_esd_get_node_to_distn) + site sums ----
    const double wt = (live && weights) ? weights[site] : (live ? 1.0 : 0.0);
    double *out = part + (size_t)blockIdx.x * nnodes * N * N;
    bool bad = false;
    {
        double w[N], tot = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            w[a] = Lb[(size_t)a * row + site] * (root_distn ? root_distn[a] : 1.0);
            tot += w[a];
        }
        if (!(tot > 0.0)) bad = true;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double d = tot > 0.0 ? w[a] / tot : 0.0;
            Dn[(size_t)a * row + site] = d;
            // slot 0 of the output: the weighted sum of the root posteriors, column 0
            double r = live ? wt * d : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
#pragma unroll
            for (int b = 0; b < N; ++b)
                if (lane == 0) out[a * N + b] = b == 0 ? r : 0.0;
        }
    }
    for (int v = 1; v < nnodes; ++v) {
        const size_t p = (size_t)parent[v];
        const double *Pv = esd + (size_t)v * N * N;
        double u[N], lv[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double pa = Dn[(p * N + a) * row + site];
            const double den = M[((size_t)v * N + a) * row + site];
            u[a] = 0.0;
            if (pa != 0.0) {
                if (den > 0.0) u[a] = pa / den;
                else bad = true;
            }
            lv[a] = Lb[((size_t)v * N + a) * row + site];
        }
#pragma unroll
        for (int b = 0; b < N; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a) acc = fma(u[a], Pv[a * N + b], acc);
            Dn[((size_t)v * N + b) * row + site] = acc * lv[b];
        }
        {
            constexpr int V = N * N;
            constexpr int P = V <= 1 ? 1 : V <= 2 ? 2 : V <= 4 ? 4 : V <= 8 ? 8 : V <= 16 ? 16
                              : V <= 32 ? 32 : 64;
            constexpr int K = P == 1 ? 0 : P == 2 ? 1 : P == 4 ? 2 : P == 8 ? 3 : P == 16 ? 4
                              : P == 32 ? 5 : 6;
            double x[P];
#pragma unroll
            for (int e = 0; e < P; ++e) x[e] = 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                const double ua = live ? wt * u[a] : 0.0;
#pragma unroll
                for (int b = 0; b < N; ++b) x[a * N + b] = ua * lv[b];
            }
            wave_sum_many<P>(x, lane);
            const int ent = (lane >> (6 - K)) & (P - 1);
            if ((lane & ((1 << (6 - K)) - 1)) == 0 && ent < V)
                out[(size_t)v * V + ent] = Pv[ent] != 0.0 ? x[0] : 0.0;
        }
    }
    if (status && live) status[site] = bad ? 2 : 0;
}

// ---------------------------------------------------------------------------
// The same computation with every dependent access in LDS.  In the kernel above a wave
// walks the tree through global memory: each node of each pass waits for one global load
// that the previous node's store may alias (0.8 ms for ONE wave of the 127-node tree,
// 1.3 us per node and pass).  Here the host schedules the two numeric passes as a stack
// program over a few LDS slots (one N-vector per lane each):
//   up   -- post-order, heaviest child first; a finished message is written to a slot or
//           multiplied into the product of its earlier siblings, so the slots in use never
//           exceed log2(nodes) + 1;
//   down -- the reverse order (parents first, heaviest child last); a node's posterior
//           lives in a slot until its last child has read it (that child may reuse it).
// The boolean passes run on an LDS copy of the wave's state sets; topology, the row /
// column supports and (when it fits) the transition matrices are LDS copies too.  Global
// traffic per site: the subtree likelihoods L[node][state] written once (up) and read once
// (down, prefetched four nodes ahead: the addresses come from the schedule, not from data);
// the message M = P L is recomputed in the down pass with the fma order of the up pass.
// ---------------------------------------------------------------------------
struct lane_plan {
    std::vector<int> ops;        // 4 ints per op: node, in | out << 8 | merge << 16, pslot | oslot << 8, 0
    int nslots = 1;
};

static bool build_lane_plan(int64_t nnodes, const int64_t *idx, const int64_t *ptr,
                            const std::vector<int> &parent, lane_plan &plan)
{
    const size_t nn = (size_t)nnodes;
    std::vector<int> size(nn, 1), first(nn, 0), nch(nn, 0);
    for (int64_t v = nnodes - 1; v >= 1; --v) size[(size_t)parent[(size_t)v]] += size[(size_t)v];
    std::vector<std::vector<int>> kids(nn);
    for (int64_t v = 0; v < nnodes; ++v) {
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) kids[(size_t)v].push_back((int)idx[e]);
        std::stable_sort(kids[(size_t)v].begin(), kids[(size_t)v].end(),
                         [&](int a, int b) { return size[(size_t)a] > size[(size_t)b]; });
        nch[(size_t)v] = (int)kids[(size_t)v].size();
        if (!kids[(size_t)v].empty()) first[(size_t)kids[(size_t)v][0]] = 1;
    }
    std::vector<int> order;
    order.reserve(nn);
    {
        std::vector<std::pair<int, int>> stack;      // node, next child position
        stack.emplace_back(0, 0);
        while (!stack.empty()) {
            auto &top = stack.back();
            if (top.second < nch[(size_t)top.first]) {
                const int c = kids[(size_t)top.first][(size_t)top.second++];
                stack.emplace_back(c, 0);
            } else {
                order.push_back(top.first);
                stack.pop_back();
            }
        }
    }
    if (order.size() != nn) return false;
    plan.ops.assign(nn * 4, 0);
    int sp = 0, hi = 0;
    for (size_t i = 0; i < nn; ++i) {
        const int v = order[i];
        int in = 255, out = 255, merge = 0;
        if (nch[(size_t)v] > 0) in = --sp;
        if (v != 0) {
            if (first[(size_t)v]) out = sp++;
            else { out = sp - 1; merge = 1; }
        }
        hi = std::max(hi, sp);
        if (sp < 0 || hi > 250) return false;
        plan.ops[i * 4 + 0] = v;
        plan.ops[i * 4 + 1] = in | out << 8 | merge << 16;
    }
    std::vector<int> slot(nn, 255), remaining(nch);
    std::vector<char> used;
    auto alloc = [&]() {
        for (size_t k = 0; k < used.size(); ++k)
            if (!used[k]) { used[k] = 1; return (int)k; }
        used.push_back(1);
        return (int)used.size() - 1;
    };
    for (size_t r = 0; r < nn; ++r) {
        const size_t i = nn - 1 - r;
        const int v = order[i];
        int ps = 255;
        if (v != 0) {
            const int p = parent[(size_t)v];
            ps = slot[(size_t)p];
            if (--remaining[(size_t)p] == 0) used[(size_t)ps] = 0;
        }
        if (nch[(size_t)v] > 0) slot[(size_t)v] = alloc();
        if (used.size() > 250) return false;
        plan.ops[i * 4 + 2] = ps | slot[(size_t)v] << 8;
    }
    plan.nslots = std::max<int>(std::max<int>(hi, (int)used.size()), 1);
    return true;
}

template <int N, bool PLDS>
__global__ void __launch_bounds__(512)
expect_lane_lds_kernel(int nnodes, long nsites, long S, int nslots, const int4 *__restrict__ ops,
                       const int *__restrict__ parent, const double *__restrict__ esd,
                       const unsigned char *__restrict__ rowbits,
                       const unsigned char *__restrict__ colbits,
                       const double *__restrict__ root_distn, const double *__restrict__ weights,
                       const unsigned char *__restrict__ sets, double *__restrict__ Lb,
                       double *__restrict__ part, int *__restrict__ status,
                       unsigned long long *__restrict__ trace)
{
#define RT_STAMP(k)                                                                   \
    if (trace && blockIdx.x == 0 && threadIdx.x == 0) trace[k] = __builtin_readcyclecounter()
    RT_STAMP(0);
    constexpr int NN = N * N;
    // W waves per workgroup share the copies of P and of the topology; each wave has its
    // own slots and state sets (40 KB for one wave alone allowed three waves per CU)
    extern __shared__ double lds_raw[];
    const int W = blockDim.x >> 6, wv = threadIdx.x >> 6;
    double *lP = lds_raw;                                           // [node][a][b] when PLDS
    int4 *lops = (int4 *)(lP + (PLDS ? ((size_t)nnodes * NN + 1) / 2 * 2 : 0));   // 16 B aligned
    int *lpar = (int *)(lops + nnodes);
    unsigned char *lrb = (unsigned char *)(lpar + nnodes);
    unsigned char *lcb = lrb + (size_t)nnodes * N;
    const size_t shared_bytes = ((size_t)(lcb + (size_t)nnodes * N - (unsigned char *)lds_raw) + 15) & ~(size_t)15;
    const size_t wave_bytes = (size_t)nslots * N * 512 + (((size_t)nnodes * 64 + 15) & ~(size_t)15);
    double *slots = (double *)((unsigned char *)lds_raw + shared_bytes + wv * wave_bytes);   // [slot][state][lane]
    unsigned char *lsets = (unsigned char *)(slots + (size_t)nslots * N * 64);   // [node][lane]
    const int lane = threadIdx.x & 63;
    const long wave_global = (long)blockIdx.x * W + wv;
    const long site = wave_global * 64 + lane;
    const bool live = site < nsites;
    const size_t row = (size_t)S;
    for (int i = threadIdx.x; i < nnodes; i += blockDim.x) {
        lops[i] = ops[i];
        lpar[i] = parent[i];
    }
    for (int i = threadIdx.x; i < nnodes * N; i += blockDim.x) {
        lrb[i] = rowbits[i];
        lcb[i] = colbits[i];
    }
    if (PLDS)
        for (int i = threadIdx.x; i < nnodes * NN; i += blockDim.x) lP[i] = esd[i];
    const bool idle = wave_global * 64 >= S;             // a wave past the padded batch
    if (!idle)
        for (int v = 0; v < nnodes; ++v) lsets[v * 64 + lane] = sets[(size_t)v * S + site];
    __syncthreads();
    if (idle) return;
    RT_STAMP(1);
    // ---- backward / forward boolean passes on the LDS copy ----
    for (int v = nnodes - 1; v >= 1; --v) {
        const unsigned sv = lsets[v * 64 + lane];
        unsigned keep = 0;
#pragma unroll
        for (int a = 0; a < N; ++a) keep |= (lrb[v * N + a] & sv) ? 1u << a : 0u;
        lsets[lpar[v] * 64 + lane] &= (unsigned char)keep;
    }
    for (int v = 1; v < nnodes; ++v) {
        const unsigned pv = lsets[lpar[v] * 64 + lane];
        unsigned reach = 0;
#pragma unroll
        for (int b = 0; b < N; ++b) reach |= (lcb[v * N + b] & pv) ? 1u << b : 0u;
        lsets[v * 64 + lane] &= (unsigned char)reach;
    }
    RT_STAMP(2);
    // ---- upward pass: post-order stack program ----
    double lroot[N];
#pragma unroll
    for (int a = 0; a < N; ++a) lroot[a] = 0.0;
    for (int i = 0; i < nnodes; ++i) {
        const int4 op = lops[i];
        const int v = __builtin_amdgcn_readfirstlane(op.x);
        const int up = __builtin_amdgcn_readfirstlane(op.y);
        const int in = up & 255, outs = (up >> 8) & 255;
        const unsigned sv = lsets[v * 64 + lane];
        double l[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double acc = in == 255 ? 1.0 : slots[((size_t)in * N + a) * 64 + lane];
            l[a] = (sv >> a) & 1u ? acc : 0.0;
            Lb[((size_t)v * N + a) * row + site] = l[a];
        }
        if (outs == 255) {
#pragma unroll
            for (int a = 0; a < N; ++a) lroot[a] = l[a];
            continue;
        }
        const double *Pv = PLDS ? lP + (size_t)v * NN : esd + (size_t)v * NN;
        const bool merge = (up >> 16) & 1;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            double sum = 0.0;
#pragma unroll
            for (int b = 0; b < N; ++b) sum = fma(Pv[a * N + b], l[b], sum);
            double *dst = slots + ((size_t)outs * N + a) * 64 + lane;
            *dst = merge ? *dst * sum : sum;
        }
    }
    RT_STAMP(3);
    // ---- downward pass (reverse order) + site sums ----
    const double wt = (live && weights) ? weights[site] : (live ? 1.0 : 0.0);
    double *out = part + (size_t)wave_global * nnodes * NN;
    bool bad = false;
    {
        const int rs = (__builtin_amdgcn_readfirstlane(lops[nnodes - 1].z) >> 8) & 255;
        double w[N], tot = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            w[a] = lroot[a] * (root_distn ? root_distn[a] : 1.0);
            tot += w[a];
        }
        if (!(tot > 0.0)) bad = true;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            const double d = tot > 0.0 ? w[a] / tot : 0.0;
            if (rs != 255) slots[((size_t)rs * N + a) * 64 + lane] = d;
            double r = live ? wt * d : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
#pragma unroll
            for (int b = 0; b < N; ++b)
                if (lane == 0) out[a * N + b] = b == 0 ? r : 0.0;
        }
    }
    RT_STAMP(4);
    constexpr int PF = 4;                     // nodes of L in flight ahead of the one in work
    double pre[PF][N];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int i = nnodes - 2 - j;
        const int v = i >= 0 ? __builtin_amdgcn_readfirstlane(lops[i].x) : 0;
#pragma unroll
        for (int a = 0; a < N; ++a) pre[j][a] = Lb[((size_t)v * N + a) * row + site];
    }
    for (int base = nnodes - 2; base >= 0; base -= PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int i = base - j;
            if (i < 0) break;
            const int4 op = lops[i];
            const int v = __builtin_amdgcn_readfirstlane(op.x);
            const int dn = __builtin_amdgcn_readfirstlane(op.z);
            const int ps = dn & 255, os = (dn >> 8) & 255;
            double lv[N];
#pragma unroll
            for (int a = 0; a < N; ++a) lv[a] = pre[j][a];
            {
                const int nx = i - PF;
                const int vn = nx >= 0 ? __builtin_amdgcn_readfirstlane(lops[nx].x) : 0;
#pragma unroll
                for (int a = 0; a < N; ++a) pre[j][a] = Lb[((size_t)vn * N + a) * row + site];
            }
            const double *Pv = PLDS ? lP + (size_t)v * NN : esd + (size_t)v * NN;
            double u[N];
#pragma unroll
            for (int a = 0; a < N; ++a) {
                double den = 0.0;
#pragma unroll
                for (int b = 0; b < N; ++b) den = fma(Pv[a * N + b], lv[b], den);
                const double pa = slots[((size_t)ps * N + a) * 64 + lane];
                u[a] = 0.0;
                if (pa != 0.0) {
                    if (den > 0.0) u[a] = pa / den;
                    else bad = true;
                }
            }
            if (os != 255) {
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    double acc = 0.0;
#pragma unroll
                    for (int a = 0; a < N; ++a) acc = fma(u[a], Pv[a * N + b], acc);
                    slots[((size_t)os * N + b) * 64 + lane] = acc * lv[b];
                }
            }
            constexpr int P = NN <= 1 ? 1 : NN <= 2 ? 2 : NN <= 4 ? 4 : NN <= 8 ? 8 : NN <= 16 ? 16
                              : NN <= 32 ? 32 : 64;
            constexpr int K = P == 1 ? 0 : P == 2 ? 1 : P == 4 ? 2 : P == 8 ? 3 : P == 16 ? 4
                              : P == 32 ? 5 : 6;
            double x[P];
#pragma unroll
            for (int e = 0; e < P; ++e) x[e] = 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                const double ua = live ? wt * u[a] : 0.0;
#pragma unroll
                for (int b = 0; b < N; ++b) x[a * N + b] = ua * lv[b];
            }
            wave_sum_many<P>(x, lane);
            const int ent = (lane >> (6 - K)) & (P - 1);
            if ((lane & ((1 << (6 - K)) - 1)) == 0 && ent < NN)
                out[(size_t)v * NN + ent] = Pv[ent] != 0.0 ? x[0] : 0.0;
        }
    }
    if (status && live) status[site] = bad ? 2 : 0;
    RT_STAMP(5);
#undef RT_STAMP
}

int expectation_weights_lane(rt_ctx *ctx, int64_t nnodes, int64_t n, int64_t nsites,
        const int64_t *idx, const int64_t *ptr, const double *esd, const double *root_distn,
        const int64_t *state_mask, int64_t nobs, const std::vector<int> &obs_idx, int kind,
        const void *data, const double *site_weights, double *edge_weights, int32_t *status)
{
    const size_t nn = (size_t)n * n, wcount = (size_t)nnodes * nn;
    const long S = (long)((nsites + 63) / 64 * 64);
    const int G = (int)(S / 64);                       // waves = partial sums
    std::vector<int> parent((size_t)nnodes, 0);
    for (int64_t v = 0; v < nnodes; ++v)
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) parent[(size_t)idx[e]] = (int)v;
    std::vector<unsigned char> rowbits((size_t)nnodes * n, 0), colbits((size_t)nnodes * n, 0);
    for (int64_t v = 1; v < nnodes; ++v)
        for (int64_t a = 0; a < n; ++a)
            for (int64_t b = 0; b < n; ++b)
                if (esd[(size_t)v * nn + a * n + b] > 0.0) {
                    rowbits[(size_t)v * n + a] |= (unsigned char)(1u << b);
                    colbits[(size_t)v * n + b] |= (unsigned char)(1u << a);
                }
    const size_t arr = (size_t)nnodes * n * S * 8;
    const size_t ni = (size_t)(nnodes > 1 ? nnodes - 1 : 1);
    // the LDS-resident kernel when its working set fits the default 64 KB of a workgroup
    // (RAOTEH_EXPECT_GLOBAL=1: the global-memory kernel, for A/B runs)
    lane_plan lp;
    size_t lds_bytes = 0;
    bool use_lds = !getenv("RAOTEH_EXPECT_GLOBAL") && build_lane_plan(nnodes, idx, ptr, parent, lp);
    bool p_in_lds = false;
    int waves_per_group = 1;
    if (use_lds) {
        // per wave: slots + state sets; per workgroup: topology (+ P when it fits).  As many
        // waves per workgroup (<= 8) as make the most waves resident on a CU's 160 KB --
        // RAOTEH_EXPECT_WAVES overrides
        const size_t wave_bytes = (size_t)lp.nslots * n * 512 + (((size_t)nnodes * 64 + 15) & ~(size_t)15);
        const size_t topo = (((size_t)nnodes * (16 + 4 + 2 * n)) + 15) & ~(size_t)15;
        const size_t pbytes = ((size_t)nnodes * nn + 1) / 2 * 2 * 8;
        const size_t cap = 160 * 1024;
        if (topo + wave_bytes > cap) use_lds = false;
        else {
            p_in_lds = topo + pbytes + wave_bytes <= cap && pbytes <= 64 * 1024;
            const size_t shared = topo + (p_in_lds ? pbytes : 0);
            int best = 1, best_resident = 0;
            for (int w = 1; w <= 8; ++w) {
                const size_t bytes = shared + w * wave_bytes;
                if (bytes > cap) break;
                const int resident = (int)(cap / bytes) * w;
                if (resident > best_resident) { best_resident = resident; best = w; }
            }
            if (const char *v = getenv("RAOTEH_EXPECT_WAVES")) {
                const int w = atoi(v);
                if (w >= 1 && w <= 8 && shared + w * wave_bytes <= cap) best = w;
            }
            waves_per_group = best;
            lds_bytes = shared + best * wave_bytes;
        }
    }
    scratch_plan plan;
    const size_t o_ops = plan.take((size_t)nnodes * 16), o_trace = plan.take(64);
    const size_t o_idx = plan.take(ni * 8), o_ptr = plan.take((size_t)(nnodes + 1) * 8);
    const size_t o_esd = plan.take(wcount * 8), o_par = plan.take((size_t)nnodes * 4);
    const size_t o_rb = plan.take((size_t)nnodes * n), o_cb = plan.take((size_t)nnodes * n);
    const size_t o_sets = plan.take((size_t)nnodes * S);
    const size_t o_M = plan.take(use_lds ? 8 : arr), o_L = plan.take(arr);
    const size_t o_D = plan.take(use_lds ? 8 : arr);
    const size_t o_root = plan.take((size_t)n * 8), o_w = plan.take((size_t)nsites * 8);
    const size_t o_st = plan.take((size_t)S * 4);
    const size_t o_part = plan.take((size_t)G * wcount * 8), o_out = plan.take(wcount * 8);
    const size_t mask_bytes = state_mask ? (size_t)nsites * nnodes * n * 8 : 0;
    const size_t data_bytes = state_mask ? 0 : (size_t)nsites * nobs * (kind == RT_OBS_STATE ? 1 : 8);
    const size_t o_data = plan.take(std::max<size_t>(std::max(data_bytes, mask_bytes), 8));
    const size_t o_obsn = plan.take(std::max<size_t>((size_t)nobs * 4, 8));
    RT_TRY(scratch_reserve(ctx, plan.total));
    unsigned char *base = ctx->d_scratch;
    long *d_idx = (long *)(base + o_idx), *d_ptr = (long *)(base + o_ptr);
    double *d_esd = (double *)(base + o_esd);
    int *d_par = (int *)(base + o_par), *d_st = (int *)(base + o_st);
    unsigned char *d_rb = base + o_rb, *d_cb = base + o_cb, *d_sets = base + o_sets;
    double *d_M = (double *)(base + o_M), *d_L = (double *)(base + o_L), *d_D = (double *)(base + o_D);
    double *d_root = root_distn ? (double *)(base + o_root) : nullptr;
    double *d_w = site_weights ? (double *)(base + o_w) : nullptr;
    double *d_part = (double *)(base + o_part), *d_out = (double *)(base + o_out);
    hipStream_t st = ctx->stream;
    int4 *d_ops = (int4 *)(base + o_ops);
    // RAOTEH_EXPECT_TRACE=1: shader-clock stamps of workgroup 0 at the phase boundaries
    unsigned long long *d_trace =
        (use_lds && getenv("RAOTEH_EXPECT_TRACE")) ? (unsigned long long *)(base + o_trace) : nullptr;
    if (use_lds)
        RT_HIP(hipMemcpyAsync(d_ops, lp.ops.data(), (size_t)nnodes * 16, hipMemcpyHostToDevice, st));
    if (nnodes > 1)
        RT_HIP(hipMemcpyAsync(d_idx, idx, (size_t)(nnodes - 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_ptr, ptr, (size_t)(nnodes + 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_esd, esd, wcount * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_par, parent.data(), (size_t)nnodes * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_rb, rowbits.data(), rowbits.size(), hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_cb, colbits.data(), colbits.size(), hipMemcpyHostToDevice, st));
    if (d_root) RT_HIP(hipMemcpyAsync(d_root, root_distn, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (d_w) RT_HIP(hipMemcpyAsync(d_w, site_weights, (size_t)nsites * 8, hipMemcpyHostToDevice, st));
    void *d_data = base + o_data;
    int *d_obsn = (int *)(base + o_obsn);
    if (state_mask) {
        RT_HIP(hipMemcpyAsync(d_data, state_mask, mask_bytes, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sets_from_mask_kernel, dim3(2048), dim3(256), 0, st, (int)nnodes, (int)n,
                           (long)nsites, S, (const long *)d_data, d_sets);
    } else {
        if (data_bytes) RT_HIP(hipMemcpyAsync(d_data, data, data_bytes, hipMemcpyHostToDevice, st));
        if (nobs)
            RT_HIP(hipMemcpyAsync(d_obsn, obs_idx.data(), (size_t)nobs * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sets_fill_kernel, dim3(2048), dim3(256), 0, st, (long)nnodes * S, (int)n,
                           d_sets);
        if (nobs)
            hipLaunchKernelGGL(sets_apply_obs_kernel, dim3(2048), dim3(256), 0, st, (int)n,
                               (long)nsites, S, (int)nobs, d_obsn, kind, d_data, d_sets);
    }
#define RT_EXPECT_LDS_ONE(NV, PL)                                                            \
    do {                                                                                      \
        if (lds_bytes > 64 * 1024)                                                            \
            RT_HIP(hipFuncSetAttribute((const void *)expect_lane_lds_kernel<NV, PL>,          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                       (int)lds_bytes));                                      \
        hipLaunchKernelGGL((expect_lane_lds_kernel<NV, PL>),                                  \
                           dim3((unsigned)((G + waves_per_group - 1) / waves_per_group)),     \
                           dim3(64 * waves_per_group), lds_bytes, st, (int)nnodes,            \
                           (long)nsites, S, lp.nslots, d_ops, d_par, d_esd, d_rb, d_cb, d_root, \
                           d_w, d_sets, d_L, d_part, d_st, d_trace);                          \
    } while (0)
#define RT_EXPECT_LDS(NV)                                                                    \
    do {                                                                                      \
        if (p_in_lds) RT_EXPECT_LDS_ONE(NV, true);                                            \
        else RT_EXPECT_LDS_ONE(NV, false);                                                    \
    } while (0)
#define RT_EXPECT_LANE(NV)                                                                   \
    if (use_lds) RT_EXPECT_LDS(NV);                                                          \
    else hipLaunchKernelGGL(expect_lane_kernel<NV>, dim3((unsigned)G), dim3(64), 0, st, (int)nnodes, \
                       (long)nsites, S, d_par, d_idx, d_ptr, d_esd, d_rb, d_cb, d_root, d_w,   \
                       d_sets, d_M, d_L, d_D, d_part, d_st)
    switch ((int)n) {
    case 1: RT_EXPECT_LANE(1); break;
    case 2: RT_EXPECT_LANE(2); break;
    case 3: RT_EXPECT_LANE(3); break;
    case 4: RT_EXPECT_LANE(4); break;
    case 5: RT_EXPECT_LANE(5); break;
    case 6: RT_EXPECT_LANE(6); break;
    case 7: RT_EXPECT_LANE(7); break;
    default: RT_EXPECT_LANE(8); break;
    }
#undef RT_EXPECT_LANE
#undef RT_EXPECT_LDS
#undef RT_EXPECT_LDS_ONE
    hipLaunchKernelGGL(sum_parts_wide_kernel, dim3((unsigned)((wcount + 3) / 4)), dim3(256), 0,
                       st, G, (long)wcount, d_part, d_out);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(edge_weights, d_out, wcount * 8, hipMemcpyDeviceToHost, st));
    if (status) RT_HIP(hipMemcpyAsync(status, d_st, (size_t)nsites * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    if (d_trace) {
        unsigned long long t[6];
        RT_HIP(hipMemcpy(t, d_trace, sizeof t, hipMemcpyDeviceToHost));
        fprintf(stderr, "[raoteh_amd] expect trace (clocks): fill %llu, sets %llu, up %llu, root %llu, "
                "down %llu; %d slots, %d waves per workgroup, %zu B of LDS\n", t[1] - t[0],
                t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], lp.nslots, waves_per_group,
                lds_bytes);
    }
    return RT_OK;
}

// ---- n <= 8 on a RESIDENT batch (rt_expect_step) -------------------------------------------
// The same fused lane-per-site kernel, with everything it needs already on the device: the
// transition matrices of the model (their zero patterns are read off by a kernel), the
// allowed sets of the batch (built once per batch from its resident layout), the stack
// program of the tree (once per model).

struct expect_lane_state_t {
    lane_plan lp;
    bool p_in_lds = false;
    int waves_per_group = 1;
    size_t lds_bytes = 0;
    int4 *d_ops = nullptr;
    int *d_par = nullptr, *d_node_of_k = nullptr;
    unsigned char *d_rb = nullptr, *d_cb = nullptr;
    ~expect_lane_state_t()
    {
        hipFree(d_ops); hipFree(d_par); hipFree(d_rb); hipFree(d_cb);
    }
};

// rowbits[v][a] bit b / colbits[v][b] bit a: P_v[a][b] > 0
__global__ void __launch_bounds__(64)
pattern_bits_kernel(int n, const double *__restrict__ esd, unsigned char *__restrict__ rowbits,
                    unsigned char *__restrict__ colbits)
{
    const int v = blockIdx.x, t = threadIdx.x;
    if (t >= n) return;
    unsigned rb = 0, cb = 0;
    if (v > 0)
        for (int k = 0; k < n; ++k) {
            if (esd[((size_t)v * n + t) * n + k] > 0.0) rb |= 1u << k;
            if (esd[((size_t)v * n + k) * n + t] > 0.0) cb |= 1u << k;
        }
    rowbits[(size_t)v * n + t] = (unsigned char)rb;
    colbits[(size_t)v * n + t] = (unsigned char)cb;
}

// allowed sets of the observed nodes from the lane family's resident layouts: dense
// [block][k][pair][lane][2] (a state is allowed where its likelihood is not zero), or one
// byte per leaf (compact == 1: a state, 255 = unobserved; compact == 2: the set itself)
__global__ void __launch_bounds__(256)
sets_from_lane_batch_kernel(int n, long nsites, long S, int K, int block_sites, int compact,
                            const void *__restrict__ obs, const int *__restrict__ node_of_k,
                            unsigned char *__restrict__ sets)
{
    const long total = nsites * K;
    const unsigned full = (1u << n) - 1u;
    const int hp = ((n + 1) & ~1) / 2, KQ = (K + 3) / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % K);
        const long site = i / K;
        const long blk = site / block_sites;
        const int lane = (int)(site - blk * block_sites);
        unsigned m = 0;
        if (compact) {
            const unsigned w = ((const unsigned *)obs)[((size_t)blk * KQ + (k >> 2)) * block_sites + lane];
            const unsigned b = (w >> (8 * (k & 3))) & 255u;
            m = compact == 2 ? (b & full) : (b >= (unsigned)n ? full : 1u << b);
        } else {
            const double *o = (const double *)obs + (((size_t)blk * K + k) * hp * block_sites + lane) * 2;
            for (int st = 0; st < n; ++st)
                if (o[(size_t)(st >> 1) * block_sites * 2 + (st & 1)] != 0.0) m |= 1u << st;
        }
        sets[(size_t)node_of_k[k] * S + site] = (unsigned char)m;
    }
}

}  // namespace

void rt_expect_lane_release(rt_model *m)
{
    if (!m || !m->expect_lane_state) return;
    delete (expect_lane_state_t *)m->expect_lane_state;
    m->expect_lane_state = nullptr;
}

// W f64[nnodes][n][n] (device; slot 0, column 0: the weighted root posterior sums) and the
// per-site status for a resident lane-family batch; asynchronous on the context's stream
int rt_expect_lane_resident(rt_model *m, rt_sites *s, double *d_W, int *d_status)
{
    rt_ctx *ctx = m->ctx;
    const int64_t n = m->n, nnodes = m->nnodes, nsites = s->nsites;
    RT_REQUIRE(n <= 8 && s->layout == RT_LAYOUT_LANE && !s->d_scratch,
               "not a resident lane-family batch");
    hipStream_t st = ctx->stream;
    const size_t nn = (size_t)n * n, wcount = (size_t)nnodes * nn;
    expect_lane_state_t *ls = (expect_lane_state_t *)m->expect_lane_state;
    if (!ls) {
        std::unique_ptr<expect_lane_state_t> fresh(new (std::nothrow) expect_lane_state_t());
        if (!fresh) return RT_ERR_NOMEM;
        if (!build_lane_plan(nnodes, m->indices.data(), m->indptr.data(), m->parent, fresh->lp)) {
            rt_set_error("rt_expect_step: no LDS stack program for this tree");
            return RT_ERR_UNSUPPORTED;
        }
        const size_t wave_bytes = (size_t)fresh->lp.nslots * n * 512 + (((size_t)nnodes * 64 + 15) & ~(size_t)15);
        const size_t topo = (((size_t)nnodes * (16 + 4 + 2 * n)) + 15) & ~(size_t)15;
        const size_t pbytes = ((size_t)nnodes * nn + 1) / 2 * 2 * 8;
        const size_t cap = 160 * 1024;
        if (topo + wave_bytes > cap) {
            rt_set_error("rt_expect_step: the tree does not fit the LDS of the fused kernel");
            return RT_ERR_UNSUPPORTED;
        }
        fresh->p_in_lds = topo + pbytes + wave_bytes <= cap && pbytes <= 64 * 1024;
        const size_t shared = topo + (fresh->p_in_lds ? pbytes : 0);
        int best = 1, best_resident = 0;
        for (int w = 1; w <= 8; ++w) {
            const size_t bytes = shared + w * wave_bytes;
            if (bytes > cap) break;
            const int resident = (int)(cap / bytes) * w;
            if (resident > best_resident) { best_resident = resident; best = w; }
        }
        fresh->waves_per_group = best;
        fresh->lds_bytes = shared + best * wave_bytes;
        RT_HIP(hipMalloc((void **)&fresh->d_ops, (size_t)nnodes * 16));
        RT_HIP(hipMalloc((void **)&fresh->d_par, (size_t)nnodes * 4));
        RT_HIP(hipMalloc((void **)&fresh->d_rb, (size_t)nnodes * n));
        RT_HIP(hipMalloc((void **)&fresh->d_cb, (size_t)nnodes * n));
        std::vector<int> par(m->parent.begin(), m->parent.end());
        par[0] = 0;
        RT_HIP(hipMemcpy(fresh->d_ops, fresh->lp.ops.data(), (size_t)nnodes * 16, hipMemcpyHostToDevice));
        RT_HIP(hipMemcpy(fresh->d_par, par.data(), (size_t)nnodes * 4, hipMemcpyHostToDevice));
        m->expect_lane_state = fresh.release();
        ls = (expect_lane_state_t *)m->expect_lane_state;
    }
    const long S = (long)((nsites + 63) / 64 * 64);
    const int G = (int)(S / 64);
    // the allowed sets of the batch, once
    if (!s->d_sets) {
        const int K = (int)s->nobs;
        std::vector<int> node_of_k((size_t)std::max(K, 1), 0);
        for (const rt_op &op : s->ops)
            if (op.obs >= 0) node_of_k[(size_t)op.obs] = op.node;
        int *d_nk = nullptr;
        RT_HIP(hipMalloc((void **)&s->d_sets, (size_t)nnodes * S));
        RT_HIP(hipMalloc((void **)&d_nk, node_of_k.size() * 4));
        RT_HIP(hipMemcpyAsync(d_nk, node_of_k.data(), node_of_k.size() * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sets_fill_kernel, dim3(2048), dim3(256), 0, st, (long)nnodes * S, (int)n,
                           s->d_sets);
        if (K > 0)
            hipLaunchKernelGGL(sets_from_lane_batch_kernel, dim3(2048), dim3(256), 0, st, (int)n,
                               (long)nsites, S, K, s->block_sites, s->compact_states,
                               (const void *)s->d_obs, d_nk, s->d_sets);
        RT_HIP(hipGetLastError());
        RT_HIP(hipStreamSynchronize(st));
        hipFree(d_nk);
    }
    const size_t arr = (size_t)nnodes * n * S * 8;
    scratch_plan plan;
    const size_t o_L = plan.take(arr), o_part = plan.take((size_t)G * wcount * 8);
    RT_TRY(scratch_reserve(ctx, plan.total));
    double *d_L = (double *)(ctx->d_scratch + o_L), *d_part = (double *)(ctx->d_scratch + o_part);
    hipLaunchKernelGGL(pattern_bits_kernel, dim3((unsigned)nnodes), dim3(64), 0, st, (int)n, m->d_P,
                       ls->d_rb, ls->d_cb);
    const size_t lds_bytes = ls->lds_bytes;
    const int wpg = ls->waves_per_group;
#define RT_EXPECT_RES_ONE(NV, PL)                                                            \
    do {                                                                                      \
        if (lds_bytes > 64 * 1024)                                                            \
            RT_HIP(hipFuncSetAttribute((const void *)expect_lane_lds_kernel<NV, PL>,          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                       (int)lds_bytes));                                      \
        hipLaunchKernelGGL((expect_lane_lds_kernel<NV, PL>), dim3((unsigned)((G + wpg - 1) / wpg)), \
                           dim3(64 * wpg), lds_bytes, st, (int)nnodes, (long)nsites, S,       \
                           ls->lp.nslots, ls->d_ops, ls->d_par, m->d_P, ls->d_rb, ls->d_cb,   \
                           m->d_root, s->d_weights, s->d_sets, d_L, d_part, d_status,         \
                           (unsigned long long *)nullptr);                                    \
    } while (0)
#define RT_EXPECT_RES(NV)                                                                    \
    do {                                                                                      \
        if (ls->p_in_lds) RT_EXPECT_RES_ONE(NV, true);                                        \
        else RT_EXPECT_RES_ONE(NV, false);                                                    \
    } while (0)
    switch ((int)n) {
    case 1: RT_EXPECT_RES(1); break;
    case 2: RT_EXPECT_RES(2); break;
    case 3: RT_EXPECT_RES(3); break;
    case 4: RT_EXPECT_RES(4); break;
    case 5: RT_EXPECT_RES(5); break;
    case 6: RT_EXPECT_RES(6); break;
    case 7: RT_EXPECT_RES(7); break;
    default: RT_EXPECT_RES(8); break;
    }
#undef RT_EXPECT_RES
#undef RT_EXPECT_RES_ONE
    hipLaunchKernelGGL(sum_parts_wide_kernel, dim3((unsigned)((wcount + 3) / 4)), dim3(256), 0, st, G,
                       (long)wcount, d_part, d_W);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

namespace {

int expectation_weights_impl(rt_ctx *ctx, int64_t nnodes, int64_t n, int64_t nsites,
        const int64_t *idx, const int64_t *ptr, const double *esd, const double *root_distn,
        const int64_t *state_mask, int64_t nobs, const int64_t *obs_nodes, int kind,
        const void *data, const double *site_weights, double *edge_weights, int32_t *status)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(check_tree(nnodes, n, nsites, idx, ptr, esd));
    if (n > RT_MAX_EXPECT_STATES) {
        rt_set_error("expectation path: n=%lld > %d", (long long)n, RT_MAX_EXPECT_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    RT_REQUIRE(edge_weights && (state_mask || data || nobs == 0 || nsites == 0), "null array");
    std::vector<int> obs_idx;
    if (!state_mask) {
        RT_REQUIRE(kind == RT_OBS_STATE || kind == RT_OBS_MASK,
                   "kind must be RT_OBS_STATE or RT_OBS_MASK");
        RT_REQUIRE(nobs >= 0 && (obs_nodes || nobs == 0), "bad observation list");
        std::vector<char> seen((size_t)nnodes, 0);
        for (int64_t j = 0; j < nobs; ++j) {
            RT_REQUIRE(obs_nodes[j] >= 0 && obs_nodes[j] < nnodes, "obs_nodes out of range");
            RT_REQUIRE(!seen[(size_t)obs_nodes[j]], "node %lld observed twice",
                       (long long)obs_nodes[j]);
            seen[(size_t)obs_nodes[j]] = 1;
            obs_idx.push_back((int)obs_nodes[j]);
        }
    }
    RT_HIP(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n, wcount = (size_t)nnodes * nn;
    if (nsites == 0) {
        memset(edge_weights, 0, wcount * 8);
        return RT_OK;
    }
    // 8 < n <= 64, compact observations: two pruning-shaped passes and the site sums on the
    // matrix pipe (expect_mfma.hip); RT_ERR_UNSUPPORTED = not its case, go on below
    if (!state_mask) {
        const int rc = rt_expectation_weights_mfma(ctx, nnodes, n, nsites, idx, ptr, esd, root_distn,
                                                   nobs, obs_nodes, kind, data, site_weights,
                                                   edge_weights, status);
        if (rc != RT_ERR_UNSUPPORTED) return rc;
    }
    // n <= 8: the fused lane-per-site kernel on the resident layout
    // (RAOTEH_EXPECT_LEGACY=1: the per-pass kernels below, for A/B runs)
    if (n <= 8 && !getenv("RAOTEH_EXPECT_LEGACY"))
        return expectation_weights_lane(ctx, nnodes, n, nsites, idx, ptr, esd, root_distn,
                                        state_mask, nobs, obs_idx, kind, data, site_weights,
                                        edge_weights, status);
    std::vector<int> parent((size_t)nnodes, 0);
    for (int64_t v = 0; v < nnodes; ++v)
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) parent[(size_t)idx[e]] = (int)v;
    const int G = (int)std::min<int64_t>(nsites, 64);
    const size_t bytes = (size_t)nsites * nnodes * n * 8;
    const size_t ni = (size_t)(nnodes > 1 ? nnodes - 1 : 1);
    scratch_plan plan;
    const size_t o_idx = plan.take(ni * 8), o_ptr = plan.take((size_t)(nnodes + 1) * 8);
    const size_t o_esd = plan.take(wcount * 8), o_par = plan.take((size_t)nnodes * 4);
    const size_t o_mask = plan.take(bytes), o_pmap = plan.take(bytes), o_distn = plan.take(bytes);
    const size_t o_root = plan.take((size_t)n * 8), o_w = plan.take((size_t)nsites * 8);
    const size_t o_st = plan.take((size_t)nsites * 4);
    const size_t o_part = plan.take((size_t)G * wcount * 8), o_out = plan.take(wcount * 8);
    const size_t data_bytes = state_mask ? 0 : (size_t)nsites * nobs * (kind == RT_OBS_STATE ? 1 : 8);
    const size_t o_data = plan.take(std::max<size_t>(data_bytes, 8));
    const size_t o_obsn = plan.take(std::max<size_t>((size_t)nobs * 4, 8));
    RT_TRY(scratch_reserve(ctx, plan.total));
    unsigned char *base = ctx->d_scratch;
    long *d_idx = (long *)(base + o_idx), *d_ptr = (long *)(base + o_ptr);
    double *d_esd = (double *)(base + o_esd);
    int *d_par = (int *)(base + o_par), *d_st = (int *)(base + o_st);
    long *d_mask = (long *)(base + o_mask);
    double *d_pmap = (double *)(base + o_pmap), *d_distn = (double *)(base + o_distn);
    double *d_root = root_distn ? (double *)(base + o_root) : nullptr;
    double *d_w = site_weights ? (double *)(base + o_w) : nullptr;
    double *d_part = (double *)(base + o_part), *d_out = (double *)(base + o_out);
    hipStream_t st = ctx->stream;
    if (nnodes > 1)
        RT_HIP(hipMemcpyAsync(d_idx, idx, (size_t)(nnodes - 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_ptr, ptr, (size_t)(nnodes + 1) * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_esd, esd, wcount * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_par, parent.data(), (size_t)nnodes * 4, hipMemcpyHostToDevice, st));
    if (state_mask) {
        RT_HIP(hipMemcpyAsync(d_mask, state_mask, bytes, hipMemcpyHostToDevice, st));
    } else {
        void *d_data = base + o_data;
        int *d_obsn = (int *)(base + o_obsn);
        if (data_bytes) RT_HIP(hipMemcpyAsync(d_data, data, data_bytes, hipMemcpyHostToDevice, st));
        if (nobs)
            RT_HIP(hipMemcpyAsync(d_obsn, obs_idx.data(), (size_t)nobs * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(expand_obs_kernel, dim3(2048), dim3(256), 0, st, (int)nnodes, (int)n,
                           (long)nsites, (int)nobs, d_obsn, kind, d_data, d_mask);
        if (nobs)
            hipLaunchKernelGGL(apply_obs_kernel, dim3(2048), dim3(256), 0, st, (int)nnodes,
                               (int)n, (long)nsites, (int)nobs, d_obsn, kind, d_data, d_mask);
    }
    if (d_root) RT_HIP(hipMemcpyAsync(d_root, root_distn, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (d_w) RT_HIP(hipMemcpyAsync(d_w, site_weights, (size_t)nsites * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pset_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_mask);
    hipLaunchKernelGGL(set_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_mask);
    hipLaunchKernelGGL(pmap_kernel, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_mask, (const double *)nullptr, d_pmap);
    hipLaunchKernelGGL(distn_kernel<false>, dim3(pass_grid(nsites, n)), dim3(pass_block(n)), 0, st, (int)nnodes, (int)n,
                       (long)nsites, d_idx, d_ptr, d_esd, d_root, d_pmap, d_distn,
                       (double *)nullptr, d_st);
    if (nn <= 128)
        hipLaunchKernelGGL(ratio_sum_small_kernel, dim3((unsigned)nnodes, (unsigned)G), dim3(256),
                           0, st, (int)nnodes, (int)n, (int)nsites, d_par, d_esd, d_pmap, d_distn,
                           d_w, d_part);
    else
        hipLaunchKernelGGL(ratio_sum_kernel, dim3((unsigned)nnodes, (unsigned)G), dim3(256), 0,
                           st, (int)nnodes, (int)n, (int)nsites, d_par, d_esd, d_pmap, d_distn,
                           d_w, d_part);
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((wcount + 255) / 256)), dim3(256), 0,
                       st, G, (long)wcount, d_part, d_out);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(edge_weights, d_out, wcount * 8, hipMemcpyDeviceToHost, st));
    if (status) RT_HIP(hipMemcpyAsync(status, d_st, (size_t)nsites * 4, hipMemcpyDeviceToHost, st));
    // parent.data(), obs_idx.data() and the caller's arrays must outlive the copies
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

}  // namespace

extern "C" int rt_mjp_esd_expectation_weights(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        const double *root_distn, const int64_t *state_mask, const double *site_weights,
        double *edge_weights, int32_t *status)
{
    RT_REQUIRE(state_mask || nsites == 0, "null array");
    return expectation_weights_impl(ctx, nnodes, n, nsites, idx, ptr, esd, root_distn, state_mask,
                                    0, nullptr, RT_OBS_MASK, nullptr, site_weights, edge_weights,
                                    status);
}

extern "C" int rt_mjp_esd_expectation_weights_obs(rt_ctx *ctx, int64_t nnodes, int64_t n,
        int64_t nsites, const int64_t *idx, const int64_t *ptr, const double *esd,
        const double *root_distn, int64_t nobs, const int64_t *obs_nodes, int kind,
        const void *data, const double *site_weights, double *edge_weights, int32_t *status)
{
    RT_REQUIRE(data || nobs == 0 || nsites == 0, "null array");
    return expectation_weights_impl(ctx, nnodes, n, nsites, idx, ptr, esd, root_distn, nullptr,
                                    nobs, obs_nodes, kind, data, site_weights, edge_weights,
                                    status);
}
