// Two more pyfelscore entry points the reference calls, next to the likelihood path:
//
//   pyfelscore.get_lb_transition_matrix(t, Q, P)     examples/p53/liwen.py:45
//       (pure-Python twin getp_lb, liwen.py:47-82): a lower bound of expm(Q t) that counts
//       the histories with at most one change per entry -- element-wise closed forms.
//   pyfelscore.tmjp_get_inhomogeneous_mjp(...)       raoteh/sampler/_tmjp_dense.py:1039-1054
//       (sparse twin _tmjp.get_inhomogeneous_mjp, _tmjp.py:863-900): the 3-state tolerance
//       rate matrix of every edge of a primary trajectory, and which tolerance states its
//       endpoints may take.  Index bookkeeping: host code.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace {

// P[b][sa][sb], one thread per entry; Q row-major, diagonal = minus the row sum
__global__ void __launch_bounds__(256)
lb_transition_kernel(int n, long count, const double *__restrict__ Q, const double *__restrict__ t,
                     double *__restrict__ P)
{
    const long total = count * n * n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / ((long)n * n);
        const int r = (int)(e - b * n * n);
        const int sa = r / n, sb = r - sa * n;
        const double tt = t[b];
        double p;
        if (sa == sb) {
            p = exp(tt * Q[(long)sa * n + sa]);           // no change in the interval (:56-58)
        } else {
            const double rab = Q[(long)sa * n + sb];
            if (rab != 0.0) {
                // one change, of this type: the integral over its time x of
                // exp(-ra x) rab exp(-rb (t - x))  (:59-77)
                const double ra = -Q[(long)sa * n + sa], rb = -Q[(long)sb * n + sb];
                if (ra == rb) p = rab * tt * exp(-rb * tt);
                else p = rab * ((exp(-ra * tt) - exp(-rb * tt)) / (rb - ra));
            } else {
                p = 0.0;
            }
        }
        P[e] = p;
    }
}

}  // namespace

extern "C" int rt_lb_transition_matrix(rt_ctx *ctx, int64_t n, int64_t count, const double *Q,
                                       const double *t, double *P)
{
    RT_REQUIRE(ctx, "null context");
    RT_REQUIRE(n >= 1 && count >= 0 && Q && t && P, "bad arguments");
    if (count == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_Q = 0, o_t = o_Q + up(nn * 8), o_P = o_t + up(count * 8),
                 total = o_P + up(count * nn * 8);
    RT_TRY(rt_scratch_reserve(ctx, total));
    double *dQ = (double *)(ctx->d_scratch + o_Q), *dt = (double *)(ctx->d_scratch + o_t),
           *dP = (double *)(ctx->d_scratch + o_P);
    hipStream_t st = ctx->stream;
    RT_HIP(hipMemcpyAsync(dQ, Q, nn * 8, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(dt, t, count * 8, hipMemcpyHostToDevice, st));
    const size_t entries = (size_t)count * nn;
    const unsigned grid = (unsigned)std::min<size_t>((entries + 255) / 256, 4096);
    hipLaunchKernelGGL(lb_transition_kernel, dim3(grid), dim3(256), 0, st, (int)n, (long)count, dQ,
                       dt, dP);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(P, dP, count * nn * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

extern "C" int rt_tmjp_get_inhomogeneous_mjp(int64_t nnodes, const int64_t *idx, const int64_t *ptr,
        const int64_t *edge_to_primary_state, int64_t nprimary, const int64_t *primary_to_part,
        const double *Q_primary, double rate_on, double rate_off, int64_t tolerance_class,
        int64_t *node_to_allowed_tolerances, double *tol_rate_matrices)
{
    RT_REQUIRE(nnodes >= 1 && ptr && (idx || nnodes == 1) && edge_to_primary_state &&
               primary_to_part && Q_primary && node_to_allowed_tolerances && tol_rate_matrices &&
               nprimary >= 1, "bad arguments");
    RT_REQUIRE(ptr[0] == 0 && ptr[nnodes] == nnodes - 1, "tree_csr_indptr does not describe a tree");
    for (int k = 0; k < 9; ++k) tol_rate_matrices[k] = 0.0;       // the root's slot: no edge above it
    for (int64_t a = 0; a < nnodes; ++a) {
        RT_REQUIRE(ptr[a + 1] >= ptr[a], "tree_csr_indptr not monotone");
        for (int64_t e = ptr[a]; e < ptr[a + 1]; ++e) {
            const int64_t b = idx[e];
            RT_REQUIRE(b > 0 && b < nnodes, "child index out of range");
            const int64_t s = edge_to_primary_state[b];
            RT_REQUIRE(s >= 0 && s < nprimary, "primary state %lld of the edge above node %lld",
                       (long long)s, (long long)b);
            const bool own = primary_to_part[s] == tolerance_class;
            // _tmjp.py:872-884: the class of the current primary state cannot be switched off;
            // absorption = the rate of leaving towards a state of the class under consideration
            const double off = own ? 0.0 : rate_off;
            double absorb = 0.0;
            for (int64_t sb = 0; sb < nprimary; ++sb)
                if (sb != s && primary_to_part[sb] == tolerance_class)
                    absorb += Q_primary[s * nprimary + sb];
            double *M = tol_rate_matrices + b * 9;
            M[0] = -rate_on; M[1] = rate_on;         M[2] = 0.0;
            M[3] = off;      M[4] = -(off + absorb); M[5] = absorb;
            M[6] = 0.0;      M[7] = 0.0;             M[8] = 0.0;
            if (own) {                                // :897-900
                node_to_allowed_tolerances[a * 2] = 0;
                node_to_allowed_tolerances[b * 2] = 0;
            }
        }
    }
    return RT_OK;
}
