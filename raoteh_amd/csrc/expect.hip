// Expected history statistics, the part after the passes: per edge ONE Frechet
// derivative of the matrix exponential, contracted on the device.
//
// Reference: _mjp_dense.get_expected_history_statistics (_mjp_dense.py:410-539) calls
// scipy.linalg.expm_frechet n + nnz(Q) times per edge and site, once per direction E_cd,
// and contracts each result with W = J / P (J the joint endpoint posterior of the edge).
// By the adjoint identity <W, L(A, E)> = <L(A^T, W), E> all of them come from one
// derivative per edge:
//     M_e = L(t_e Q_e^T, W_e)      dwell[c] += t_e M_e[c][c]
//                                  trans[c][d] += t_e Q_e[c][d] M_e[c][d]   (Q_e[c][d] != 0)
// and L(A, W) is the upper right block of expm([[A, W], [0, A]]).  Here: a kernel
// assembles the 2n x 2n blocks (W scaled to unit size so that it does not drive the
// scaling and squaring), the Taylor expm kernel (order <= 128: n <= 64, the codon model
// included) exponentiates all edges in one launch, and a kernel contracts the corner
// blocks over the edges in a fixed order.  2n + n^2 numbers leave the device.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
frechet_assemble_kernel(int n, const double *__restrict__ Q, const int *__restrict__ qidx,
                        const double *__restrict__ t, const double *__restrict__ W,
                        double *__restrict__ blocks, double *__restrict__ scale)
{
    __shared__ double smax[256];
    const int e = blockIdx.x;
    const int nn = n * n, m = 2 * n;
    const double *We = W + (long)e * nn;
    const double *Qe = Q + (long)qidx[e] * nn;
    double mx = 0.0;
    for (int k = threadIdx.x; k < nn; k += 256) mx = fmax(mx, fabs(We[k]));
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + w]);
        __syncthreads();
    }
    double sc = smax[0];
    if (!(sc > 0.0) || !(sc < 1e308 * 10.0)) sc = 1.0;
    // ... and a further 2^-17 (exact): the corner block is linear in W, so its size is ours to
    // choose, and at max |entry| = 1 its column sums (up to n) decided the number of squarings
    // of the whole block -- seven at 61 states where t Q alone needs none (the series for the
    // corner has the tail of exp's one degree down, times |W|: its relative accuracy does not
    // depend on the scale).  With column sums <= 2^-11 the order comes from t Q.
    sc = ldexp(sc, 17);
    if (!(sc < 1e308 * 10.0)) sc = ldexp(sc, -17);
    if (threadIdx.x == 0) scale[e] = sc;
    const double te = t[e], inv = 1.0 / sc;
    double *B = blocks + (long)e * m * m;
    for (int k = threadIdx.x; k < m * m; k += 256) {
        const int r = k / m, c = k - r * m;
        double v = 0.0;
        if (r < n && c < n) v = te * Qe[c * n + r];                     // t Q^T
        else if (r >= n && c >= n) v = te * Qe[(c - n) * n + (r - n)];
        else if (r < n) v = We[r * n + (c - n)] * inv;                  // W / |W|
        B[k] = v;
    }
}

// one thread per (c, d): edges added in index order (fixed rounding)
__global__ void __launch_bounds__(256)
frechet_contract_kernel(int n, int nedges, const double *__restrict__ Q,
                        const int *__restrict__ qidx, const double *__restrict__ t,
                        const double *__restrict__ E, const double *__restrict__ scale,
                        double *__restrict__ dwell, double *__restrict__ trans)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int nn = n * n, m = 2 * n;
    if (k >= nn) return;
    const int c = k / n, d = k - c * n;
    double acc = 0.0, dw = 0.0;
    // (unrolled: the loads of eight edges in flight; the additions keep their order)
#pragma unroll 8
    for (int e = 0; e < nedges; ++e) {
        const double q = Q[(long)qidx[e] * nn + k];
        const double Mcd = E[(long)e * m * m + (long)c * m + (n + d)] * scale[e];
        if (c == d) dw += t[e] * Mcd;
        if (q != 0.0) acc += t[e] * q * Mcd;
    }
    trans[k] = acc;
    if (c == d) dwell[c] = dw;
}

}  // namespace

// The three launches on device-resident operands, asynchronously on the context's stream:
// Q [nq][n][n], qidx / t / W per edge, scratch B and E [nedges][2n][2n], scale [nedges],
// ones f64[nedges] and ident int32[nedges] (0, 1, 2, ...); dwell [n] and trans [n][n] out.
int rt_frechet_statistics_device(rt_ctx *ctx, int64_t n, int64_t nedges, const double *dQ,
                                 const int32_t *dqidx, const double *dt, const double *dW,
                                 double *dB, double *dE, double *dscale, const double *dones,
                                 const int32_t *dident, double *ddwell, double *dtrans)
{
    hipStream_t st = ctx->stream;
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(frechet_assemble_kernel, dim3((unsigned)nedges), dim3(256), 0, st, (int)n, dQ,
                       dqidx, dt, dW, dB, dscale);
    RT_HIP(hipGetLastError());
    RT_TRY(rt_launch_expm(ctx, 2 * n, nedges, dB, dident, dones, dE, nullptr, nullptr, 0, nullptr));
    hipLaunchKernelGGL(frechet_contract_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st,
                       (int)n, (int)nedges, dQ, dqidx, dt, dE, dscale, ddwell, dtrans);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_mjp_frechet_statistics(rt_ctx *ctx, int64_t n, int64_t nedges, const double *Q,
                                         int64_t nq, const int64_t *q_index, const double *t,
                                         const double *W, double *dwell, double *trans)
{
    RT_REQUIRE(ctx && Q && t && W && dwell && trans, "null pointer");
    RT_REQUIRE(n >= 1 && nedges >= 0 && nq >= 1, "bad sizes");
    if (2 * n > RT_MAX_EXPM_STATES) {
        rt_set_error("expected history statistics need the expm kernel at order 2n = %lld; it "
                     "covers order <= %d", (long long)(2 * n), RT_MAX_EXPM_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    const size_t nn = (size_t)n * n, mm = 4 * nn;
    if (nedges == 0) {
        memset(dwell, 0, n * 8);
        memset(trans, 0, nn * 8);
        return RT_OK;
    }
    std::vector<int32_t> qi((size_t)nedges);
    for (int64_t e = 0; e < nedges; ++e) {
        const int64_t v = q_index ? q_index[e] : (nq == 1 ? 0 : e);
        RT_REQUIRE(v >= 0 && v < nq, "q_index[%lld]=%lld out of range", (long long)e, (long long)v);
        qi[(size_t)e] = (int32_t)v;
    }
    RT_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // one allocation: Q | W | t | ones | qidx | ident | blocks | expm | scale | dwell | trans
    const size_t bytes = (nq * nn + nedges * nn + 2 * nedges + nedges * mm * 2 + nedges + n + nn) * 8 +
                         2 * nedges * 4 + 64;
    unsigned char *base = nullptr;
    RT_HIP(hipMalloc((void **)&base, bytes));
    double *dQ = (double *)base, *dW = dQ + nq * nn, *dt = dW + nedges * nn, *dones = dt + nedges;
    double *dB = dones + nedges, *dE = dB + nedges * mm, *dscale = dE + nedges * mm;
    double *ddwell = dscale + nedges, *dtrans = ddwell + n;
    int32_t *dqi = (int32_t *)(dtrans + nn), *dident = dqi + nedges;
    std::vector<double> ones((size_t)nedges, 1.0);
    std::vector<int32_t> ident((size_t)nedges);
    for (int64_t e = 0; e < nedges; ++e) ident[(size_t)e] = (int32_t)e;
    int rc = RT_OK;
    hipError_t e = hipMemcpyAsync(dQ, Q, nq * nn * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dW, W, nedges * nn * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dt, t, nedges * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dones, ones.data(), nedges * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dqi, qi.data(), nedges * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dident, ident.data(), nedges * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(frechet_assemble_kernel, dim3((unsigned)nedges), dim3(256), 0, st, (int)n,
                           dQ, dqi, dt, dW, dB, dscale);
        e = hipGetLastError();
    }
    // expm of every block (order 2n, "rate matrix" = the block, "branch length" = 1)
    if (e == hipSuccess)
        rc = rt_launch_expm(ctx, 2 * n, nedges, dB, dident, dones, dE, nullptr, nullptr, 0, nullptr);
    if (e == hipSuccess && rc == RT_OK) {
        hipLaunchKernelGGL(frechet_contract_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0,
                           st, (int)n, (int)nedges, dQ, dqi, dt, dE, dscale, ddwell, dtrans);
        e = hipGetLastError();
    }
    if (e == hipSuccess && rc == RT_OK)
        e = hipMemcpyAsync(dwell, ddwell, n * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == RT_OK)
        e = hipMemcpyAsync(trans, dtrans, nn * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(base);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) {
        rt_set_error("rt_mjp_frechet_statistics: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return RT_OK;
}
