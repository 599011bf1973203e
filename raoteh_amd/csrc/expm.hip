// P = expm(Q * t) for a batch of small dense rate matrices, f64, one workgroup
// per matrix, every intermediate resident in LDS.
//
// Replaces scipy.linalg.expm(Q * weight) at raoteh/sampler/_mjp_dense.py:24-25
// (one call per edge per site in the reference, _mjp_dense.py:352-358) and
// pyfelscore.get_tolerance_rate_matrix (_tmjp_dense.py:239).
//
// Algorithm: N. J. Higham, "The scaling and squaring method for the matrix
// exponential revisited", SIAM J. Matrix Anal. Appl. 26(4), 2005, Algorithm 2.3:
// degree m in {3,5,7,9,13} chosen from ||A||_1 against theta_m, scaling by 2^-s
// for m = 13, [m/m] Pade approximant r = (V-U)^-1 (V+U), s squarings.  The
// linear solve is Gauss-Jordan elimination with partial (row) pivoting on the
// augmented system, rows kept in place and un-permuted at the end.
//
// LDS budget: five n x ld f64 buffers (ld = n | 1 so that column walks do not
// sit on one bank) -- 148.8 KB at n = 61 -- every matrix product is accumulated
// in registers and written back after a barrier, so products may overwrite
// their own operands and no sixth buffer is needed.
#include "common.h"

namespace {

__constant__ double c_theta[5] = {1.495585217958292e-2, 2.539398330063230e-1,
                                  9.504178996162932e-1, 2.097847961257068e0,
                                  5.371920351148152e0};
// Pade numerator coefficients b_0..b_m (Higham 2005, eq. 2.5 / table 2.3)
__constant__ double c_b3[4] = {120., 60., 12., 1.};
__constant__ double c_b5[6] = {30240., 15120., 3360., 420., 30., 1.};
__constant__ double c_b7[8] = {17297280., 8648640., 1995840., 277200.,
                               25200., 1512., 56., 1.};
__constant__ double c_b9[10] = {17643225600., 8821612800., 2075673600.,
                                302702400., 30270240., 2162160., 110880.,
                                3960., 90., 1.};
__constant__ double c_b13[14] = {64764752532480000., 32382376266240000.,
                                 7771770303897600., 1187353796428800.,
                                 129060195264000., 10559470521600.,
                                 670442572800., 33522128640., 1323241920.,
                                 40840800., 960960., 16380., 182., 1.};

constexpr int TPB = 256;

// C = A * B for n x n matrices in LDS (leading dimension ld).  Each thread owns
// a 4x4 tile of C, accumulates it in registers, and stores it after a barrier,
// so C may alias A and/or B.
__device__ __forceinline__ void lds_matmul(const double *A, const double *B,
                                           double *C, int n, int ld)
{
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    int ri[4], cj[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ri[r] = min(4 * ti + r, n - 1);
        cj[r] = min(4 * tj + r, n - 1);
    }
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int k = 0; k < n; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = A[ri[r] * ld + k];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = B[k * ld + cj[c]];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * ti + r, j = 4 * tj + c;
            if (i < n && j < n) C[i * ld + j] = acc[r][c];
        }
    __syncthreads();
}

__global__ void __launch_bounds__(TPB)
expm_lds_kernel(int n, const double *__restrict__ Q,
                const int *__restrict__ qidx, const double *__restrict__ tt,
                double *__restrict__ P, int *__restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ld = n | 1;
    const int msz = n * ld;
    double *B0 = (double *)smem;
    double *B1 = B0 + msz;
    double *B2 = B1 + msz;
    double *B3 = B2 + msz;
    double *B4 = B3 + msz;
    double *colsum = B4 + msz;                 // [64]
    int *ibuf = (int *)(colsum + 64);          // [0]=m [1]=s [2]=pivot row
    int *rowof = ibuf + 8;                     // [64]
    int *used = rowof + 64;                    // [64]

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int nn = n * n;
    double *Pb = P + (long)b * nn;
    const int qi = qidx[b];
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
        for (int e = tid; e < nn; e += TPB) Pb[e] = 0.0;
        if (info && tid == 0) { info[2 * b] = 0; info[2 * b + 1] = 0; }
        return;
    }
    const double *Qb = Q + (long)qi * nn;
    const double t = tt[b];

    // A = Q * t
    for (int e = tid; e < nn; e += TPB) {
        const int i = e / n, j = e - i * n;
        B0[i * ld + j] = Qb[e] * t;
    }
    __syncthreads();
    // ||A||_1 = max column sum
    if (tid < n) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += fabs(B0[i * ld + tid]);
        colsum[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double nrm = 0.0;
        for (int j = 0; j < n; ++j) nrm = fmax(nrm, colsum[j]);
        int m = 13, s = 0;
        if (nrm <= c_theta[0]) m = 3;
        else if (nrm <= c_theta[1]) m = 5;
        else if (nrm <= c_theta[2]) m = 7;
        else if (nrm <= c_theta[3]) m = 9;
        else if (nrm > c_theta[4]) {
            // s = ceil(log2(nrm / theta13)), exact via frexp on the ratio
            int e;
            const double f = frexp(nrm / c_theta[4], &e);   // ratio = f * 2^e
            s = (f == 0.5) ? e - 1 : e;
            if (s < 0) s = 0;
        }
        ibuf[0] = m;
        ibuf[1] = s;
    }
    __syncthreads();
    const int m = ibuf[0];
    const int s = ibuf[1];
    if (info && tid == 0) { info[2 * b] = m; info[2 * b + 1] = s; }
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n;
            B0[i * ld + j] *= sc;
        }
        __syncthreads();
    }

    double *U, *V;     // results of the Pade stage
    double *Mb, *Rb, *Xb;
    if (m == 13) {
        lds_matmul(B0, B0, B1, n, ld);         // A2
        lds_matmul(B1, B1, B2, n, ld);         // A4
        lds_matmul(B2, B1, B3, n, ld);         // A6
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n, o = i * ld + j;
            B4[o] = c_b13[13] * B3[o] + c_b13[11] * B2[o] + c_b13[9] * B1[o];
        }
        __syncthreads();
        lds_matmul(B3, B4, B4, n, ld);         // A6 * (...)
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n, o = i * ld + j;
            B4[o] += c_b13[7] * B3[o] + c_b13[5] * B2[o] + c_b13[3] * B1[o] +
                     (i == j ? c_b13[1] : 0.0);
        }
        __syncthreads();
        lds_matmul(B0, B4, B4, n, ld);         // U = A * W
        for (int e = tid; e < nn; e += TPB) {  // A is dead: reuse B0
            const int i = e / n, j = e - i * n, o = i * ld + j;
            B0[o] = c_b13[12] * B3[o] + c_b13[10] * B2[o] + c_b13[8] * B1[o];
        }
        __syncthreads();
        lds_matmul(B3, B0, B0, n, ld);
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n, o = i * ld + j;
            B0[o] += c_b13[6] * B3[o] + c_b13[4] * B2[o] + c_b13[2] * B1[o] +
                     (i == j ? c_b13[0] : 0.0);
        }
        __syncthreads();
        U = B4; V = B0; Mb = B1; Rb = B2; Xb = B3;
    } else {
        const double *bc = (m == 3) ? c_b3 : (m == 5) ? c_b5 : (m == 7) ? c_b7 : c_b9;
        lds_matmul(B0, B0, B1, n, ld);                       // A2
        if (m >= 5) lds_matmul(B1, B1, B2, n, ld);           // A4
        if (m >= 7) lds_matmul(B2, B1, B3, n, ld);           // A6
        if (m >= 9) lds_matmul(B3, B1, B4, n, ld);           // A8
        // W (odd coefficients) -> B4, V (even coefficients) -> B3, elementwise
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n, o = i * ld + j;
            const double a2 = B1[o];
            const double a4 = (m >= 5) ? B2[o] : 0.0;
            const double a6 = (m >= 7) ? B3[o] : 0.0;
            const double a8 = (m >= 9) ? B4[o] : 0.0;
            double w = bc[3] * a2 + (i == j ? bc[1] : 0.0);
            double v = bc[2] * a2 + (i == j ? bc[0] : 0.0);
            if (m >= 5) { w += bc[5] * a4; v += bc[4] * a4; }
            if (m >= 7) { w += bc[7] * a6; v += bc[6] * a6; }
            if (m >= 9) { w += bc[9] * a8; v += bc[8] * a8; }
            B4[o] = w;
            B3[o] = v;
        }
        __syncthreads();
        lds_matmul(B0, B4, B4, n, ld);                       // U = A * W
        U = B4; V = B3; Mb = B1; Rb = B2; Xb = B0;
    }

    // M = V - U, R = V + U
    for (int e = tid; e < nn; e += TPB) {
        const int i = e / n, j = e - i * n, o = i * ld + j;
        const double u = U[o], v = V[o];
        Mb[o] = v - u;
        Rb[o] = v + u;
    }
    if (tid < 64) used[tid] = 0;
    __syncthreads();

    // Gauss-Jordan with partial pivoting, rows left in place
    int singular = 0;
    for (int k = 0; k < n; ++k) {
        if (tid < 64) {
            double v = -1.0;
            int r = tid;
            if (tid < n && !used[tid]) v = fabs(Mb[tid * ld + k]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(v, o, 64);
                const int orow = __shfl_xor(r, o, 64);
                if (ov > v || (ov == v && orow < r)) { v = ov; r = orow; }
            }
            if (tid == 0) {
                ibuf[2] = r;
                ibuf[3] = (v > 0.0) ? 0 : 1;
                used[r] = 1;
                rowof[k] = r;
            }
        }
        __syncthreads();
        const int pr = ibuf[2];
        if (ibuf[3]) { singular = 1; break; }
        const double rinv = 1.0 / Mb[pr * ld + k];
        // rows i != pr: M[i][j>k] -= f * M[pr][j], R[i][:] -= f * R[pr][:]
        const int tx = tid & 63, ty = tid >> 6;
        for (int i = ty; i < n; i += 4) {
            if (i == pr) continue;
            const double f = Mb[i * ld + k] * rinv;
            for (int cc = k + 1 + tx; cc < 2 * n; cc += 64) {
                if (cc < n) Mb[i * ld + cc] = fma(-f, Mb[pr * ld + cc], Mb[i * ld + cc]);
                else Rb[i * ld + cc - n] = fma(-f, Rb[pr * ld + cc - n], Rb[i * ld + cc - n]);
            }
        }
        __syncthreads();
    }
    if (singular) {
        for (int e = tid; e < nn; e += TPB) Pb[e] = __builtin_nan("");
        if (info && tid == 0) info[2 * b] = -1;
        return;
    }
    // X[k][:] = R[rowof[k]][:] / M[rowof[k]][k]
    for (int e = tid; e < nn; e += TPB) {
        const int k = e / n, j = e - k * n;
        const int r = rowof[k];
        Xb[k * ld + j] = Rb[r * ld + j] / Mb[r * ld + k];
    }
    __syncthreads();
    for (int q = 0; q < s; ++q) lds_matmul(Xb, Xb, Xb, n, ld);
    for (int e = tid; e < nn; e += TPB) {
        const int i = e / n, j = e - i * n;
        Pb[e] = Xb[i * ld + j];
    }
}

}  // namespace

int rt_launch_expm(rt_ctx *ctx, int64_t n, int64_t count, const double *d_Q,
                   const int32_t *d_qidx, const double *d_t, double *d_P,
                   int32_t *d_info)
{
    if (n < 1 || n > RT_MAX_EXPM_STATES) {
        rt_set_error("expm: n=%lld outside 1..%d", (long long)n, RT_MAX_EXPM_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    if (count <= 0) return RT_OK;
    const int ld = (int)n | 1;
    const size_t lds = (size_t)5 * n * ld * 8 + 64 * 8 + (8 + 64 + 64) * 4;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        RT_HIP(hipFuncSetAttribute((const void *)expm_lds_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_lds = lds;
    }
    hipEvent_t ev = nullptr;
    rt_time_begin(ctx, RT_K_EXPM, "expm_lds", &ev);
    hipLaunchKernelGGL(expm_lds_kernel, dim3((unsigned)count), dim3(TPB), lds,
                       ctx->stream, (int)n, d_Q, d_qidx, d_t, d_P, d_info);
    RT_HIP(hipGetLastError());
    rt_time_end(ctx, RT_K_EXPM, ev);
    return RT_OK;
}
