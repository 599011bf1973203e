// Spectral reconstruction of the per-edge transition matrices of ONE time-reversible rate
// matrix at many branch lengths:
//
//     P_e = A · diag(exp(lam · t_e)) · B ,     P_e[i][i] = 1 where D[i] == 0
//
// -- the reference's optional fast path examples/p53/qtop.py:76-88 (getp_spectral_v2) on
// the decomposition of qtop.py:128-152 (decompose_spectral_v2: Q = S diag(D),
// eigh(diag(sqrt D) S diag(sqrt D)) = U diag(lam) U^T, A = diag(1/sqrt D) U,
// B = U^T diag(sqrt D)).  The decomposition is once per rate matrix and stays with the caller
// (raoteh_amd/_spectral.py uses numpy's eigh, as the reference uses scipy's); this file is the
// per-branch-length part, which the reference runs as 2 numpy products per edge and site
// batch: here one workgroup per (edge, row tile), the product on the f64 matrix pipe, the result left in
// the layouts the pruning kernels read (esd_transitions, Pfrag, Pquad) exactly as the expm
// kernels leave it (expm.hip).  n <= 64.
#include "common.h"
#include "reduce.h"

namespace {

constexpr int TPB = 256;

// one workgroup per (edge, row tile of 16 states): 4 x as many workgroups as edges at 61
// states (the 126 edges of a 64-leaf tree alone leave half the CUs idle); wave c owns column
// tile c.  All global loads of a thread are issued before the first use.
template <int NT>
__global__ void __launch_bounds__(TPB)
spectral_kernel(int n, int count, const double *__restrict__ A, const double *__restrict__ lam,
                const double *__restrict__ B, const double *__restrict__ D,
                const int *__restrict__ qidx, const double *__restrict__ tt,
                double *__restrict__ P, int *__restrict__ info,
                const int *__restrict__ step_of_node, int frag_kind,
                double *__restrict__ Pfrag, double *__restrict__ Pquad, rt_reduce_args red)
{
    if (red.partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction (rt_step)
        rt_reduce_partials_body(red.partial, red.npartials, red.totals, red.nsites);
        return;
    }
    constexpr int RN = 16 * NT;
    constexpr int LD = RN + 1;
    __shared__ double As[16 * LD];             // rows 16m.. of A diag(exp(lam t)), zero-padded
    __shared__ double Bs[RN * LD];             // B, zero-padded
    __shared__ double Xs[16 * LD];             // rows 16m.. of the product
    const int b = blockIdx.x / NT;
    const int m = blockIdx.x - b * NT;         // row tile
    if (b >= count) return;
    const int tid = threadIdx.x;
    const int nn = n * n;
    const int KSn = (n + 3) / 4, NTn = (n + 15) / 16;
    if (m >= NTn) return;
    double *Pb = P + (long)b * nn;
    const int qi = qidx ? qidx[b] : 0;
    const int step = step_of_node ? step_of_node[b] : -1;
    const int KP = (KSn + 1) / 2;
    const long frag_total = (long)NTn * KP * 128;
    if (info && tid == 0 && m == 0) { info[2 * b] = 0; info[2 * b + 1] = 0; }
    // every global load of the workgroup is issued before the first dependent use (the index
    // of the rate matrix included: the root's workgroups load operands they do not need)
    const double t = tt[b];
    // (the stationary weight of row 16m + tid, for the diagonal fix after the product)
    const double dw = (D && tid < 16 && 16 * m + tid < n) ? D[16 * m + tid] : 1.0;
    // B: thread = (column, row group); A: 16 x RN entries, consecutive threads along a row
    constexpr int RP = TPB / RN;
    constexpr int PER = (RN + RP - 1) / RP;
    constexpr int APER = (16 * RN + TPB - 1) / TPB;
    const int jc = tid % RN, g = tid / RN;
    double rb[PER], ra[APER], rl[APER];
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = g + RP * k;
            rb[k] = (g < RP && i < n && jc < n) ? B[i * n + jc] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < APER; ++k) {
            const int e = tid + TPB * k;
            const int i = 16 * m + e / RN, j = e % RN;
            const bool in = e < 16 * RN && i < n && j < n;
            ra[k] = in ? A[i * n + j] : 0.0;
            rl[k] = in ? lam[j] : 0.0;
        }
    }
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
        for (int e = tid; e < 16 * n; e += TPB)
            if (16 * m + e / n < n) Pb[16 * m * n + e] = 0.0;
        if (step >= 0 && frag_kind == 0)
            for (int e = tid; e < 16 * n; e += TPB)
                if (16 * m + e / n < n) Pfrag[(long)step * nn + 16 * m * n + e] = 0.0;
        if (step >= 0 && frag_kind == 1) {
            for (int e = tid; e < KP * 128; e += TPB)
                Pfrag[(long)step * frag_total + (long)m * KP * 128 + e] = 0.0;
            if (Pquad)
                for (int e = tid; e < 4 * KSn * 16; e += TPB)
                    if (4 * m + e / (KSn * 16) < KSn)
                        Pquad[(long)step * rt_quad_stride(n) + 4 * m * KSn * 16 + e] = 0.0;
        }
        return;
    }
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = g + RP * k;
            if (g < RP && i < RN) Bs[i * LD + jc] = rb[k];
        }
#pragma unroll
        for (int k = 0; k < APER; ++k) {
            const int e = tid + TPB * k;
            if (e < 16 * RN) As[(e / RN) * LD + e % RN] = ra[k] * exp(rl[k] * t);
        }
    }
    __syncthreads();
    // D register r on lane l = X[16m + 4r + (l >> 4)][16c + (l & 15)]
    const int lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < NT; c += TPB / 64) {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4 * NT; ++kk) {
            const double a = As[(lane & 15) * LD + 4 * kk + (lane >> 4)];
            const double bb = Bs[(4 * kk + (lane >> 4)) * LD + 16 * c + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int il = 4 * r + (lane >> 4), j = 16 * c + (lane & 15);
            Xs[il * LD + j] = acc[r];
        }
    }
    __syncthreads();
    // states the stationary distribution gives no mass: the row of A is zero there; the
    // reference sets the diagonal entry to one (qtop.py:86-87)
    if (dw == 0.0) Xs[tid * LD + 16 * m + tid] = 1.0;
    __syncthreads();
    for (int e = tid; e < 16 * RN; e += TPB) {
        const int il = e / RN, j = e % RN;
        if (16 * m + il < n && j < n) {
            Pb[(16 * m + il) * n + j] = Xs[il * LD + j];
            if (step >= 0 && frag_kind == 0)
                Pfrag[(long)step * nn + (16 * m + il) * n + j] = Xs[il * LD + j];
        }
    }
    if (step >= 0 && frag_kind == 1) {
        // Pfrag[step][m][q][lane][e2] = P[16m + (lane&15)][4(2q+e2) + (lane>>4)] (prune.hip)
        for (int e = tid; e < KP * 128; e += TPB) {
            const int e2 = e & 1;
            const int ln = (e >> 1) & 63;
            const int qq = e >> 7;
            const int il = ln & 15;
            const int col = 4 * (2 * qq + e2) + (ln >> 4);
            Pfrag[(long)step * frag_total + (long)m * KP * 128 + e] =
                (col < n && 16 * m + il < n) ? Xs[il * LD + col] : 0.0;
        }
        if (Pquad) {
            // Pquad[step][rq][kk][k][i] = P[4 rq + i][4 kk + k], rq = 4m .. 4m + 3
            for (int e = tid; e < 4 * KSn * 16; e += TPB) {
                const int i = e & 3, k = (e >> 2) & 3, blk = e >> 4;
                const int rq = 4 * m + blk / KSn, col = 4 * (blk % KSn) + k;
                if (rq < KSn)
                    Pquad[(long)step * rt_quad_stride(n) + (long)rq * KSn * 16 + (blk % KSn) * 16 +
                          (e & 15)] =
                        (4 * rq + i < n && col < n) ? Xs[(4 * (blk / KSn) + i) * LD + col] : 0.0;
            }
        }
    }
}

}  // namespace

int rt_launch_spectral(rt_ctx *ctx, int64_t n, int64_t count, const double *d_A,
                       const double *d_lam, const double *d_B, const double *d_D,
                       const int32_t *d_qidx, const double *d_t, double *d_P, int32_t *d_info,
                       const int32_t *d_step_of_node, int frag_kind, double *d_Pfrag,
                       const rt_reduce_args *fused_reduce, double *d_Pquad)
{
    const rt_reduce_args red = fused_reduce ? *fused_reduce : rt_reduce_args();
    const unsigned extra = red.partial ? 1u : 0u;
    if (n < 1 || n > 64) {
        rt_set_error("spectral reconstruction: n=%lld outside 1..64", (long long)n);
        return RT_ERR_UNSUPPORTED;
    }
    if (count <= 0 && !extra) return RT_OK;
    const int nt = (int)((n + 15) / 16);
    hipEvent_t ev = nullptr;
    rt_time_begin(ctx, RT_K_EXPM, "spectral_reconstruct_mfma", &ev);
#define RT_SPECTRAL(NTV)                                                                       \
    RT_LAUNCH_TIMED(ctx, spectral_kernel<NTV>, dim3((unsigned)count * NTV + extra), dim3(TPB), \
                    0, (int)n, (int)count, d_A, d_lam, d_B, d_D, d_qidx, d_t, d_P, d_info,     \
                    d_step_of_node, frag_kind, d_Pfrag, d_Pquad, red)
    switch (nt) {
    case 1: RT_SPECTRAL(1); break;
    case 2: RT_SPECTRAL(2); break;
    case 3: RT_SPECTRAL(3); break;
    default: RT_SPECTRAL(4); break;
    }
#undef RT_SPECTRAL
    RT_HIP(hipGetLastError());
    rt_time_end(ctx, RT_K_EXPM, ev);
    return RT_OK;
}
