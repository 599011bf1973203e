// P = expm(Q * t) for a batch of small dense rate matrices, f64, one workgroup
// (4 waves) per matrix, every intermediate resident in LDS or registers.
//
// Replaces scipy.linalg.expm(Q * weight) at raoteh/sampler/_mjp_dense.py:24-25
// (one call per edge per site in the reference, _mjp_dense.py:352-358) and
// pyfelscore.get_tolerance_rate_matrix (_tmjp_dense.py:239).
//
// Algorithm: N. J. Higham, "The scaling and squaring method for the matrix
// exponential revisited", SIAM J. Matrix Anal. Appl. 26(4), 2005, Algorithm 2.3:
// degree m in {3,5,7,9,13} chosen from ||A||_1 against theta_m, scaling by 2^-s
// for m = 13, [m/m] Pade approximant r = (V-U)^-1 (V+U), s squarings.
//
// * Matrix products: v_mfma_f64_16x16x4_f64 on n padded to 16*NT; the NT*NT
//   output tiles are dealt over the four waves; every product is accumulated
//   in registers and stored after a barrier, so a product may overwrite its own
//   operands and five n x (n|1) LDS buffers suffice (148.8 KB at n = 61).
// * Linear solve: Gauss-Jordan with partial pivoting on the augmented system
//   [V-U | V+U], held ENTIRELY IN REGISTERS in a 2-D cyclic distribution
//   (thread (ri, ci) owns rows ri+16a, columns ci+16b: 4 x 8 elements).  Per
//   pivot step only the pivot column and the pivot row travel through LDS (two
//   barriers), every wave finds the pivot itself (wave-wide argmax), rows are
//   never swapped: the permutation is undone when X is written back.
// * Epilogue: P is written in the reference's esd order and, fused, in the
//   step-ordered layout the pruning kernel of this model reads (lane family:
//   [step][n][n]; MFMA family: A-fragment order), which removes a launch.
#include "common.h"
#include "reduce.h"

#include <cstdlib>

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
typedef double double4_t __attribute__((ext_vector_type(4)));

// Wave-wide unsigned max in registers (DPP scan: row_shr 1/2/4/8, row_bcast 15/31;
// the total lands in lane 63).  Checked against a shuffle reduction by
// tools/micro/dpp_umax_test.hip.  Six dependent ds_bpermute round trips (what
// __shfl_xor compiles to) would cost ~10x more per pivot step.
__device__ __forceinline__ unsigned wave_umax_dpp(unsigned v)
{
#define RT_DPP(ctrl, rmask) (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
    unsigned t;
    t = RT_DPP(0x111, 0xf); v = v > t ? v : t;   // row_shr:1
    t = RT_DPP(0x112, 0xf); v = v > t ? v : t;   // row_shr:2
    t = RT_DPP(0x114, 0xf); v = v > t ? v : t;   // row_shr:4
    t = RT_DPP(0x118, 0xf); v = v > t ? v : t;   // row_shr:8
    t = RT_DPP(0x142, 0xa); v = v > t ? v : t;   // row_bcast:15
    t = RT_DPP(0x143, 0xc); v = v > t ? v : t;   // row_bcast:31
#undef RT_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// C = A * B (n x n, leading dimension ld, in LDS) on the f64 matrix pipe.
// A lane l: A[16m + (l&15)][4kk + (l>>4)], B lane l: B[4kk + (l>>4)][16j + (l&15)],
// D lane l reg r: C[16m + 4r + (l>>4)][16j + (l&15)].  Elements outside n x n
// read as zero.  C may alias A and/or B.
__device__ __forceinline__ void lds_matmul(const double *A, const double *B, double *C,
                                           int n, int ld, int NT, int KS)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int nitems = NT * NT;
    double4_t acc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        acc[it] = (double4_t){0.0, 0.0, 0.0, 0.0};
        const int item = wave + 4 * it;
        if (item < nitems) {
            const int m = item / NT, j = item - m * NT;
            const int arow = 16 * m + lr, bcol = 16 * j + lr;
            const bool aok = arow < n, bok = bcol < n;
            const double *ap = A + (aok ? arow : 0) * ld;
            const double *bp = B + (bok ? bcol : 0);
            // groups of four k-steps: the eight LDS reads of a group are in flight
            // together (one dependent read -> MFMA pair per iteration is bound by the
            // LDS latency: C3 expm 129 -> 111 us).  Advancing the wave's four output
            // tiles together instead (four independent chains) measured slower (122 us).
            int kk = 0;
            for (; kk + 4 <= KS; kk += 4) {
                double a[4], bq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 4 * (kk + u) + lq;
                    const bool kok = k < n;
                    a[u] = (aok && kok) ? ap[kok ? k : 0] : 0.0;
                    bq[u] = (bok && kok) ? bp[(kok ? k : 0) * ld] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bq[u], acc[it], 0, 0, 0);
            }
            for (; kk < KS; ++kk) {
                const int k = 4 * kk + lq;
                const bool kok = k < n;
                const double a = (aok && kok) ? ap[kok ? k : 0] : 0.0;
                const double b = (bok && kok) ? bp[(kok ? k : 0) * ld] : 0.0;
                acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[it], 0, 0, 0);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = wave + 4 * it;
        if (item < nitems) {
            const int m = item / NT, j = item - m * NT;
            const int col = 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * m + 4 * r + lq;
                if (row < n && col < n) C[row * ld + col] = acc[it][r];
            }
        }
    }
    __syncthreads();
}

// all n x n elements, four rows per pass, no integer division: thread (tid>>6, tid&63)
#define RT_FOR_EACH_ELEMENT(I, J, O)                                              \
    for (int I = tid >> 6, J = tid & 63, O = (tid >> 6) * ld + (tid & 63); I < n; \
         I += 4, O += 4 * ld)                                                      \
        if (J < n)

__global__ void __launch_bounds__(TPB)
expm_kernel(int n, const double *__restrict__ Q, const int *__restrict__ qidx,
            const double *__restrict__ tt, double *__restrict__ P,
            int *__restrict__ info,
            // fused repack (all optional): step of each node, layout, output
            const int *__restrict__ step_of_node, int frag_kind,
            double *__restrict__ Pfrag, rt_reduce_args red)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (red.partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction
        rt_reduce_partials_body(red.partial, red.npartials, red.totals, red.nsites);
        return;
    }
    const int ld = n | 1;
    const int msz = n * ld;
    const int NT = (n + 15) / 16;
    const int KS = (n + 3) / 4;
    double *B0 = (double *)smem;
    double *B1 = B0 + msz;
    double *B2 = B1 + msz;
    double *B3 = B2 + msz;
    double *B4 = B3 + msz;
    double *colbuf = B4 + msz;                 // [2][64]
    double *rowbuf = colbuf + 128;             // [128]
    double *dinv = rowbuf + 128;               // [64] 1 / pivot of the row
    int *ibuf = (int *)(dinv + 64);            // [0]=m [1]=s [2]=singular
    int *kof = ibuf + 8;                       // [64] pivot column of each row

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int nn = n * n;
    double *Pb = P + (long)b * nn;
    const int qi = qidx[b];
    const int step = step_of_node ? step_of_node[b] : -1;
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
        for (int e = tid; e < nn; e += TPB) Pb[e] = 0.0;
        if (info && tid == 0) { info[2 * b] = 0; info[2 * b + 1] = 0; }
        if (step >= 0 && frag_kind == 0)
            for (int e = tid; e < nn; e += TPB) Pfrag[(long)step * nn + e] = 0.0;
        if (step >= 0 && frag_kind == 1) {
            const int total = NT * ((KS + 1) / 2) * 128;
            for (int e = tid; e < total; e += TPB) Pfrag[(long)step * total + e] = 0.0;
        }
        return;
    }
    const double *Qb = Q + (long)qi * nn;
    const double t = tt[b];

    // A = Q * t
    RT_FOR_EACH_ELEMENT(i, j, o_unused_) {
        const int e = i * n + j;
        B0[i * ld + j] = Qb[e] * t;
    }
    __syncthreads();
    // ||A||_1 = max column sum; every wave computes it (no second barrier)
    double nrm;
    {
        double s = 0.0;
        if (lane < n)
            for (int i = 0; i < n; ++i) s += fabs(B0[i * ld + lane]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
        nrm = s;
    }
    int m = 13, s = 0;
    if (nrm <= c_theta[0]) m = 3;
    else if (nrm <= c_theta[1]) m = 5;
    else if (nrm <= c_theta[2]) m = 7;
    else if (nrm <= c_theta[3]) m = 9;
    else if (nrm > c_theta[4]) {
        // s = ceil(log2(nrm / theta13)), exact via frexp on the ratio
        int e;
        const double f = frexp(nrm / c_theta[4], &e);       // ratio = f * 2^e
        s = (f == 0.5) ? e - 1 : e;
        if (s < 0) s = 0;
    }
    m = __builtin_amdgcn_readfirstlane(m);
    s = __builtin_amdgcn_readfirstlane(s);
    if (info && tid == 0) { info[2 * b] = m; info[2 * b + 1] = s; }
    if (s > 0) {
        __syncthreads();
        const double sc = ldexp(1.0, -s);
        RT_FOR_EACH_ELEMENT(i, j, o) {
            B0[o] *= sc;
        }
        __syncthreads();
    }

    double *U, *V, *Xb;
    if (m == 13) {
        lds_matmul(B0, B0, B1, n, ld, NT, KS);         // A2
        lds_matmul(B1, B1, B2, n, ld, NT, KS);         // A4
        lds_matmul(B2, B1, B3, n, ld, NT, KS);         // A6
        RT_FOR_EACH_ELEMENT(i, j, o) {
            B4[o] = c_b13[13] * B3[o] + c_b13[11] * B2[o] + c_b13[9] * B1[o];
        }
        __syncthreads();
        lds_matmul(B3, B4, B4, n, ld, NT, KS);         // A6 * (...)
        RT_FOR_EACH_ELEMENT(i, j, o) {
            B4[o] += c_b13[7] * B3[o] + c_b13[5] * B2[o] + c_b13[3] * B1[o] +
                     (i == j ? c_b13[1] : 0.0);
        }
        __syncthreads();
        lds_matmul(B0, B4, B4, n, ld, NT, KS);         // U = A * W
        RT_FOR_EACH_ELEMENT(i, j, o) {                 // A is dead: reuse B0
            B0[o] = c_b13[12] * B3[o] + c_b13[10] * B2[o] + c_b13[8] * B1[o];
        }
        __syncthreads();
        lds_matmul(B3, B0, B0, n, ld, NT, KS);
        RT_FOR_EACH_ELEMENT(i, j, o) {
            B0[o] += c_b13[6] * B3[o] + c_b13[4] * B2[o] + c_b13[2] * B1[o] +
                     (i == j ? c_b13[0] : 0.0);
        }
        __syncthreads();
        U = B4; V = B0; Xb = B1;
    } else {
        const double *bc = (m == 3) ? c_b3 : (m == 5) ? c_b5 : (m == 7) ? c_b7 : c_b9;
        lds_matmul(B0, B0, B1, n, ld, NT, KS);                       // A2
        if (m >= 5) lds_matmul(B1, B1, B2, n, ld, NT, KS);           // A4
        if (m >= 7) lds_matmul(B2, B1, B3, n, ld, NT, KS);           // A6
        if (m >= 9) lds_matmul(B3, B1, B4, n, ld, NT, KS);           // A8
        // W (odd coefficients) -> B4, V (even coefficients) -> B3, elementwise
        RT_FOR_EACH_ELEMENT(i, j, o) {
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
        lds_matmul(B0, B4, B4, n, ld, NT, KS);                       // U = A * W
        U = B4; V = B3; Xb = B0;
    }

    // ---- Gauss-Jordan on [M | R] = [V - U | V + U] in registers -------------
    // thread (ri, ci): rows ri + 16a (a < 4), augmented columns ci + 16b (b < 8)
    const int ri = tid & 15, ci = tid >> 4;
    double g[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = ri + 16 * a;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const int c = ci + 16 * bb;
            double v = 0.0;
            if (i < n && c < 2 * n) {
                const int j = c < n ? c : c - n;
                const double u = U[i * ld + j], w = V[i * ld + j];
                v = c < n ? w - u : w + u;
            }
            g[a][bb] = v;
        }
    }
    // column 0 -> colbuf[0]
    if (ci == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) colbuf[ri + 16 * a] = g[a][0];
    }
    if (tid == 0) ibuf[2] = 0;
    __syncthreads();
    bool used = lane >= n;              // replicated in every wave: row = lane
    for (int k = 0; k < n; ++k) {
        const double *cb = colbuf + (k & 1) * 64;
        // Every wave finds the pivot row itself.  Key = the top 26 bits of |value|
        // (exponent + 15 mantissa bits: partial pivoting does not need more) with
        // 63 - lane in the low 6 bits, so one unsigned max yields the row too
        // (ties -> lowest row).
        const double cv = cb[lane];
        // this thread's rows of column k (for the multipliers below): requested now,
        // used after the barrier
        double ck[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) ck[a] = cb[ri + 16 * a];
        const unsigned hi = (unsigned)(__double_as_longlong(fabs(cv)) >> 32);
        unsigned key = used ? 0u : ((hi & ~63u) | (63u - (unsigned)lane));
        // every lane inverts its own candidate while the scan runs; the winner's
        // value and reciprocal then come out of registers (v_readlane), not LDS
        const double rc = 1.0 / cv;
        key = wave_umax_dpp(key);
        const int pr = 63 - (int)(key & 63u);
        const double pvt = __longlong_as_double(
            ((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(cv) >> 32), pr) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(cv), pr));
        if ((key & ~63u) == 0u || !(fabs(pvt) > 0.0) || !(fabs(pvt) < 1e308 * 10.0)) {
            // zero / denormal column, or inf / NaN: wave-uniform
            if (tid == 0) ibuf[2] = 1;
            break;
        }
        if (lane == pr) used = true;
        const double rinv = __longlong_as_double(
            ((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(rc) >> 32), pr) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(rc), pr));
        if (tid == 0) { dinv[pr] = rinv; kof[pr] = k; }
        // pivot-row owners publish their 8 columns
        if (ri == (pr & 15)) {
            double *rb = rowbuf + ci;
            switch (pr >> 4) {                  // wave-uniform
            case 0:
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) rb[16 * bb] = g[0][bb];
                break;
            case 1:
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) rb[16 * bb] = g[1][bb];
                break;
            case 2:
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) rb[16 * bb] = g[2][bb];
                break;
            default:
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) rb[16 * bb] = g[3][bb];
                break;
            }
        }
        __syncthreads();
        double f[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = ri + 16 * a;
            f[a] = (i == pr || i >= n) ? 0.0 : ck[a] * rinv;
        }
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const int c = ci + 16 * bb;
            // columns <= k of M are finished; skipping them keeps the pivots exact
            const double prow = (c > k && c < 2 * n) ? rowbuf[c] : 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) g[a][bb] = fma(-f[a], prow, g[a][bb]);
        }
        // owners of column k+1 publish it for the next step (other colbuf half)
        if (k + 1 < n && ci == ((k + 1) & 15)) {
            double *nb = colbuf + ((k + 1) & 1) * 64 + ri;
            switch ((k + 1) >> 4) {             // wave-uniform
            case 0:
#pragma unroll
                for (int a = 0; a < 4; ++a) nb[16 * a] = g[a][0];
                break;
            case 1:
#pragma unroll
                for (int a = 0; a < 4; ++a) nb[16 * a] = g[a][1];
                break;
            case 2:
#pragma unroll
                for (int a = 0; a < 4; ++a) nb[16 * a] = g[a][2];
                break;
            default:
#pragma unroll
                for (int a = 0; a < 4; ++a) nb[16 * a] = g[a][3];
                break;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (ibuf[2]) {
        for (int e = tid; e < nn; e += TPB) Pb[e] = __builtin_nan("");
        if (info && tid == 0) info[2 * b] = -1;
        return;
    }
    // X[kof[i]][j] = R[i][j] / pivot(i)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = ri + 16 * a;
        if (i < n) {
            const double d = dinv[i];
            const int xr = kof[i];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                const int c = ci + 16 * bb;
                if (c >= n && c < 2 * n) Xb[xr * ld + (c - n)] = g[a][bb] * d;
            }
        }
    }
    __syncthreads();
    for (int q = 0; q < s; ++q) lds_matmul(Xb, Xb, Xb, n, ld, NT, KS);

    RT_FOR_EACH_ELEMENT(i, j, o_unused_) {
        const int e = i * n + j;
        Pb[e] = Xb[i * ld + j];
    }
    if (step >= 0 && frag_kind == 0) {
        RT_FOR_EACH_ELEMENT(i, j, o_unused_) {
            const int e = i * n + j;
            Pfrag[(long)step * nn + e] = Xb[i * ld + j];
        }
    } else if (step >= 0 && frag_kind == 1) {
        // Pfrag[step][m][q][lane][e2] = P[16m + (lane&15)][4(2q+e2) + (lane>>4)]
        const int KP = (KS + 1) / 2;
        const int total = NT * KP * 128;
        for (int e = tid; e < total; e += TPB) {
            const int e2 = e & 1;
            const int ln = (e >> 1) & 63;
            const int q = (e >> 7) % KP;
            const int mm = (e >> 7) / KP;
            const int row = 16 * mm + (ln & 15);
            const int col = 4 * (2 * q + e2) + (ln >> 4);
            Pfrag[(long)step * total + e] = (row < n && col < n) ? Xb[row * ld + col] : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------
// 4 < n <= 62, default: scaling and squaring around a TAYLOR polynomial evaluated
// by Paterson-Stockmeyer -- matrix products only, no linear solve.
//
// The [m/m] Pade kernel above spends half of its time in the n pivot steps of the
// solve (two workgroup barriers each, a chain of n dependent rank-1 updates that the
// matrix pipe cannot help with: 50 of 103 us at n = 61).  A Taylor polynomial of
// degree m = k q costs (k - 1) + (q - 1) products and nothing else:
//     powers   A^2 = A A, ..., A^k = A A^(k-1)                      (k - 1 products)
//     Horner   T = c_m A^k + B_(q-1);  T = A^k T + B_j, j = q-2..0  (q - 1 products)
//     with B_j = sum_{i<k} c_(kj+i) A^i  (elementwise),  c_i = 1 / i!
// Degree from ||A||_1 against theta_m, the largest norm for which the backward error
// of T_m stays below 2^-53 (computed as in Higham 2005 sec. 2 for the Taylor series of
// log(e^-x T_m(x)); the same numbers as Al-Mohy & Higham 2011, table 3.1):
//     m = 4 (k = 2: 2 products), 8, 12, 16 (k = 4: 4, 5, 6 products);
// beyond theta_16 = 0.78 one squaring per doubling of the norm (a squaring doubles the
// range for one product, a higher degree does not).  Entrywise accuracy against
// 60-digit arithmetic on the codon matrix: 6e-15 relative at t = 0.1 (scipy's Pade
// approximant: 5e-14).
//
// Products run on the f64 matrix pipe as in lds_matmul, but with the operands of
// k-step kk + 1 requested before the MFMAs of k-step kk are issued and the up to four
// output tiles of a wave advanced together (four independent accumulation chains), and
// the "+ B_j" of a Horner step is applied when the product is stored.
// ---------------------------------------------------------------------------

__constant__ double c_theta_taylor[4] = {3.3971688e-4, 4.9912289e-2, 2.9961589e-1,
                                         7.8028743e-1};
// 1 / i!, i = 0..16
__constant__ double c_inv_fact[17] = {
    1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
    1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0,
    1.0 / 6227020800.0, 1.0 / 87178291200.0, 1.0 / 1307674368000.0,
    1.0 / 20922789888000.0};

// C = X * Y (+ B) on the matrix pipe; n x n matrices in LDS, leading dimension ld.
// B = cf[0] I + cf[1] P1 + cf[2] P2 + cf[3] P3 when cf != nullptr (P2 / P3 may be null).
// C may alias X and / or Y (results are held in registers across a barrier).
// NT (row / column tiles, n <= 16 NT) is a template parameter so that the number of
// output tiles per wave NI is a constant and the loop body is straight-line code: every
// LDS read is unconditional (clamped address, value selected afterwards) and every wave
// issues NI MFMAs per k-step -- a branch per load or per MFMA makes hipcc serialise the
// whole product (one exec-mask region per read, the accumulators copied in and out of the
// AGPRs around every MFMA: 78 us per 61-state expm instead of 20).
template <int NT>
__device__ __forceinline__ void lds_matmul_pipelined(const double *X, const double *Y, double *C,
                                                     int n, int ld, int KS,
                                                     const double *cf, const double *P1,
                                                     const double *P2, const double *P3)
{
    constexpr int NI = NT == 4 ? 4 : NT == 3 ? 3 : 1;      // items (output tiles) per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    constexpr int nitems = NT * NT;
    double4_t acc[NI];
    const double *ap[NI];
    const double *bp[NI];
    bool aok[NI], bok[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        acc[it] = (double4_t){0.0, 0.0, 0.0, 0.0};
        const int item = wave + 4 * it;
        const bool valid = item < nitems;
        const int m = valid ? item / NT : 0, j = valid ? item - m * NT : 0;
        const int arow = 16 * m + lr, bcol = 16 * j + lr;
        aok[it] = valid && arow < n;
        bok[it] = valid && bcol < n;
        ap[it] = X + (aok[it] ? arow : 0) * ld;
        bp[it] = Y + (bok[it] ? bcol : 0);
    }
    double a0[NI], b0[NI];
    {
        const bool kok = lq < n;
        const int k = kok ? lq : 0;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const double av = ap[it][k], bv = bp[it][k * ld];      // always in bounds
            a0[it] = (aok[it] && kok) ? av : 0.0;
            b0[it] = (bok[it] && kok) ? bv : 0.0;
        }
    }
    for (int kk = 0; kk < KS; ++kk) {
        double a1[NI], b1[NI];
        {
            // operands of the next k-step (the last iteration re-reads its own)
            const int kn = 4 * (kk + 1 < KS ? kk + 1 : kk) + lq;
            const bool kok = kn < n;
            const int k = kok ? kn : 0;
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const double av = ap[it][k], bv = bp[it][k * ld];
                a1[it] = (aok[it] && kok) ? av : 0.0;
                b1[it] = (bok[it] && kok) ? bv : 0.0;
            }
        }
#pragma unroll
        for (int it = 0; it < NI; ++it)
            acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[it], b0[it], acc[it], 0, 0, 0);
#pragma unroll
        for (int it = 0; it < NI; ++it) { a0[it] = a1[it]; b0[it] = b1[it]; }
    }
    // the addend of a Horner step, read before the barrier
    double4_t add[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        add[it] = (double4_t){0.0, 0.0, 0.0, 0.0};
        const int item = wave + 4 * it;
        if (cf && item < nitems) {
            const int m = item / NT, j = item - m * NT;
            const int col = 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * m + 4 * r + lq;
                if (row < n && col < n) {
                    const int o = row * ld + col;
                    double v = cf[1] * P1[o] + (row == col ? cf[0] : 0.0);
                    if (P2) v += cf[2] * P2[o];
                    if (P3) v += cf[3] * P3[o];
                    add[it][r] = v;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int item = wave + 4 * it;
        if (item < nitems) {
            const int m = item / NT, j = item - m * NT;
            const int col = 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * m + 4 * r + lq;
                if (row < n && col < n) C[row * ld + col] = acc[it][r] + add[it][r];
            }
        }
    }
    __syncthreads();
}

template <int NT>
__global__ void __launch_bounds__(TPB)
expm_taylor_kernel(int n, const double *__restrict__ Q, const int *__restrict__ qidx,
                   const double *__restrict__ tt, double *__restrict__ P,
                   int *__restrict__ info, const int *__restrict__ step_of_node,
                   int frag_kind, double *__restrict__ Pfrag, rt_reduce_args red)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (red.partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction
        rt_reduce_partials_body(red.partial, red.npartials, red.totals, red.nsites);
        return;
    }
    const int ld = n | 1;
    const int msz = n * ld;
    const int KS = (n + 3) / 4;
    double *B0 = (double *)smem;               // A
    double *B1 = B0 + msz;                     // A^2
    double *B2 = B1 + msz;                     // A^3
    double *B3 = B2 + msz;                     // A^4
    double *B4 = B3 + msz;                     // T
    double *cfs = B4 + msz;                    // [4] coefficients of the current B_j

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int nn = n * n;
    double *Pb = P + (long)b * nn;
    const int qi = qidx[b];
    const int step = step_of_node ? step_of_node[b] : -1;
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
        for (int e = tid; e < nn; e += TPB) Pb[e] = 0.0;
        if (info && tid == 0) { info[2 * b] = 0; info[2 * b + 1] = 0; }
        if (step >= 0 && frag_kind == 0)
            for (int e = tid; e < nn; e += TPB) Pfrag[(long)step * nn + e] = 0.0;
        if (step >= 0 && frag_kind == 1) {
            const int total = NT * ((KS + 1) / 2) * 128;
            for (int e = tid; e < total; e += TPB) Pfrag[(long)step * total + e] = 0.0;
        }
        return;
    }
    const double *Qb = Q + (long)qi * nn;
    const double t = tt[b];
    RT_FOR_EACH_ELEMENT(i, j, o_unused_) {
        B0[i * ld + j] = Qb[i * n + j] * t;
    }
    __syncthreads();
    double nrm;
    {
        double s = 0.0;
        if (lane < n)
            for (int i = 0; i < n; ++i) s += fabs(B0[i * ld + lane]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
        nrm = s;
    }
    if (!(nrm < 1e300)) {                      // inf / NaN in Q * t (wave- and block-uniform)
        for (int e = tid; e < nn; e += TPB) Pb[e] = __builtin_nan("");
        if (info && tid == 0) { info[2 * b] = -1; info[2 * b + 1] = 0; }
        return;
    }
    int m = 16, s = 0;
    if (nrm <= c_theta_taylor[0]) m = 4;
    else if (nrm <= c_theta_taylor[1]) m = 8;
    else if (nrm <= c_theta_taylor[2]) m = 12;
    else if (nrm > c_theta_taylor[3]) {
        int e;
        const double f = frexp(nrm / c_theta_taylor[3], &e);    // ratio = f * 2^e
        s = (f == 0.5) ? e - 1 : e;
        if (s < 0) s = 0;
    }
    m = __builtin_amdgcn_readfirstlane(m);
    s = __builtin_amdgcn_readfirstlane(s);
    if (info && tid == 0) { info[2 * b] = m; info[2 * b + 1] = s; }
    if (s > 0) {
        __syncthreads();
        const double sc = ldexp(1.0, -s);
        RT_FOR_EACH_ELEMENT(i, j, o) {
            B0[o] *= sc;
        }
        __syncthreads();
    }
    const int k = (m == 4) ? 2 : 4;
    const int q = m / k;
    lds_matmul_pipelined<NT>(B0, B0, B1, n, ld, KS, nullptr, nullptr, nullptr, nullptr);   // A^2
    if (k == 4) {
        lds_matmul_pipelined<NT>(B0, B1, B2, n, ld, KS, nullptr, nullptr, nullptr, nullptr);  // A^3
        lds_matmul_pipelined<NT>(B0, B2, B3, n, ld, KS, nullptr, nullptr, nullptr, nullptr);  // A^4
    }
    const double *Pk = (k == 4) ? B3 : B1;
    const double *P2 = (k == 4) ? B1 : nullptr;
    const double *P3 = (k == 4) ? B2 : nullptr;
    // T = c_m A^k + B_(q-1)
    {
        const int base = k * (q - 1);
        RT_FOR_EACH_ELEMENT(i, j, o) {
            double v = c_inv_fact[m] * Pk[o] + c_inv_fact[base + 1] * B0[o] +
                       (i == j ? c_inv_fact[base] : 0.0);
            if (k == 4) v += c_inv_fact[base + 2] * B1[o] + c_inv_fact[base + 3] * B2[o];
            B4[o] = v;
        }
        __syncthreads();
    }
    for (int jj = q - 2; jj >= 0; --jj) {
        if (tid < 4) cfs[tid] = (tid < k) ? c_inv_fact[k * jj + tid] : 0.0;
        __syncthreads();
        lds_matmul_pipelined<NT>(Pk, B4, B4, n, ld, KS, cfs, B0, P2, P3);   // T = A^k T + B_j
    }
    for (int r = 0; r < s; ++r)
        lds_matmul_pipelined<NT>(B4, B4, B4, n, ld, KS, nullptr, nullptr, nullptr, nullptr);

    const double *Xb = B4;
    RT_FOR_EACH_ELEMENT(i, j, o_unused_) {
        Pb[i * n + j] = Xb[i * ld + j];
    }
    if (step >= 0 && frag_kind == 0) {
        RT_FOR_EACH_ELEMENT(i, j, o_unused_) {
            Pfrag[(long)step * nn + i * n + j] = Xb[i * ld + j];
        }
    } else if (step >= 0 && frag_kind == 1) {
        // Pfrag[step][m][q][lane][e2] = P[16m + (lane&15)][4(2q+e2) + (lane>>4)]
        const int KP = (KS + 1) / 2;
        const int total = NT * KP * 128;
        for (int e = tid; e < total; e += TPB) {
            const int e2 = e & 1;
            const int ln = (e >> 1) & 63;
            const int qq = (e >> 7) % KP;
            const int mm = (e >> 7) / KP;
            const int row = 16 * mm + (ln & 15);
            const int col = 4 * (2 * qq + e2) + (ln >> 4);
            Pfrag[(long)step * total + e] = (row < n && col < n) ? Xb[row * ld + col] : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------
// n <= 4: one LANE per matrix, everything in registers (compile-time N, fully
// unrolled, statically indexed).  A 64-leaf 4-state tree has 126 matrices: the
// workgroup-per-matrix kernel above spends its time in barriers there, and the
// tolerance processes of the reference (n = 3, a different Q on every edge,
// _tmjp.py:863-893) want exactly this shape.  Same algorithm (Higham 2005);
// the solve is Gaussian elimination with partial pivoting by predicated row
// swaps + back substitution.
// ---------------------------------------------------------------------------

template <int N>
struct SmallMat {
    double a[N][N];
};

template <int N>
__device__ __forceinline__ void sm_mul(const SmallMat<N> &A, const SmallMat<N> &B,
                                       SmallMat<N> &C)
{
    SmallMat<N> T;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double acc = A.a[i][0] * B.a[0][j];
#pragma unroll
            for (int k = 1; k < N; ++k) acc = fma(A.a[i][k], B.a[k][j], acc);
            T.a[i][j] = acc;
        }
    C = T;
}

// C = alpha*A + beta*B + gamma*Cin + delta*I   (any of the matrices may be unused)
template <int N>
__device__ __forceinline__ void sm_comb(SmallMat<N> &out, double ca, const SmallMat<N> &A,
                                        double cb, const SmallMat<N> &B, double cc,
                                        const SmallMat<N> &C, double ci)
{
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
            out.a[i][j] = ca * A.a[i][j] + cb * B.a[i][j] + cc * C.a[i][j] +
                          (i == j ? ci : 0.0);
}

template <int N>
__global__ void __launch_bounds__(256)
expm_small_kernel(int count, const double *__restrict__ Q, const int *__restrict__ qidx,
                  const double *__restrict__ tt, double *__restrict__ P,
                  int *__restrict__ info, const int *__restrict__ step_of_node,
                  double *__restrict__ Pfrag, rt_reduce_args red)
{
    if (red.partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction
        rt_reduce_partials_body(red.partial, red.npartials, red.totals, red.nsites);
        return;
    }
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= count) return;
    constexpr int NN = N * N;
    double *Pb = P + (long)b * NN;
    const int qi = qidx[b];
    const int step = step_of_node ? step_of_node[b] : -1;
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
#pragma unroll
        for (int e = 0; e < NN; ++e) Pb[e] = 0.0;
        if (info) { info[2 * b] = 0; info[2 * b + 1] = 0; }
        if (step >= 0)
#pragma unroll
            for (int e = 0; e < NN; ++e) Pfrag[(long)step * NN + e] = 0.0;
        return;
    }
    const double *Qb = Q + (long)qi * NN;
    const double t = tt[b];
    SmallMat<N> A;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) A.a[i][j] = Qb[i * N + j] * t;
    double nrm = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) cs += fabs(A.a[i][j]);
        nrm = fmax(nrm, cs);
    }
    int m = 13, s = 0;
    if (nrm <= c_theta[0]) m = 3;
    else if (nrm <= c_theta[1]) m = 5;
    else if (nrm <= c_theta[2]) m = 7;
    else if (nrm <= c_theta[3]) m = 9;
    else if (nrm > c_theta[4]) {
        int e;
        const double f = frexp(nrm / c_theta[4], &e);
        s = (f == 0.5) ? e - 1 : e;
        if (s < 0) s = 0;
    }
    if (info) { info[2 * b] = m; info[2 * b + 1] = s; }
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) A.a[i][j] *= sc;
    }
    SmallMat<N> A2, A4, A6, U, V, W, Z;
    sm_comb(Z, 0.0, A, 0.0, A, 0.0, A, 0.0);           // Z = 0
    sm_mul(A, A, A2);
    A4 = Z; A6 = Z;
    if (m >= 5) sm_mul(A2, A2, A4);
    if (m >= 7) sm_mul(A4, A2, A6);
    if (m == 13) {
        sm_comb(W, c_b13[13], A6, c_b13[11], A4, c_b13[9], A2, 0.0);
        sm_mul(A6, W, W);
        sm_comb(U, c_b13[7], A6, c_b13[5], A4, c_b13[3], A2, c_b13[1]);
        sm_comb(W, 1.0, W, 1.0, U, 0.0, Z, 0.0);
        sm_mul(A, W, U);
        sm_comb(W, c_b13[12], A6, c_b13[10], A4, c_b13[8], A2, 0.0);
        sm_mul(A6, W, W);
        sm_comb(V, c_b13[6], A6, c_b13[4], A4, c_b13[2], A2, c_b13[0]);
        sm_comb(V, 1.0, V, 1.0, W, 0.0, Z, 0.0);
    } else {
        SmallMat<N> A8 = Z;
        if (m >= 9) sm_mul(A6, A2, A8);
        const double *bc = (m == 3) ? c_b3 : (m == 5) ? c_b5 : (m == 7) ? c_b7 : c_b9;
        const double b5 = m >= 5 ? bc[5] : 0.0, b4 = m >= 5 ? bc[4] : 0.0;
        const double b7 = m >= 7 ? bc[7] : 0.0, b6 = m >= 7 ? bc[6] : 0.0;
        const double b9 = m >= 9 ? bc[9] : 0.0, b8 = m >= 9 ? bc[8] : 0.0;
        sm_comb(W, bc[3], A2, b5, A4, b7, A6, bc[1]);
        sm_comb(W, 1.0, W, b9, A8, 0.0, Z, 0.0);
        sm_comb(V, bc[2], A2, b4, A4, b6, A6, bc[0]);
        sm_comb(V, 1.0, V, b8, A8, 0.0, Z, 0.0);
        sm_mul(A, W, U);
    }
    // solve (V - U) X = (V + U)
    SmallMat<N> M, X;
    sm_comb(M, 1.0, V, -1.0, U, 0.0, Z, 0.0);
    sm_comb(X, 1.0, V, 1.0, U, 0.0, Z, 0.0);
    bool singular = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // partial pivoting: bring the largest |M[i][k]|, i >= k, to row k
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const bool sw = fabs(M.a[i][k]) > fabs(M.a[k][k]);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double mk = M.a[k][j], mi = M.a[i][j];
                M.a[k][j] = sw ? mi : mk;
                M.a[i][j] = sw ? mk : mi;
                const double xk = X.a[k][j], xi = X.a[i][j];
                X.a[k][j] = sw ? xi : xk;
                X.a[i][j] = sw ? xk : xi;
            }
        }
        singular |= !(fabs(M.a[k][k]) > 0.0);
        const double rinv = 1.0 / M.a[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double f = M.a[i][k] * rinv;
#pragma unroll
            for (int j = k + 1; j < N; ++j) M.a[i][j] = fma(-f, M.a[k][j], M.a[i][j]);
#pragma unroll
            for (int j = 0; j < N; ++j) X.a[i][j] = fma(-f, X.a[k][j], X.a[i][j]);
        }
    }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        const double rinv = 1.0 / M.a[k][k];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double acc = X.a[k][j];
#pragma unroll
            for (int i = k + 1; i < N; ++i) acc = fma(-M.a[k][i], X.a[i][j], acc);
            X.a[k][j] = acc * rinv;
        }
    }
    for (int q = 0; q < s; ++q) sm_mul(X, X, X);
    if (singular) {
        if (info) info[2 * b] = -1;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) X.a[i][j] = __builtin_nan("");
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            Pb[i * N + j] = X.a[i][j];
            if (step >= 0) Pfrag[(long)step * NN + i * N + j] = X.a[i][j];
        }
}

}  // namespace

int rt_launch_expm(rt_ctx *ctx, int64_t n, int64_t count, const double *d_Q,
                   const int32_t *d_qidx, const double *d_t, double *d_P,
                   int32_t *d_info, const int32_t *d_step_of_node, int frag_kind,
                   double *d_Pfrag, const rt_reduce_args *fused_reduce)
{
    // one extra workgroup when the launch carries the pending reduction of a batch
    const rt_reduce_args red = fused_reduce ? *fused_reduce : rt_reduce_args();
    const unsigned extra = red.partial ? 1u : 0u;
    if (n < 1 || n > RT_MAX_EXPM_STATES) {
        rt_set_error("expm: n=%lld outside 1..%d", (long long)n, RT_MAX_EXPM_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    if (count <= 0 && !extra) return RT_OK;
    const int ld = (int)n | 1;
    const size_t lds = (size_t)5 * n * ld * 8 + (128 + 128 + 64) * 8 + (8 + 64) * 4;
    size_t &attr_lds = ctx->expm_attr_lds;
    if (lds > attr_lds) {
        RT_HIP(hipFuncSetAttribute((const void *)expm_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_lds = lds;
    }
    if (n <= 4 && !getenv("RAOTEH_EXPM_NO_SMALL")) {
        hipEvent_t ev = nullptr;
        rt_time_begin(ctx, RT_K_EXPM, "expm_small_lane_per_matrix", &ev);
        const unsigned grid = (unsigned)((count + 255) / 256) + extra;
        const int *son = frag_kind == 0 ? d_step_of_node : nullptr;
#define RT_SMALL(NV)                                                                 \
        RT_LAUNCH_TIMED(ctx, expm_small_kernel<NV>, dim3(grid), dim3(256), 0, \
                           (int)count, d_Q, d_qidx, d_t, d_P, d_info, son, d_Pfrag, red)
        switch ((int)n) {
        case 1: RT_SMALL(1); break;
        case 2: RT_SMALL(2); break;
        case 3: RT_SMALL(3); break;
        default: RT_SMALL(4); break;
        }
#undef RT_SMALL
        RT_HIP(hipGetLastError());
        rt_time_end(ctx, RT_K_EXPM, ev);
        return RT_OK;
    }
    // default: the Taylor / Paterson-Stockmeyer kernel (products only);
    // RAOTEH_EXPM=pade keeps the Pade + register Gauss-Jordan kernel (A/B runs, soak)
    const char *which = getenv("RAOTEH_EXPM");
    if (!(which && strcmp(which, "pade") == 0)) {
        const size_t lds_t = (size_t)5 * n * ld * 8 + 4 * 8;
        const int nt = (int)((n + 15) / 16);
        hipEvent_t ev = nullptr;
        rt_time_begin(ctx, RT_K_EXPM, "expm_taylor_ps_mfma", &ev);
#define RT_TAYLOR(NTV)                                                                          \
        do {                                                                                    \
            if (lds_t > ctx->expm_ts_attr_lds[NTV - 1]) {                                       \
                RT_HIP(hipFuncSetAttribute((const void *)expm_taylor_kernel<NTV>,               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                           (int)lds_t));                                        \
                ctx->expm_ts_attr_lds[NTV - 1] = lds_t;                                         \
            }                                                                                   \
            RT_LAUNCH_TIMED(ctx, expm_taylor_kernel<NTV>, dim3((unsigned)count + extra),        \
                            dim3(TPB), lds_t, (int)n, d_Q, d_qidx, d_t, d_P, d_info,            \
                            d_step_of_node, frag_kind, d_Pfrag, red);                           \
        } while (0)
        switch (nt) {
        case 1: RT_TAYLOR(1); break;
        case 2: RT_TAYLOR(2); break;
        case 3: RT_TAYLOR(3); break;
        default: RT_TAYLOR(4); break;
        }
#undef RT_TAYLOR
        RT_HIP(hipGetLastError());
        rt_time_end(ctx, RT_K_EXPM, ev);
        return RT_OK;
    }
    hipEvent_t ev = nullptr;
    rt_time_begin(ctx, RT_K_EXPM, "expm_mfma_regsolve", &ev);
    RT_LAUNCH_TIMED(ctx, expm_kernel, dim3((unsigned)count + extra), dim3(TPB), lds,
                       (int)n, d_Q, d_qidx, d_t, d_P, d_info, d_step_of_node, frag_kind,
                       d_Pfrag, red);
    RT_HIP(hipGetLastError());
    rt_time_end(ctx, RT_K_EXPM, ev);
    return RT_OK;
}
