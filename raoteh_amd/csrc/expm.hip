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

#include <algorithm>
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
            double *__restrict__ Pfrag, double *__restrict__ Pquad, rt_reduce_args red)
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
            if (Pquad)
                for (int e = tid; e < KS * KS * 16; e += TPB) Pquad[(long)step * rt_quad_stride(n) + e] = 0.0;
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
        if (Pquad) {
            // Pquad[step][rq][kk][k][i] = P[4 rq + i][4 kk + k]
            const int tq = KS * KS * 16;
            for (int e = tid; e < tq; e += TPB) {
                const int i = e & 3, k = (e >> 2) & 3, blk = e >> 4;
                const int row = 4 * (blk / KS) + i, col = 4 * (blk % KS) + k;
                Pquad[(long)step * rt_quad_stride(n) + e] = (row < n && col < n) ? Xb[row * ld + col] : 0.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// n > 4, default: scaling and squaring around a TAYLOR polynomial evaluated by
// Paterson-Stockmeyer -- matrix products only, no linear solve.
//
// The [m/m] Pade kernel above spends half of its time in the n pivot steps of the
// solve (two workgroup barriers each, a chain of n dependent rank-1 updates that the
// matrix pipe cannot help with: 50 of 103 us at n = 61).  A Taylor polynomial of
// degree m = 3 q costs 2 + (q - 1) products and nothing else:
//     powers   A^2 = A A,  A^3 = A A^2                                (2 products)
//     Horner   T = B_(q-1);  T = A^3 T + B_j, j = q-2..0              (q - 1 products)
//     with B_j = c_(3j) I + c_(3j+1) A + c_(3j+2) A^2 (elementwise),  c_i = 1 / i!
// Degree from ||A||_1 against theta_m, the largest norm for which the backward error
// of T_m stays below 2^-53 (computed as in Higham 2005 sec. 2 for the Taylor series of
// log(e^-x T_m(x)); the same numbers as Al-Mohy & Higham 2011, table 3.1):
//     m = 3, 6, 9, 12, 15  (2, 3, 4, 5, 6 products);
// beyond theta_15 = 0.64 one squaring per doubling of the norm (a squaring doubles the
// range for one product, a higher degree does not).  Entrywise accuracy against
// 60-digit arithmetic on the codon matrix: 6e-15 relative at t = 0.1 (scipy's Pade
// approximant: 5e-14).
//
// Four matrices (A, A^2, A^3, T) live zero-padded to 16 NT rows with an odd leading
// dimension 16 NT + 1 -- in LDS for n <= 64 (133 KB), in a per-workgroup slice of global
// scratch (L2-resident) for 64 < n <= 128, the order of the Frechet blocks of the codon
// model -- so that no operand fetch needs a bounds check.  The four waves tile the
// output 2 x 2; each holds RH x RH output tiles (RH = ceil(NT / 2)) and per k-step
// fetches RH slices of X and RH of Y for RH^2 MFMAs, the fetches of k-step kk + 1 issued
// before the MFMAs of k-step kk.  "+ B_j" is applied when the product is stored.
// ---------------------------------------------------------------------------

__constant__ double c_theta_taylor[5] = {1.3863479e-5, 9.0656564e-3, 8.9577602e-2,
                                         2.9961589e-1, 6.4108352e-1};
// 1 / i!, i = 0..15
__constant__ double c_inv_fact[16] = {
    1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
    1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0,
    1.0 / 6227020800.0, 1.0 / 87178291200.0, 1.0 / 1307674368000.0};

// C = X * Y (+ cf0 I + cf1 P1 + cf2 P2 when cf != nullptr) on the matrix pipe.  All
// matrices RN x RN (RN = 16 NT), leading dimension LD = RN + 1, zero outside n x n.
// C may alias X and / or Y (results are held in registers across a barrier).
// The elements of a matrix at the positions of this lane's accumulator tiles (the D
// layout of v_mfma_f64_16x16x4: register r of tile (m, j) on lane l is row 16m + 4r + (l >> 4),
// column 16j + (l & 15)).  The positions are the same for every product, so A and A^2 are
// read into registers once and the addend of a Horner step is arithmetic only -- read from
// LDS at every step it cost 4 000 of a product's 9 300 clocks (phase stamps,
// RAOTEH_EXPM_TRACE).
template <int NT>
struct lane_tiles {
    double v[(NT + 1) / 2][(NT + 1) / 2][4];
};

template <int NT>
__device__ __forceinline__ void load_tiles(const double *M, lane_tiles<NT> &t)
{
    constexpr int RH = (NT + 1) / 2;
    constexpr int LD = 16 * NT + 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int u = 0; u < RH; ++u)
#pragma unroll
        for (int v = 0; v < RH; ++v) {
            const int m = wr * RH + u, j = wc * RH + v;
            const bool ok = m < NT && j < NT;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                t.v[u][v][r] = ok ? M[(16 * m + 4 * r + lq) * LD + 16 * j + lr] : 0.0;
        }
}

template <int NT>
__device__ __forceinline__ void mm_blocked(const double *X, const double *Y, double *C,
                                           const double *cf, const double *P1, const double *P2,
                                           const lane_tiles<NT> *R1 = nullptr,
                                           const lane_tiles<NT> *R2 = nullptr,
                                           lane_tiles<NT> *out = nullptr,
                                           const double *top = nullptr, double *C2 = nullptr)
{
    constexpr int RH = (NT + 1) / 2;
    constexpr int LD = 16 * NT + 1;
    constexpr int KS = 4 * NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    double4_t acc[RH][RH];
    const double *ap[RH];
    const double *bp[RH];
    bool rok[RH], cok[RH];
#pragma unroll
    for (int u = 0; u < RH; ++u) {
        const int m = wr * RH + u, j = wc * RH + u;
        rok[u] = m < NT;                       // wave-uniform: NT odd leaves a slot empty
        cok[u] = j < NT;
        ap[u] = X + (16 * (rok[u] ? m : 0) + lr) * LD + lq;
        bp[u] = Y + lq * LD + 16 * (cok[u] ? j : 0) + lr;
#pragma unroll
        for (int v = 0; v < RH; ++v) acc[u][v] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    // the addend of a Horner step seeds the accumulators (added after the k loop it waited
    // for the matrix pipe to drain and stood between the last MFMA and the barrier)
    if (cf) {
        const double c0 = cf[0], c1 = cf[1], c2 = cf[2];
#pragma unroll
        for (int u = 0; u < RH; ++u)
#pragma unroll
            for (int v = 0; v < RH; ++v)
                if (rok[u] && cok[v]) {
                    const int m = wr * RH + u, j = wc * RH + v;
                    const int col = 16 * j + lr;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * m + 4 * r + lq;
                        const int o = row * LD + col;
                        // (explicit fma: the same contraction in every variant of the kernel)
                        double w = c1 * (R1 ? R1->v[u][v][r] : P1[o]);
                        if (R2) w = fma(c2, R2->v[u][v][r], w);
                        else if (P2) w = fma(c2, P2[o], w);
                        // the identity only inside n x n: P1 (= A) is zero outside, so
                        // the diagonal of the padding stays as cf[3] says (0 or c0)
                        if (row == col && (double)row < cf[3]) w += c0;
                        acc[u][v][r] = w;
                    }
                }
    }
    double a0[RH], b0[RH], a1[RH], b1[RH];
#pragma unroll
    for (int u = 0; u < RH; ++u) {
        a0[u] = ap[u][0];
        b0[u] = bp[u][0];
        a1[u] = ap[u][KS > 1 ? 4 : 0];
        b1[u] = bp[u][KS > 1 ? 4 * LD : 0];
    }
    for (int kk = 0; kk < KS; ++kk) {
        // operands two k-steps ahead (the last iterations re-read the last k-step's)
        const int kn = kk + 2 < KS ? kk + 2 : KS - 1;
        double a2[RH], b2[RH];
#pragma unroll
        for (int u = 0; u < RH; ++u) { a2[u] = ap[u][4 * kn]; b2[u] = bp[u][4 * kn * LD]; }
        __builtin_amdgcn_sched_barrier(0);      // the reads stay ahead of the MFMAs (see mm_half)
#pragma unroll
        for (int u = 0; u < RH; ++u)
#pragma unroll
            for (int v = 0; v < RH; ++v)
                acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[v], acc[u][v], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < RH; ++u) { a0[u] = a1[u]; b0[u] = b1[u]; a1[u] = a2[u]; b1[u] = b2[u]; }
    }
    if (out) {
#pragma unroll
        for (int u = 0; u < RH; ++u)
#pragma unroll
            for (int v = 0; v < RH; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) out->v[u][v][r] = acc[u][v][r];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RH; ++u)
#pragma unroll
        for (int v = 0; v < RH; ++v)
            if (rok[u] && cok[v]) {
                const int m = wr * RH + u, j = wc * RH + v;
                const int col = 16 * j + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * m + 4 * r + lq;
                    C[row * LD + col] = acc[u][v][r];
                    // the top block of the polynomial from the product just made (A^3) and
                    // the register copies of A and A^2: top = {c0, c1, c2, c3, n}
                    if (top) {
                        double w = top[1] * R1->v[u][v][r];
                        w = fma(top[2], R2->v[u][v][r], w);
                        w = fma(top[3], acc[u][v][r], w);
                        C2[row * LD + col] = w + ((row == col && (double)row < top[4]) ? top[0] : 0.0);
                    }
                }
            }
    __syncthreads();
}

// SPLIT (NT = 4, few matrices): TWO workgroups per matrix.  With 126 edges of a 64-leaf tree
// the kernel above occupies 126 of 256 CUs and its time is a chain of 5-6 dependent 64^3
// products.  A Horner step T <- A^3 T + B_j only needs the COLUMNS of T it produces, so after
// A, A^2 and A^3 (computed by both workgroups, in full) workgroup h of a pair runs the Horner
// steps on columns 32 h .. 32 h + 31 alone -- 4 x 2 output tiles, one row tile and two MFMAs
// per k-step and wave instead of 2 x 2 tiles and four -- and stores that half of the result.
// No exchange between the two, the same arithmetic per entry (bit-identical results).  When the
// matrix needs squarings (which need all of T) both workgroups run the whole chain.
struct half_tiles {
    double v[2][4];          // column tiles 2 h, 2 h + 1 of this wave's row tile
};

__device__ __forceinline__ void load_half_tiles(const double *M, half_tiles &t, int h)
{
    constexpr int LD = 65;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            t.v[v][r] = M[(16 * wave + 4 * r + lq) * LD + 16 * (2 * h + v) + lr];
}

// (RAOTEH_EXPM_TRACE: finer stamps of one Horner half product, slots 8..13)
__device__ unsigned long long rt_expm_fine[8];
#define RT_FINE(k) if (fine && threadIdx.x == 0) rt_expm_fine[k] = __builtin_readcyclecounter()

// C[:, half h] = X * Y[:, half h] + cf0 I + cf1 R1 + cf2 R2 (64 x 64 matrices, LD 65);
// C may alias Y
__device__ __forceinline__ void mm_half(const double *X, const double *Y, double *C,
                                        double c0, double c1, double c2, double nd,
                                        const half_tiles &R1, const half_tiles &R2, int h,
                                        bool fine = false)
{
    RT_FINE(0);
    constexpr int LD = 65, KS = 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // row tile
    const int lr = lane & 15, lq = lane >> 4;
    double4_t acc[2];
    const double *ap = X + (16 * wave + lr) * LD + lq;
    const double *bp = Y + lq * LD + 32 * h + lr;
    // operands two k-steps ahead (one wave per SIMD: nobody else hides the LDS latency, and a
    // k-step is only two MFMAs long); the first ones are requested before the accumulators are
    // seeded, and the coefficients of the step arrive in registers (read from LDS here, four
    // dependent-looking reads stood in front of the first MFMA: 1 200-1 500 clocks of a half
    // product's 4 700, fine stamps of RAOTEH_EXPM_TRACE)
    double a0 = ap[0], b00 = bp[0], b01 = bp[16];
    double a1 = ap[4], b10 = bp[4 * LD], b11 = bp[4 * LD + 16];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int col = 16 * (2 * h + v) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + 4 * r + lq;
            double w = c1 * R1.v[v][r];
            w = fma(c2, R2.v[v][r], w);
            if (row == col && (double)row < nd) w += c0;
            acc[v][r] = w;
        }
    }
    RT_FINE(1);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int kn = kk + 2 < KS ? kk + 2 : KS - 1;
        const double a2 = ap[4 * kn], b20 = bp[4 * kn * LD], b21 = bp[4 * kn * LD + 16];
        // (pinned: left alone the scheduler sinks every operand read to just before its MFMA
        // and waits for it in full -- read, wait, two MFMAs, read, wait: 103 clocks per MFMA)
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b00, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b01, acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a1; b00 = b10; b01 = b11;
        a1 = a2; b10 = b20; b11 = b21;
    }
    RT_FINE(2);
    __syncthreads();
    RT_FINE(3);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int col = 16 * (2 * h + v) + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(16 * wave + 4 * r + lq) * LD + col] = acc[v][r];
    }
    RT_FINE(4);
    __syncthreads();
    RT_FINE(5);
}

// GLOBAL = false: the four matrices in LDS (n <= 64); true: in scratch[blockIdx] (n <= 128)
// RAOTEH_EXPM_TRACE=1: workgroup 1 of the Taylor kernel stamps the shader clock at its
// phase boundaries (printed by the launcher): diagnostics
__device__ int rt_expm_trace_on = 0;
__device__ unsigned long long rt_expm_trace[8];
#define RT_EXPM_STAMP(k)                                                              \
    if (trace_on && blockIdx.x == (SPLIT ? 2 : 1) && threadIdx.x == 0)                \
        rt_expm_trace[k] = __builtin_readcyclecounter()

template <int NT, bool GLOBAL, bool SPLIT = false>
__global__ void __launch_bounds__(TPB)
expm_taylor_kernel(int n, const double *__restrict__ Q, const int *__restrict__ qidx,
                   const double *__restrict__ tt, double *__restrict__ P,
                   int *__restrict__ info, const int *__restrict__ step_of_node,
                   int frag_kind, double *__restrict__ Pfrag, double *__restrict__ Pquad,
                   double *__restrict__ scratch, rt_reduce_args red)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (red.partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction
        rt_reduce_partials_body(red.partial, red.npartials, red.totals, red.nsites);
        return;
    }
    constexpr int RN = 16 * NT;
    constexpr int LD = RN + 1;
    constexpr int MSZ = RN * LD;
    __shared__ double colsum[RN];
    const int trace_on = rt_expm_trace_on;
    RT_EXPM_STAMP(0);
    double *B0 = GLOBAL ? scratch + (size_t)blockIdx.x * 4 * MSZ : (double *)smem;   // A
    double *B1 = B0 + MSZ;                     // A^2
    double *B2 = B1 + MSZ;                     // A^3
    double *B3 = B2 + MSZ;                     // T

    static_assert(!SPLIT || (NT == 4 && !GLOBAL), "two workgroups per matrix: 49 <= n <= 64");
    const int b = SPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int half = SPLIT ? (int)(blockIdx.x & 1) : 0;      // which columns this workgroup stores
    const int tid = threadIdx.x;
    const int nn = n * n;
    const int KSn = (n + 3) / 4, NTn = (n + 15) / 16;      // of the n x n matrix (Pfrag)
    double *Pb = P + (long)b * nn;
    const int qi = qidx[b];
    const int step = step_of_node ? step_of_node[b] : -1;
    // LDS-resident sizes: rate matrix 0 is fetched before the workgroup knows which matrix
    // it uses (one rate matrix for all edges is the common case; the others fetch again), so
    // that the index and the matrix are one memory round trip, not two
    constexpr int RP0 = GLOBAL ? 1 : TPB / RN;
    constexpr int PER0 = GLOBAL ? 1 : (RN + RP0 - 1) / RP0;
    double q0[PER0];
    if (!GLOBAL) {
        const int jc = tid % RN, g = tid / RN;
#pragma unroll
        for (int u = 0; u < PER0; ++u) {
            const int i = g + u * RP0;
            q0[u] = (g < RP0 && i < n && jc < n) ? Q[i * n + jc] : 0.0;
        }
    }
    const double t_early = tt[b];
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
        for (int e = tid; e < nn; e += TPB) Pb[e] = 0.0;
        if (info && tid == 0) { info[2 * b] = 0; info[2 * b + 1] = 0; }
        if (step >= 0 && frag_kind == 0)
            for (int e = tid; e < nn; e += TPB) Pfrag[(long)step * nn + e] = 0.0;
        if (step >= 0 && frag_kind == 1) {
            const int total = NTn * ((KSn + 1) / 2) * 128;
            for (int e = tid; e < total; e += TPB) Pfrag[(long)step * total + e] = 0.0;
            if (Pquad)
                for (int e = tid; e < KSn * KSn * 16; e += TPB)
                    Pquad[(long)step * rt_quad_stride(n) + e] = 0.0;
        }
        return;
    }
    const double *Qb = Q + (long)qi * nn;
    const double t = t_early;
    // A = Q t, zero-padded.  Thread = (column j, row group g): rows g, g + RP, ... of one
    // column -- consecutive threads read consecutive addresses, no index division, all of a
    // thread's loads in flight together (n <= 64: 16 of them), and the column's |.| sum
    // falls out of the registers: ||A||_1 = max column sum without re-reading LDS.
    double nrm = 0.0;
    if (!GLOBAL) {
        constexpr int RP = TPB / RN;                 // row groups (4 at RN = 64)
        constexpr int PER = (RN + RP - 1) / RP;      // rows per thread
        __shared__ double colpart[RP][RN];
        const int jc = tid % RN, g = tid / RN;
        const bool active = g < RP;
        double v[PER];
        double part = 0.0;
        if (qi > 0) {                                // (block-uniform)
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = g + u * RP;
                q0[u] = (active && i < n && jc < n) ? Qb[i * n + jc] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) v[u] = q0[u] * t;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = g + u * RP;
            if (active && i < RN) B0[i * LD + jc] = v[u];
            part += fabs(v[u]);
        }
        if (active) colpart[g][jc] = part;
        __syncthreads();
        if (tid < RN) {
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < RP; ++r) sum += colpart[r][tid];
            colsum[tid] = sum;
        }
        __syncthreads();
        // max over the RN <= 64 column sums: one value per lane and a butterfly (a loop of RN
        // dependent LDS reads per thread was a quarter of this phase)
        {
            const int ln = tid & 63;
            nrm = ln < RN ? colsum[ln] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o, 64));
        }
    } else {
        constexpr int PER = (MSZ + TPB - 1) / TPB;
        for (int c0 = 0; c0 < PER; c0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = (c0 + u) * TPB + tid;
                const int i = e / LD, jx = e - i * LD;
                v[u] = (e < MSZ && i < n && jx < n) ? Qb[i * n + jx] * t : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = (c0 + u) * TPB + tid;
                if (e < MSZ) B0[e] = v[u];
            }
        }
        __syncthreads();
        // ||A||_1 = max column sum
        for (int jx = tid; jx < RN; jx += TPB) {
            double sum = 0.0;
            for (int i = 0; i < n; ++i) sum += fabs(B0[i * LD + jx]);
            colsum[jx] = sum;
        }
        __syncthreads();
        for (int jj = 0; jj < RN; ++jj) nrm = fmax(nrm, colsum[jj]);
    }
    RT_EXPM_STAMP(1);
    if (!(nrm < 1e300)) {                      // inf / NaN in Q * t (block-uniform)
        for (int e = tid; e < nn; e += TPB) Pb[e] = __builtin_nan("");
        if (info && tid == 0) { info[2 * b] = -1; info[2 * b + 1] = 0; }
        return;
    }
    int m = 15, s = 0;
    if (nrm <= c_theta_taylor[0]) m = 3;
    else if (nrm <= c_theta_taylor[1]) m = 6;
    else if (nrm <= c_theta_taylor[2]) m = 9;
    else if (nrm <= c_theta_taylor[3]) m = 12;
    else if (nrm > c_theta_taylor[4]) {
        int e;
        const double f = frexp(nrm / c_theta_taylor[4], &e);    // ratio = f * 2^e
        s = (f == 0.5) ? e - 1 : e;
        if (s < 0) s = 0;
    }
    m = __builtin_amdgcn_readfirstlane(m);
    s = __builtin_amdgcn_readfirstlane(s);
    if (info && tid == 0) { info[2 * b] = m; info[2 * b + 1] = s; }
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
        for (int e = tid; e < MSZ; e += TPB) B0[e] *= sc;
        __syncthreads();
    }
    const int q = m / 3;
    RT_EXPM_STAMP(2);
    lane_tiles<NT> ra, ra2;
    load_tiles<NT>(B0, ra);
    mm_blocked<NT>(B0, B0, B1, nullptr, nullptr, nullptr, nullptr, nullptr, &ra2);     // A^2
    // A^3, and with it T = B_(q-1) = c I + c A + c A^2 + c_m A^3 (the degree-m polynomial
    // is sum_{j<q} A^(3j) B_j plus c_m A^m, and c_m A^m = A^(3(q-1)) (c_m A^3): the top
    // block carries the A^3 term)
    __shared__ double tops[5];
    {
        const int base = 3 * (q - 1);
        if (tid == 0) {
            tops[0] = c_inv_fact[base];
            tops[1] = c_inv_fact[base + 1];
            tops[2] = c_inv_fact[base + 2];
            tops[3] = c_inv_fact[m];
            tops[4] = (double)n;
        }
        __syncthreads();
    }
    mm_blocked<NT>(B0, B1, B2, nullptr, nullptr, nullptr, &ra, &ra2, nullptr, tops, B3);
    RT_EXPM_STAMP(3);
    RT_EXPM_STAMP(4);
    // the coefficients of every Horner step at once (one barrier, not one per step)
    __shared__ double cfs_all[5][4];
    if (tid < 20) cfs_all[tid >> 2][tid & 3] = (tid & 3) == 3 ? (double)n
                                                             : c_inv_fact[3 * (tid >> 2) + (tid & 3)];
    __syncthreads();
    if (SPLIT && s == 0) {
        if constexpr (SPLIT) {
            half_tiles ha, ha2;
            load_half_tiles(B0, ha, half);
            load_half_tiles(B1, ha2, half);
            // (the coefficients of all four possible steps in registers: block-uniform loads
            // from constant memory, long before they are needed)
            double hc[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 3; ++e) hc[u][e] = c_inv_fact[3 * u + e];
            const double nd = (double)n;
#pragma unroll
            for (int u = 3; u >= 0; --u)
                if (u <= q - 2)            // T[:, half] = A^3 T[:, half] + B_u
                    mm_half(B2, B3, B3, hc[u][0], hc[u][1], hc[u][2], nd, ha, ha2, half,
                            trace_on && blockIdx.x == 2 && u == 0);
        }
    } else {
        for (int jj = q - 2; jj >= 0; --jj)
            mm_blocked<NT>(B2, B3, B3, cfs_all[jj], B0, B1, &ra, &ra2);    // T = A^3 T + B_j
    }
    RT_EXPM_STAMP(5);
    for (int r = 0; r < s; ++r) mm_blocked<NT>(B3, B3, B3, nullptr, nullptr, nullptr);
    RT_EXPM_STAMP(6);

    const double *Xb = B3;
    if constexpr (SPLIT) {
        // this workgroup's columns only: P in the reference's order, the A fragments of its
        // k-pairs (column c of P is k-step c / 4: columns 32 h .. 32 h + 31 = pairs 4 h .. 4 h + 3)
        const int jc = 32 * half + (tid & 31), g = tid >> 5;
        if (jc < n)
            for (int i = g; i < n; i += TPB / 32) Pb[i * n + jc] = Xb[i * LD + jc];
        if (step >= 0 && frag_kind == 1) {
            const int KP = (KSn + 1) / 2;
            const int total = NTn * KP * 128;
            for (int blk = tid >> 7; blk < NTn * 4; blk += TPB >> 7) {
                const int mm = blk >> 2, qq = 4 * half + (blk & 3);
                if (qq < KP) {
                    const int e = (mm * KP + qq) * 128 + (tid & 127);
                    const int e2 = e & 1;
                    const int ln = (e >> 1) & 63;
                    const int row = 16 * mm + (ln & 15);
                    const int col = 4 * (2 * qq + e2) + (ln >> 4);
                    Pfrag[(long)step * total + e] = (col < RN) ? Xb[row * LD + col] : 0.0;
                }
            }
        }
        RT_EXPM_STAMP(7);
        return;
    }
    if (!GLOBAL) {               // thread = (column, row group) as in the load: no division
        constexpr int RP = TPB / RN;
        const int jc = tid % RN, g = tid / RN;
        if (g < RP && jc < n)
            for (int i = g; i < n; i += RP) Pb[i * n + jc] = Xb[i * LD + jc];
    } else {
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n;
            Pb[e] = Xb[i * LD + j];
        }
    }
    if (step >= 0 && frag_kind == 0) {
        for (int e = tid; e < nn; e += TPB) {
            const int i = e / n, j = e - i * n;
            Pfrag[(long)step * nn + e] = Xb[i * LD + j];
        }
    } else if (step >= 0 && frag_kind == 1) {
        // Pfrag[step][m][q][lane][e2] = P[16m + (lane&15)][4(2q+e2) + (lane>>4)]; the
        // padding of Xb is zero, so no bounds check (KP pairs cover <= RN columns)
        const int KP = (KSn + 1) / 2;
        const int total = NTn * KP * 128;
        // 128 entries per (row tile, k-pair): the block index is wave-uniform arithmetic
        for (int blk = tid >> 7; blk < NTn * KP; blk += TPB >> 7) {
            const int e = blk * 128 + (tid & 127);
            const int e2 = e & 1;
            const int ln = (e >> 1) & 63;
            const int qq = blk % KP;
            const int mm = blk / KP;
            const int row = 16 * mm + (ln & 15);
            const int col = 4 * (2 * qq + e2) + (ln >> 4);
            Pfrag[(long)step * total + e] = (col < RN) ? Xb[row * LD + col] : 0.0;
        }
        if (Pquad) {
            // Pquad[step][rq][kk][k][i] = P[4 rq + i][4 kk + k] (the padding is zero)
            const int tq = KSn * KSn * 16;
            for (int e = tid; e < tq; e += TPB) {
                const int i = e & 3, k = (e >> 2) & 3, blk = e >> 4;
                const int row = 4 * (blk / KSn) + i, col = 4 * (blk % KSn) + k;
                Pquad[(long)step * rt_quad_stride(n) + e] = Xb[row * LD + col];
            }
        }
    }
    RT_EXPM_STAMP(7);
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

#define RT_SHARED_SOURCE(...) __VA_ARGS__
#include "expm_small.inc"
#undef RT_SHARED_SOURCE

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

// n <= 4, default: the Taylor / Paterson-Stockmeyer scheme of expm_taylor_kernel per lane
// (A^2, A^3, Horner in A^3, squarings): at most 6 small products and no solve, against
// up to 6 products plus an LU factorisation with predicated row swaps for the Pade form
// above -- this kernel is one wave of dependent register arithmetic, and on config 2 it
// sits in front of a 34 us pruning kernel in every step.
// (inv_fact_lit of expm_small.inc: 1 / i! as a compile-time expression -- with the Horner loop
// unrolled every coefficient is an instruction literal.  Indexed by the per-lane order m, the
// table in constant memory was a per-lane gather: one dependent memory round trip for the
// seed of the recurrence and one more per Horner step.)
template <int N>
__global__ void __launch_bounds__(256)
expm_small_taylor_kernel(int count, const double *__restrict__ Q, const int *__restrict__ qidx,
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
    // every load of the lane is issued at once: matrix 0 is fetched before the lane knows
    // which matrix it uses (one rate matrix for all edges is the common case; the others
    // pay a second fetch)
    const int qi = qidx[b];
    const int step = step_of_node ? step_of_node[b] : -1;
    const double t = tt[b];
    SmallMat<N> A;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) A.a[i][j] = Q[i * N + j];
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
#pragma unroll
        for (int e = 0; e < NN; ++e) Pb[e] = 0.0;
        if (info) { info[2 * b] = 0; info[2 * b + 1] = 0; }
        if (step >= 0)
#pragma unroll
            for (int e = 0; e < NN; ++e) Pfrag[(long)step * NN + e] = 0.0;
        return;
    }
    if (qi > 0) {
        const double *Qb = Q + (long)qi * NN;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) A.a[i][j] = Qb[i * N + j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) A.a[i][j] *= t;
    // the arithmetic itself: expm_small.inc, shared with the lane kernels that compute their
    // own transition matrices (jit.hip)
    SmallMat<N> X;
    int m, s;
    rt_expm_small_taylor<N>(A, X, m, s);
    if (info) { info[2 * b] = m; info[2 * b + 1] = s; }
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
                   double *d_Pfrag, const rt_reduce_args *fused_reduce, double *d_Pquad)
{
    // one extra workgroup when the launch carries the pending reduction of a batch
    const rt_reduce_args red = fused_reduce ? *fused_reduce : rt_reduce_args();
    const unsigned extra = red.partial ? 1u : 0u;
    if (n < 1 || n > RT_MAX_EXPM_STATES) {
        rt_set_error("expm: n=%lld outside 1..%d", (long long)n, RT_MAX_EXPM_STATES);
        return RT_ERR_UNSUPPORTED;
    }
    const int64_t RT_MAX_PADE_STATES = 62;     // five n x (n|1) LDS buffers
    if (count <= 0 && !extra) return RT_OK;
    const int ld = (int)n | 1;
    const size_t lds = (size_t)5 * n * ld * 8 + (128 + 128 + 64) * 8 + (8 + 64) * 4;
    size_t &attr_lds = ctx->expm_attr_lds;
    if (n <= RT_MAX_PADE_STATES && lds > attr_lds) {
        RT_HIP(hipFuncSetAttribute((const void *)expm_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        attr_lds = lds;
    }
    if (n <= 4 && !getenv("RAOTEH_EXPM_NO_SMALL")) {
        hipEvent_t ev = nullptr;
        const char *wh = getenv("RAOTEH_EXPM");
        const bool pade = wh && strcmp(wh, "pade") == 0;
        rt_time_begin(ctx, RT_K_EXPM, pade ? "expm_small_lane_per_matrix_pade"
                                           : "expm_small_lane_per_matrix", &ev);
        const unsigned grid = (unsigned)((count + 255) / 256) + extra;
        const int *son = frag_kind == 0 ? d_step_of_node : nullptr;
        if (!pade) {
#define RT_SMALLT(NV)                                                                \
            RT_LAUNCH_TIMED(ctx, expm_small_taylor_kernel<NV>, dim3(grid), dim3(256), 0, \
                            (int)count, d_Q, d_qidx, d_t, d_P, d_info, son, d_Pfrag, red)
            switch ((int)n) {
            case 1: RT_SMALLT(1); break;
            case 2: RT_SMALLT(2); break;
            case 3: RT_SMALLT(3); break;
            default: RT_SMALLT(4); break;
            }
#undef RT_SMALLT
            RT_HIP(hipGetLastError());
            rt_time_end(ctx, RT_K_EXPM, ev);
            return RT_OK;
        }
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
    if (n > RT_MAX_PADE_STATES || !(which && strcmp(which, "pade") == 0)) {
        const int nt = (int)((n + 15) / 16);
        const size_t msz = (size_t)(16 * nt) * (16 * nt + 1);
        const bool global = nt > 4;
        const size_t lds_t = global ? 0 : 4 * msz * 8;
        double *scratch = nullptr;
        if (global) {
            // four matrices per workgroup in global scratch (grow-only, per context)
            const size_t need = (size_t)count * 4 * msz * 8;
            if (need > ctx->expm_scratch_bytes) {
                RT_HIP(hipStreamSynchronize(ctx->stream));
                hipFree(ctx->d_expm_scratch);
                ctx->d_expm_scratch = nullptr;
                ctx->expm_scratch_bytes = 0;
                RT_HIP(hipMalloc((void **)&ctx->d_expm_scratch, need));
                ctx->expm_scratch_bytes = need;
            }
            scratch = ctx->d_expm_scratch;
        }
        // 64 < n <= 128: the LDS-resident kernel (one matrix in LDS, the rest in registers);
        // RAOTEH_EXPM_WIDE=0: the global-scratch form.  Few matrices: two workgroups each
        const char *wv = getenv("RAOTEH_EXPM_WIDE");
        if (global && !(wv && atoi(wv) == 0) && !d_Pquad) {
            bool split2 = 2 * count + extra <= (int64_t)std::max(2, ctx->num_cus);
            if (const char *v = getenv("RAOTEH_EXPM_SPLIT")) split2 = atoi(v) != 0;
            const size_t wgs = (size_t)count * (split2 ? 2 : 1);
            const size_t need = wgs * ((nt + 1) / 2) * 4 * nt * 256 * 8;   // A, A^2 in D layout per workgroup
            if (need > ctx->expm_scratch_bytes) {
                RT_HIP(hipStreamSynchronize(ctx->stream));
                hipFree(ctx->d_expm_scratch);
                ctx->d_expm_scratch = nullptr;
                ctx->expm_scratch_bytes = 0;
                RT_HIP(hipMalloc((void **)&ctx->d_expm_scratch, need));
                ctx->expm_scratch_bytes = need;
            }
            RT_TRY(rt_expm_wide_launch(ctx, nt, split2, wgs + extra, n, d_Q, d_qidx, d_t, d_P, d_info,
                                       d_step_of_node, frag_kind, d_Pfrag, red));
            return RT_OK;
        }
        hipEvent_t ev = nullptr;
        rt_time_begin(ctx, RT_K_EXPM, global ? "expm_taylor_ps_mfma_global" : "expm_taylor_ps_mfma",
                      &ev);
#define RT_TAYLOR(NTV, GL)                                                                      \
        do {                                                                                    \
            if (lds_t > ctx->expm_ts_attr_lds[NTV - 1]) {                                       \
                RT_HIP(hipFuncSetAttribute((const void *)expm_taylor_kernel<NTV, GL>,           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                           (int)lds_t));                                        \
                ctx->expm_ts_attr_lds[NTV - 1] = lds_t;                                         \
            }                                                                                   \
            RT_LAUNCH_TIMED(ctx, (expm_taylor_kernel<NTV, GL>), dim3((unsigned)count + extra),  \
                            dim3(TPB), lds_t, (int)n, d_Q, d_qidx, d_t, d_P, d_info,            \
                            d_step_of_node, frag_kind, d_Pfrag, d_Pquad, scratch, red);         \
        } while (0)
        // few 49..64-state matrices (the edges of one tree): two workgroups per matrix, the
        // Horner steps in column halves (RAOTEH_EXPM_SPLIT=0 / 1 overrides)
        bool split2 = nt == 4 && 2 * count + extra <= (int64_t)std::max(2, ctx->num_cus) && !d_Pquad;
        if (const char *v = getenv("RAOTEH_EXPM_SPLIT")) split2 = nt == 4 && !d_Pquad && atoi(v) != 0;
        if (split2) {
            if (lds_t > ctx->expm_split_attr_lds) {
                RT_HIP(hipFuncSetAttribute((const void *)expm_taylor_kernel<4, false, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                ctx->expm_split_attr_lds = lds_t;
            }
            snprintf(ctx->slots[RT_K_EXPM].name, sizeof(ctx->slots[RT_K_EXPM].name),
                     "expm_taylor_ps_mfma_split2");
            RT_LAUNCH_TIMED(ctx, (expm_taylor_kernel<4, false, true>),
                            dim3(2u * (unsigned)count + extra), dim3(TPB), lds_t, (int)n, d_Q, d_qidx,
                            d_t, d_P, d_info, d_step_of_node, frag_kind, d_Pfrag, d_Pquad, scratch,
                            red);
        } else
        switch (nt) {
        case 1: RT_TAYLOR(1, false); break;
        case 2: RT_TAYLOR(2, false); break;
        case 3: RT_TAYLOR(3, false); break;
        case 4: RT_TAYLOR(4, false); break;
        case 5: RT_TAYLOR(5, true); break;
        case 6: RT_TAYLOR(6, true); break;
        case 7: RT_TAYLOR(7, true); break;
        default: RT_TAYLOR(8, true); break;
        }
#undef RT_TAYLOR
        RT_HIP(hipGetLastError());
        if (getenv("RAOTEH_EXPM_TRACE")) {
            static int armed = 0;
            unsigned long long tr[8];
            RT_HIP(hipStreamSynchronize(ctx->stream));
            if (armed) {
                RT_HIP(hipMemcpyFromSymbol(tr, HIP_SYMBOL(rt_expm_trace), sizeof tr));
                unsigned long long fn[8];
                RT_HIP(hipMemcpyFromSymbol(fn, HIP_SYMBOL(rt_expm_fine), sizeof fn));
                fprintf(stderr, "[raoteh_amd] last Horner half product: seed + first operands %llu, k loop "
                        "%llu, barrier %llu, write-back %llu, barrier %llu\n", fn[1] - fn[0],
                        fn[2] - fn[1], fn[3] - fn[2], fn[4] - fn[3], fn[5] - fn[4]);
                fprintf(stderr, "[raoteh_amd] expm trace (clocks, workgroup 1): load + norm %llu, order %llu, "
                        "A^2 A^3 %llu, block %llu, Horner %llu, squarings %llu, store %llu; total %llu\n",
                        tr[1] - tr[0], tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4],
                        tr[6] - tr[5], tr[7] - tr[6], tr[7] - tr[0]);
            }
            const int on = 1;
            RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rt_expm_trace_on), &on, sizeof on));
            armed = 1;
        }
        rt_time_end(ctx, RT_K_EXPM, ev);
        return RT_OK;
    }
    hipEvent_t ev = nullptr;
    rt_time_begin(ctx, RT_K_EXPM, "expm_mfma_regsolve", &ev);
    RT_LAUNCH_TIMED(ctx, expm_kernel, dim3((unsigned)count + extra), dim3(TPB), lds,
                       (int)n, d_Q, d_qidx, d_t, d_P, d_info, d_step_of_node, frag_kind,
                       d_Pfrag, d_Pquad, red);
    RT_HIP(hipGetLastError());
    rt_time_end(ctx, RT_K_EXPM, ev);
    return RT_OK;
}
