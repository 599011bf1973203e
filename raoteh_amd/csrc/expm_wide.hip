// expm for 64 < n <= 128 with ONE matrix in LDS (see below).  Its own translation unit: the
// rest of the expm kernels are compiled with -amdgpu-mfma-vgpr-form=1 (accumulators in VGPRs),
// which leaves this kernel 256 registers for 128 of accumulators and 128 of A operands and
// spills the rest; with the default form the accumulators live in the AGPR half of the file.
#include "common.h"
#include "reduce.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

// The lane-private copies of A and A^2 (D layout) are read through a buffer descriptor: one
// VGPR of lane offset for all of them and the entry's offset in the scalar operand.  As flat
// loads every one of the 128 entries got its own 64-bit address register, hoisted out of the
// Horner loop -- 256 registers of addresses, spilled and reloaded around every load.
struct wide_buf {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;                               // lane * 8
    __device__ __forceinline__ double load(int entry) const
    {
        const uint2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, entry * 512, 0);
        return __builtin_bit_cast(double, v);
    }
    __device__ __forceinline__ void store(int entry, double x) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2_t, x), rs, voff, entry * 512, 0);
    }
};

// (the constants of expm.hip's Taylor kernel)
__constant__ double c_theta_taylor[5] = {1.3863479e-5, 9.0656564e-3, 8.9577602e-2,
                                         2.9961589e-1, 6.4108352e-1};
// 1 / i!, i = 0..15
__constant__ double c_inv_fact[16] = {
    1.0, 1.0, 0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
    1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0,
    1.0 / 6227020800.0, 1.0 / 87178291200.0, 1.0 / 1307674368000.0};

// ---------------------------------------------------------------------------
// 64 < n <= 128, "wide": ONE matrix in LDS, the rest in registers
// ---------------------------------------------------------------------------
//
// The global-scratch form above keeps four 128 x 129 matrices per workgroup in a scratch
// buffer (528 KB each: the 126 edges of a tree are 66 MB, which lives in the Infinity Cache, not
// in L2) and reads both operands of every product from there.  Here a workgroup is ceil(NT / 2)
// waves, one per SIMD with the whole register file, and wave w owns row tiles 2 w, 2 w + 1 of
// every product C = X Y:
//   X  its 32 rows as A operands in registers (2 x 4 NT doubles per lane), read once per product;
//   Y  the whole matrix in LDS in B-fragment order, Yl[k-step][column tile][lane] =
//      Y[4 kk + (lane >> 4)][16 j + (lane & 15)] -- one conflict-free ds_read_b64 per TWO MFMAs,
//      shared by the waves; a row of k-step blocks is padded by 4 doubles so that reading the
//      SAME image as an A operand (rows 16 m + (lane & 15), a transposed walk) is 2-way, not
//      8-way, conflicted: that is how A^3 and T become left operands without a second LDS matrix;
//   C  2 x NT tiles of accumulators; the D layout of row tile m is the B-fragment layout of
//      k-steps 4 m .. 4 m + 3, so a product is written back with plain stores.
// The addends of a Horner step (c0 I + c1 A + c2 A^2) seed the accumulators: A again from Q
// (the same two roundings), A^2 from a scratch copy each lane wrote itself.  Same arithmetic per
// entry as the global form (bit-identical; RAOTEH_EXPM_WIDE=0 selects the old kernel).
// SPLIT: two workgroups per matrix as for n <= 64 -- A^2, A^3 in full in both, the Horner steps
// on half of the column tiles each (no exchange); with squarings both run the whole chain.

template <int NT, int JC>
__device__ __forceinline__ void wide_kloop(const double (&xop)[2][4 * NT], const double *Yl, int jb,
                                           double4_t (&acc)[2][NT])
{
    constexpr int KS = 4 * NT, RS = NT * 64 + 4;
    const int lane = threadIdx.x & 63;
    const double *yp = Yl + jb * 64 + lane;
    double bc[JC], bn[JC];
#pragma unroll
    for (int v = 0; v < JC; ++v) bc[v] = yp[v * 64];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int kn = kk + 1 < KS ? kk + 1 : kk;
#pragma unroll
        for (int v = 0; v < JC; ++v) bn[v] = yp[kn * RS + v * 64];
        __builtin_amdgcn_sched_barrier(0);      // next k-step's operands are requested first
#pragma unroll
        for (int v = 0; v < JC; ++v) {
            acc[0][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[0][kk], bc[v], acc[0][v], 0, 0, 0);
            acc[1][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(xop[1][kk], bc[v], acc[1][v], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < JC; ++v) bc[v] = bn[v];
    }
}

// this wave's rows of the matrix in Yl as A operands:
// xop[u][kk] = M[16 (2 w + u) + (lane & 15)][4 kk + (lane >> 4)] (zero for a row tile >= NT)
template <int NT>
__device__ __forceinline__ void wide_read_aop(const double *Yl, double (&xop)[2][4 * NT])
{
    constexpr int RS = NT * 64 + 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int mt = 2 * w + u;
        const bool ok = mt < NT;
        const double *p = Yl + (4 * (ok ? mt : 0) + (lr >> 2)) * RS + (lr & 3) * 16 + lq;
#pragma unroll
        for (int kk = 0; kk < 4 * NT; ++kk) xop[u][kk] = ok ? p[(kk >> 2) * 64 + 4 * (kk & 3)] : 0.0;
    }
}

// column tiles jb .. jb + JC - 1 of this wave's row tiles -> Yl (B-fragment order)
template <int NT, int JC>
__device__ __forceinline__ void wide_store(double *Yl, int jb, const double4_t (&acc)[2][NT])
{
    constexpr int RS = NT * 64 + 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int mt = 2 * w + u;
        if (mt < NT) {
#pragma unroll
            for (int v = 0; v < JC; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) Yl[(4 * mt + r) * RS + (jb + v) * 64 + lane] = acc[u][v][r];
        }
    }
}

// A = (Q t) [2^-s] at this lane's D-layout positions of row tile mt, column tile j: the loads
// are unconditional (indices clamped into the matrix) and the padding is selected to zero
// afterwards -- a predicated load is a branch per entry
__device__ __forceinline__ void wide_a_tile(const double *Qb, int n, int mt, int j, double t, double sc,
                                            bool scaled, double (&a)[4])
{
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int col = 16 * j + lr;
    const int cc = col < n ? col : n - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * r + lq;
        const int rc = row < n ? row : n - 1;
        double v = Qb[rc * n + cc] * t;
        if (scaled) v *= sc;
        a[r] = (row < n && col < n) ? v : 0.0;
    }
}

// one Horner step on column tiles jb .. jb + JC - 1: T <- A^3 T + c0 I + c1 A + c2 A^2.  The
// addends come from the lane's scratch copies of A and A^2 (tile jb + v at entry
// (u * ST + sb + v) * 4 + r, A^2 8 NT entries further: all tiles, or only this half's).  (A read
// again from Q -- one matrix for all workgroups of the launch -- was slower than the private
// copy: 250 workgroups asking for the same lines at the same time; in registers the two
// halves' addends make the allocator spill 200 registers next to 128 of A operands.)
// A1LDS: A of this half's tiles sits in the OTHER half's tiles of Yl (column tiles jo + v), which
// hold nothing a Horner step in halves reads -- half of the addend traffic stays on the CU.
template <int NT, int JC, bool A1LDS = false>
__device__ __forceinline__ void wide_horner(const double (&xop)[2][4 * NT], double *Yl, int jb, int n,
                                            const wide_buf &SB, int ST, int sb, double c0, double c1,
                                            double c2, int jo = 0)
{
    constexpr int RS = NT * 64 + 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    double4_t acc[2][NT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int v = 0; v < JC; ++v) {
            const int col = 16 * (jb + v) + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (2 * w + u) + 4 * r + lq;
                const int mt = 2 * w + u;
                const double a1 = A1LDS ? (mt < NT ? Yl[(4 * mt + r) * RS + (jo + v) * 64 + lane] : 0.0)
                                        : SB.load((u * ST + sb + v) * 4 + r);
                const double a2 = SB.load(8 * NT + (u * ST + sb + v) * 4 + r);
                double wv = c1 * a1;
                wv = fma(c2, a2, wv);
                if (row == col && row < n) wv += c0;
                acc[u][v][r] = wv;
            }
        }
        // (one row tile's addends in flight at a time: both at once fill the register file)
        if (JC > NT / 2 + 1) asm volatile("" ::: "memory");
    }
    wide_kloop<NT, JC>(xop, Yl, jb, acc);
    __syncthreads();                             // every wave has read the old T
    wide_store<NT, JC>(Yl, jb, acc);
    __syncthreads();
}

// T = top1 A + top2 A^2 + top3 A^3 + top0 I on column tiles JB .. JB + JC - 1 (acc: A^3 in, T out)
// LDS_OUT (halves, equal halves): T goes straight to this half's tiles of Yl and the values of A
// just loaded to the other half's (column tiles jo + v), where the Horner steps read them
template <int NT, int JC, int JB, bool LDS_OUT>
__device__ __forceinline__ void wide_top(double4_t (&acc)[2][NT], int n, const wide_buf &SB, int ST,
                                         int sb, double top0, double top1, double top2, double top3,
                                         double *Yl = nullptr, int jo = 0)
{
    constexpr int RS = NT * 64 + 4;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int v = 0; v < JC; ++v) {
            const int col = 16 * (JB + v) + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (2 * w + u) + 4 * r + lq;
                const double a1 = SB.load((u * ST + sb + v) * 4 + r);
                const double a2 = SB.load(8 * NT + (u * ST + sb + v) * 4 + r);
                double wv = top1 * a1;
                wv = fma(top2, a2, wv);
                wv = fma(top3, acc[u][JB + v][r], wv);
                wv += (row == col && row < n) ? top0 : 0.0;
                if (LDS_OUT) {
                    const int mt = 2 * w + u;
                    if (mt < NT) {
                        Yl[(4 * mt + r) * RS + (JB + v) * 64 + lane] = wv;
                        Yl[(4 * mt + r) * RS + (jo + v) * 64 + lane] = a1;
                    }
                } else {
                    acc[u][JB + v][r] = wv;
                }
            }
        }
        if (JC > NT / 2 + 1) asm volatile("" ::: "memory");
    }
}

// RAOTEH_EXPM_TRACE=1: workgroup 0 stamps the shader clock at its phase boundaries
__device__ int rt_expm_wide_trace_on = 0;
__device__ unsigned long long rt_expm_wide_trace[12];
#define RT_WSTAMP(k)                                                              \
    if (trace_on && blockIdx.x == (SPLIT ? 2 : 1) && threadIdx.x == 0)            \
        rt_expm_wide_trace[k] = __builtin_readcyclecounter()

template <int NT, bool SPLIT>
__global__ void __launch_bounds__(64 * ((NT + 1) / 2))
expm_taylor_wide_kernel(int n, const double *__restrict__ Q, const int *__restrict__ qidx,
                        const double *__restrict__ tt, double *__restrict__ P,
                        int *__restrict__ info, const int *__restrict__ step_of_node,
                        int frag_kind, double *__restrict__ Pfrag, double *__restrict__ scratch,
                        rt_reduce_args red)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (red.partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction
        rt_reduce_partials_body(red.partial, red.npartials, red.totals, red.nsites);
        return;
    }
    constexpr int WV = (NT + 1) / 2;           // waves
    constexpr int TPBW = 64 * WV, RN = 16 * NT, KS = 4 * NT, RS = NT * 64 + 4;
    constexpr int JH = (NT + 1) / 2;           // column tiles of half 0 (half 1: NT - JH)
    double *Yl = (double *)smem;               // [KS][RS]
    __shared__ double colsum[RN];
    const int b = SPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int half = SPLIT ? (int)(blockIdx.x & 1) : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nn = n * n;
    const int KSn = (n + 3) / 4, NTn = (n + 15) / 16;
    double *Pb = P + (long)b * nn;
    const int qi = qidx[b];
    const int step = step_of_node ? step_of_node[b] : -1;
    const double t = tt[b];
    if (qi < 0) {                              // root slot: zeros (_density.py:171)
        if (half == 0) {
            for (int e = tid; e < nn; e += TPBW) Pb[e] = 0.0;
            if (info && tid == 0) { info[2 * b] = 0; info[2 * b + 1] = 0; }
            if (step >= 0 && frag_kind == 0)
                for (int e = tid; e < nn; e += TPBW) Pfrag[(long)step * nn + e] = 0.0;
            if (step >= 0 && frag_kind == 1) {
                const int total = NTn * ((KSn + 1) / 2) * 128;
                for (int e = tid; e < total; e += TPBW) Pfrag[(long)step * total + e] = 0.0;
            }
        }
        return;
    }
    const int trace_on = rt_expm_wide_trace_on;
    RT_WSTAMP(0);
    const double *Qb = Q + (long)qi * nn;
    // A = Q t, zero-padded, straight into B-fragment order (each lane the entries of its D layout)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int mt = 2 * w + u;
        if (mt < NT) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                double v[4];
                wide_a_tile(Qb, n, mt, j, t, 1.0, false, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) Yl[(4 * mt + r) * RS + j * 64 + lane] = v[r];
            }
        }
    }
    __syncthreads();
    // ||A||_1 = max column sum (rows in ascending order, as the global form adds them)
    if (tid < RN) {
        const double *cp = Yl + (tid >> 4) * 64 + (tid & 15);
        double sum = 0.0;
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) sum += fabs(cp[kk * RS + k4 * 16]);
        colsum[tid] = sum;
    }
    __syncthreads();
    double nrm = fmax(lane < RN ? colsum[lane] : 0.0, lane + 64 < RN ? colsum[lane + 64] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, o, 64));
    if (!(nrm < 1e300)) {                      // inf / NaN in Q * t (block-uniform)
        if (half == 0) {
            for (int e = tid; e < nn; e += TPBW) Pb[e] = __builtin_nan("");
            if (info && tid == 0) { info[2 * b] = -1; info[2 * b + 1] = 0; }
        }
        return;
    }
    RT_WSTAMP(1);
    int mdeg = 15, s = 0;
    if (nrm <= c_theta_taylor[0]) mdeg = 3;
    else if (nrm <= c_theta_taylor[1]) mdeg = 6;
    else if (nrm <= c_theta_taylor[2]) mdeg = 9;
    else if (nrm <= c_theta_taylor[3]) mdeg = 12;
    else if (nrm > c_theta_taylor[4]) {
        int e;
        const double f = frexp(nrm / c_theta_taylor[4], &e);    // ratio = f * 2^e
        s = (f == 0.5) ? e - 1 : e;
        if (s < 0) s = 0;
    }
    mdeg = __builtin_amdgcn_readfirstlane(mdeg);
    s = __builtin_amdgcn_readfirstlane(s);
    if (info && tid == 0 && half == 0) { info[2 * b] = mdeg; info[2 * b + 1] = s; }
    const double sc = ldexp(1.0, -s);
    if (s > 0) {
        // each lane rescales the entries it wrote
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mt = 2 * w + u;
            if (mt < NT) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Yl[(4 * mt + r) * RS + j * 64 + lane] *= sc;
            }
        }
        __syncthreads();
    }
    const int q = mdeg / 3;
    const int jb = half * JH;
    const bool halves = SPLIT && s == 0;       // the Horner steps on this half's column tiles only
    // entries [0, 8 NT): A, [8 NT, 16 NT): A^2, D layout; entry e of lane l at double e * 64 + l
    wide_buf SB;
    {
        double *sb = scratch + ((size_t)blockIdx.x * WV + w) * (4 * NT * 256);     // wave-uniform
        const unsigned long long a = (unsigned long long)sb;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        SB.rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0,
                                                  4 * NT * 256 * 8, 0x00020000);
        SB.voff = (unsigned)lane * 8u;
    }

    RT_WSTAMP(2);
    double xop[2][KS];
    wide_read_aop<NT>(Yl, xop);                // A, this wave's rows
    double4_t acc[2][NT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[u][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    RT_WSTAMP(3);
    wide_kloop<NT, NT>(xop, Yl, 0, acc);       // A^2 = A A
    RT_WSTAMP(4);
    // A (still in Yl, at the positions this lane wrote) and A^2 at this lane's D-layout
    // positions -> the scratch copies the Horner steps add them from: this half's tiles only
    // (entries (u * JH + v) * 4 + r) or all (entries (u * NT + j) * 4 + r)
    if (halves) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mt = 2 * w + u;
#pragma unroll
            for (int v = 0; v < JH; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = mt < NT && jb + v < NT;
                    SB.store((u * JH + v) * 4 + r,
                             ok ? Yl[(4 * mt + r) * RS + (ok ? jb + v : 0) * 64 + lane] : 0.0);
                    const double hi = JH + v < NT ? acc[u][JH + v < NT ? JH + v : 0][r] : 0.0;
                    SB.store(8 * NT + (u * JH + v) * 4 + r, half ? hi : acc[u][v][r]);
                }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mt = 2 * w + u;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    SB.store((u * NT + j) * 4 + r, mt < NT ? Yl[(4 * mt + r) * RS + j * 64 + lane] : 0.0);
                    SB.store(8 * NT + (u * NT + j) * 4 + r, acc[u][j][r]);
                }
        }
    }
    __syncthreads();
    wide_store<NT, NT>(Yl, 0, acc);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[u][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    RT_WSTAMP(5);
    wide_kloop<NT, NT>(xop, Yl, 0, acc);       // A^3 = A A^2
    RT_WSTAMP(6);
    const int base = 3 * (q - 1);
    const double top0 = c_inv_fact[base], top1 = c_inv_fact[base + 1], top2 = c_inv_fact[base + 2],
                 top3 = c_inv_fact[mdeg];
    __syncthreads();
    wide_store<NT, NT>(Yl, 0, acc);
    __syncthreads();
    // T = B_(q-1) + c_m A^3 (the top block of the polynomial carries the A^3 term); with the
    // Horner steps in halves only this half's columns of T are ever a right operand
    constexpr bool A1LDS = SPLIT && NT % 2 == 0;   // (equal halves: the other one has the room)
    if (halves && A1LDS) {
        wide_read_aop<NT>(Yl, xop);            // A^3, this wave's rows: the left operand from here on
        __syncthreads();                       // every wave has its rows of A^3
        if (half == 0) wide_top<NT, JH, 0, true>(acc, n, SB, JH, 0, top0, top1, top2, top3, Yl, JH);
        else wide_top<NT, NT - JH, JH, true>(acc, n, SB, JH, 0, top0, top1, top2, top3, Yl, 0);
    } else {
        if (halves) {
            if (half == 0) wide_top<NT, JH, 0, false>(acc, n, SB, JH, 0, top0, top1, top2, top3);
            else wide_top<NT, NT - JH, JH, false>(acc, n, SB, JH, 0, top0, top1, top2, top3);
        } else {
            wide_top<NT, NT, 0, false>(acc, n, SB, NT, 0, top0, top1, top2, top3);
        }
        wide_read_aop<NT>(Yl, xop);
        __syncthreads();
        wide_store<NT, NT>(Yl, 0, acc);        // (in halves the other tiles keep A^3, which nobody reads)
    }
    __syncthreads();
    RT_WSTAMP(7);
    if (halves) {
        for (int jj = q - 2; jj >= 0; --jj) {
            const double c0 = c_inv_fact[3 * jj], c1 = c_inv_fact[3 * jj + 1], c2 = c_inv_fact[3 * jj + 2];
            if (half == 0) wide_horner<NT, JH, A1LDS>(xop, Yl, 0, n, SB, JH, 0, c0, c1, c2, JH);
            else wide_horner<NT, NT - JH, A1LDS>(xop, Yl, JH, n, SB, JH, 0, c0, c1, c2, 0);
        }
    } else {
        for (int jj = q - 2; jj >= 0; --jj)
            wide_horner<NT, NT>(xop, Yl, 0, n, SB, NT, 0, c_inv_fact[3 * jj],
                                c_inv_fact[3 * jj + 1], c_inv_fact[3 * jj + 2]);
        for (int r = 0; r < s; ++r) {          // T <- T T
            wide_read_aop<NT>(Yl, xop);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[u][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
            wide_kloop<NT, NT>(xop, Yl, 0, acc);
            __syncthreads();
            wide_store<NT, NT>(Yl, 0, acc);
            __syncthreads();
        }
    }
    RT_WSTAMP(8);
    // the result, this workgroup's columns: element (row, col) of the LDS image
    auto at = [&](int row, int col) {
        return Yl[(row >> 2) * RS + (col >> 4) * 64 + (row & 3) * 16 + (col & 15)];
    };
    const int c_lo = SPLIT ? 16 * jb : 0;
    const int c_hi = SPLIT ? (half ? RN : 16 * JH) : RN;
    for (int jc = c_lo + (tid & 63); jc < c_hi && jc < n; jc += 64)
        for (int i = tid >> 6; i < n; i += WV) {
            const double v = at(i, jc);
            Pb[i * n + jc] = v;
            if (step >= 0 && frag_kind == 0) Pfrag[(long)step * nn + i * n + jc] = v;
        }
    if (step >= 0 && frag_kind == 1) {
        // Pfrag[step][m][q][lane][e2] = P[16 m + (lane & 15)][4 (2 q + e2) + (lane >> 4)]: pair q
        // covers columns 8 q .. 8 q + 7
        const int KP = (KSn + 1) / 2;
        const int total = NTn * KP * 128;
        const int q_lo = c_lo / 8, q_hi = (c_hi / 8 < KP) ? c_hi / 8 : KP;
        const int e2 = tid & 1, ln = (tid >> 1) & 63;
        for (int mm = 0; mm < NTn; ++mm)
            for (int qq = q_lo + (tid >> 7); qq < q_hi; qq += TPBW >> 7) {
                const int e = (mm * KP + qq) * 128 + (tid & 127);
                const int row = 16 * mm + (ln & 15);
                const int col = 4 * (2 * qq + e2) + (ln >> 4);
                Pfrag[(long)step * total + e] = (col < RN) ? at(row, col) : 0.0;
            }
    }
    RT_WSTAMP(9);
}


}  // namespace

// grid = workgroups (two per matrix when split2) + the carried reduction's, if any
int rt_expm_wide_launch(rt_ctx *ctx, int nt, bool split2, size_t grid, int64_t n, const double *d_Q,
                        const int *d_qidx, const double *d_t, double *d_P, int *d_info,
                        const int *d_step_of_node, int frag_kind, double *d_Pfrag,
                        const rt_reduce_args &red)
{
    const size_t lds_w = (size_t)(4 * nt) * (nt * 64 + 4) * 8;
    hipEvent_t evw = nullptr;
    rt_time_begin(ctx, RT_K_EXPM, split2 ? "expm_taylor_ps_mfma_wide_split2"
                                          : "expm_taylor_ps_mfma_wide", &evw);
#define RT_WIDE(NTV, SP)                                                                        \
    do {                                                                                        \
        if (lds_w > ctx->expm_wide_attr_lds[(NTV - 5) * 2 + (SP ? 1 : 0)]) {                    \
            RT_HIP(hipFuncSetAttribute((const void *)expm_taylor_wide_kernel<NTV, SP>,          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize,              \
                                       (int)lds_w));                                            \
            ctx->expm_wide_attr_lds[(NTV - 5) * 2 + (SP ? 1 : 0)] = lds_w;                      \
        }                                                                                       \
        RT_LAUNCH_TIMED(ctx, (expm_taylor_wide_kernel<NTV, SP>), dim3((unsigned)grid),          \
                        dim3(64 * ((NTV + 1) / 2)), lds_w, (int)n, d_Q, d_qidx, d_t, d_P,       \
                        d_info, d_step_of_node, frag_kind, d_Pfrag, ctx->d_expm_scratch, red);  \
    } while (0)
    if (split2) {
        switch (nt) {
        case 5: RT_WIDE(5, true); break;
        case 6: RT_WIDE(6, true); break;
        case 7: RT_WIDE(7, true); break;
        default: RT_WIDE(8, true); break;
        }
    } else {
        switch (nt) {
        case 5: RT_WIDE(5, false); break;
        case 6: RT_WIDE(6, false); break;
        case 7: RT_WIDE(7, false); break;
        default: RT_WIDE(8, false); break;
        }
    }
#undef RT_WIDE
    RT_HIP(hipGetLastError());
    if (getenv("RAOTEH_EXPM_TRACE")) {
        static int armed = 0;
        unsigned long long tr[12];
        RT_HIP(hipStreamSynchronize(ctx->stream));
        if (armed) {
            RT_HIP(hipMemcpyFromSymbol(tr, HIP_SYMBOL(rt_expm_wide_trace), sizeof tr));
            fprintf(stderr, "[raoteh_amd] wide expm trace (clocks, workgroup 0): load + norm %llu, order + "
                    "scale %llu, A operands %llu, A^2 k-loop %llu, store %llu, A^3 k-loop %llu, top block "
                    "%llu, Horner + squarings %llu, output %llu; total %llu\n", tr[1] - tr[0],
                    tr[2] - tr[1], tr[3] - tr[2], tr[4] - tr[3], tr[5] - tr[4], tr[6] - tr[5],
                    tr[7] - tr[6], tr[8] - tr[7], tr[9] - tr[8], tr[9] - tr[0]);
        }
        const int on = 1;
        RT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rt_expm_wide_trace_on), &on, sizeof on));
        armed = 1;
    }
    rt_time_end(ctx, RT_K_EXPM, evw);
    return RT_OK;
}
