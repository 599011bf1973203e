// Expected history statistics on the matrix pipe (8 < n <= 64: the codon and compound models).
//
// rt_mjp_esd_expectation_weights_obs needs, per edge e = (p -> v) and summed over the sites,
//     W_e[a][b] = sum_s w_s u_s[a] L_v,s[b],   u = D_p / M_v,   D_v = (P_v^T u) * L_v,
// with L the subtree likelihoods, M_v = P_v L_v the message to the parent and D the posterior
// node marginals (raoteh/sampler/_mjp_dense.py:458-533 after the passes of _mcy_dense.py:57-230).
// The reference-format kernels of passes.hip do that one site per wave: 80 ms for 10 000 sites
// of the 61-state model, next to 0.24 ms for the likelihood of the same batch.  The work is two
// pruning-sized passes and one GEMM over the sites per edge, so it runs here as
//
//   up     the split-M interpreter pruning kernel (prune.hip) with its own rows of L_v and M_v
//          of every step written out, [step][tile][m][r][lane] (the D layout of
//          v_mfma_f64_16x16x4: register r of row tile m on lane l = state 16m + 4r + (l >> 4) of
//          site l & 15 of the 16-site tile);
//   down   the same tile-per-workgroup shape over the steps in reverse: u = D_p / M_v in
//          registers, exchanged through LDS as B operands, P_v^T as A fragments, D_v = (.) * L_v;
//          u is kept for the site sums; leaves skip the product (nobody reads their D);
//   wsum   per edge and chunk of site tiles: W += U^T-tile x L-tile with K = sites (4 sites per
//          MFMA), accumulators in registers over the chunk, one partial per chunk;
//   finish the chunks summed in order, masked by P != 0, written in the reference's
//          [node][a][b] order; the root's posterior sums.
//
// Everything is summed in a fixed order.  The boolean passes are not run: with the upward pass
// carrying exact zeros they change no number here (an infeasible site has likelihood 0 and
// contributes nothing, as before).
#include "common.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int EX_CHUNKS = 8;              // site-tile chunks per edge in the site sums

// A fragments of P^T in step order: T[step][m][q][lane][e2] = P_v[4(2q+e2) + (lane>>4)][16m + (lane&15)]
__global__ void __launch_bounds__(256)
pack_pt_kernel(int n, int NT, int KP, int nops, const int *__restrict__ step_node,
               const double *__restrict__ P, double *__restrict__ out)
{
    const long total = (long)nops * NT * KP * 128;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int e2 = (int)(e & 1);
        const int ln = (int)((e >> 1) & 63);
        long rest = e >> 7;
        const int q = (int)(rest % KP);
        rest /= KP;
        const int m = (int)(rest % NT);
        const int step = (int)(rest / NT);
        const int b = 16 * m + (ln & 15);                 // output state (row of P^T)
        const int a = 4 * (2 * q + e2) + (ln >> 4);       // input state (k)
        const int v = step_node[step];
        out[e] = (a < n && b < n) ? P[((long)v * n + a) * n + b] : 0.0;
    }
}

template <int NT, int KS>
__global__ void __launch_bounds__(64 * NT)
expect_down_kernel(const double *__restrict__ PfragT, int nops, const int *__restrict__ parent_step,
                   const unsigned char *__restrict__ internal, const double *__restrict__ Larr,
                   const double *__restrict__ Marr, double *__restrict__ Darr,
                   double *__restrict__ Uarr, const double *__restrict__ root_w, int n,
                   int *__restrict__ status, long nsites, long nblocks)
{
    constexpr int KP = (KS + 1) / 2;
    __shared__ double xb[NT * 4 * 64];
    __shared__ double red[NT][16];
    const int lane = threadIdx.x & 63;
    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long blk = blockIdx.x;
    const size_t tile_stride = (size_t)NT * 256;
    auto at = [&](int step) { return ((size_t)step * nblocks + blk) * tile_stride + (m * 4) * 64 + lane; };
    bool bad = false;
    // root: D = w L / sum_states(w L)  (_mc0_dense.py:400-489 with the prior weights)
    {
        const int i = nops - 1;
        const size_t o = at(i);
        double wl[4], s = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * m + 4 * r + (lane >> 4);
            const double w = row < n ? (root_w ? root_w[row] : 1.0) : 0.0;
            wl[r] = w * Larr[o + r * 64];
            s += wl[r];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) red[m][lane] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int mm = 0; mm < NT; ++mm) tot += red[mm][lane & 15];
        if (!(tot > 0.0)) bad = true;
#pragma unroll
        for (int r = 0; r < 4; ++r) Darr[o + r * 64] = tot > 0.0 ? wl[r] / tot : 0.0;
    }
    const double *ag = PfragT + ((size_t)m * KP * 64 + lane) * 2;
    constexpr size_t ASTRIDE = (size_t)NT * KP * 128;
    for (int i = nops - 2; i >= 0; --i) {
        const size_t o = at(i), po = at(parent_step[i]);
        double u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double dp = Darr[po + r * 64];
            const double den = Marr[o + r * 64];
            u[r] = 0.0;
            if (dp != 0.0) {
                if (den > 0.0) u[r] = dp / den;
                else bad = true;
            }
            Uarr[o + r * 64] = u[r];
        }
        if (!internal[i]) continue;              // a leaf: nobody reads its D (uniform)
        __syncthreads();                         // the previous step's operands are read
#pragma unroll
        for (int r = 0; r < 4; ++r) xb[(4 * m + r) * 64 + lane] = u[r];
        double a[2 * KP];
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            const double2 v = *(const double2 *)(ag + (size_t)i * ASTRIDE + q * 128);
            a[2 * q] = v.x;
            a[2 * q + 1] = v.y;
        }
        __syncthreads();
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], xb[kk * 64 + lane], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Darr[o + r * 64] = acc[r] * Larr[o + r * 64];
    }
    const long site = blk * 16 + (lane & 15);
    if (bad && site < nsites) atomicOr(&status[site], 2);
}

// The downward pass with the posterior marginals of the inner nodes in LDS.  The kernel above
// reads D of the parent back from HBM at every step and fetches M, L and the A fragments of a
// step when it gets there: 126 dependent round trips per tile (870 us at 10 000 codon sites for
// a pass whose products take 85).  Here
//   * D_v of an inner node lives in an LDS slot (own rows: 4 doubles per lane, read and written
//     by the same lane, so no barrier) -- the slot the upward pass kept v's accumulator in: read
//     backwards, the accumulator's lifetime (first child .. v's own step) is exactly D_v's
//     (v's step .. first child), so the upward schedule's slots never collide here either;
//   * M, L and the A fragments of the next step are requested before this step's work;
//   * the highest slot (the deepest inner nodes: the most frequent) is four registers, so that a
//     64-leaf balanced tree needs 5 slots + the exchange buffer = 48 KB and three workgroups
//     share a CU: 625 tiles are then one round on 256 CUs, not two;
//   * only U (the site sums read it) and the root's D leave the CU.
// meta[i] = {slot of the parent's D, own slot (-1: a leaf)}; regslot: the slot kept in registers.
// (three workgroups per CU is what the slot budget above is for: the registers must allow it too)
template <int NT, int KS>
__global__ void __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(3, 3)))
expect_down_lds_kernel(const double *__restrict__ PfragT, int nops, const int2 *__restrict__ meta,
                       const double *__restrict__ Larr, const double *__restrict__ Marr,
                       double *__restrict__ Darr, double *__restrict__ Uarr,
                       const double *__restrict__ root_w, int n, int *__restrict__ status, long nsites,
                       long nblocks, int regslot)
{
    constexpr int KP = (KS + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double down_sm[];
    double *xb = down_sm;                          // [NT * 4 * 64]
    double *slots = down_sm + NT * 256;            // [slot < regslot][NT][4][64]
    // the step table in LDS (read from global at the top of every step it was a dependent
    // round trip per step)
    int2 *smeta = (int2 *)(slots + (size_t)regslot * NT * 256);
    for (int i = threadIdx.x; i < nops; i += 64 * NT) smeta[i] = meta[i];
    double regD[4] = {0.0, 0.0, 0.0, 0.0};
    __shared__ double red[NT][16];
    const int lane = threadIdx.x & 63;
    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long blk = blockIdx.x;
    const size_t tile_stride = (size_t)NT * 256;
    auto at = [&](int step) { return ((size_t)step * nblocks + blk) * tile_stride + (m * 4) * 64 + lane; };
    auto slot = [&](int sidx) { return slots + ((size_t)(sidx < regslot ? sidx : 0) * NT + m) * 256 + lane; };
    bool bad = false;
    // root: D = w L / sum_states(w L)  (_mc0_dense.py:400-489 with the prior weights)
    {
        const int i = nops - 1;
        const size_t o = at(i);
        double wl[4], s = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * m + 4 * r + (lane >> 4);
            const double w = row < n ? (root_w ? root_w[row] : 1.0) : 0.0;
            wl[r] = w * Larr[o + r * 64];
            s += wl[r];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) red[m][lane] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int mm = 0; mm < NT; ++mm) tot += red[mm][lane & 15];
        if (!(tot > 0.0)) bad = true;
        const int own = smeta[i].y;              // (after the barrier above: the table is there)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double d = tot > 0.0 ? wl[r] / tot : 0.0;
            Darr[o + r * 64] = d;
            if (own == regslot) regD[r] = d;
            else if (own >= 0) slot(own)[r * 64] = d;
        }
    }
    const double *ag = PfragT + ((size_t)m * KP * 64 + lane) * 2;
    constexpr size_t ASTRIDE = (size_t)NT * KP * 128;
    // What step i needs from memory: M (every step) and L (inner nodes) from HBM, requested
    // two steps ahead, the A fragments (L2) one step ahead.  Named buffers, the loop unrolled,
    // running pointers: nothing in flight is ever copied (a register copy of a
    // pending load waits for it) and no address is computed into a borrowed register.
    const size_t step_stride = (size_t)nblocks * tile_stride;
    const double *pM = Marr + at(nops - 2), *pL = Larr + at(nops - 2);     // the fetch stream
    double *pU = Uarr + at(nops - 2);                                      // the current step
    const double *pA = ag + (size_t)(nops - 2) * ASTRIDE;
    int fi = nops - 2;                           // next step to fetch M / L of
    auto fetch_ml = [&](double (&M4)[4], double (&L4)[4]) {
        if (fi >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) M4[r] = pM[r * 64];
            if (smeta[fi].y >= 0) {              // (uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r) L4[r] = pL[r * 64];
            }
        }
        pM -= step_stride;
        pL -= step_stride;
        fi -= 1;
    };
    auto fetch_a = [&](int i, double (&A)[2 * KP]) {
        if (i >= 0 && smeta[i].y >= 0) {
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                const double2 v = *(const double2 *)(pA + q * 128);
                A[2 * q] = v.x;
                A[2 * q + 1] = v.y;
            }
        }
        pA -= ASTRIDE;
    };
    // one step: u = D_parent / M, U out, and for an inner node D = (P^T u) * L into its slot
    auto step = [&](int i, const double (&Mc)[4], const double (&Lc)[4], const double (&Ac)[2 * KP]) {
        const int2 mt = smeta[i];
        const double *ps = slot(mt.x);
        double u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double dp = mt.x == regslot ? regD[r] : ps[r * 64];
            u[r] = 0.0;
            if (dp != 0.0) {
                if (Mc[r] > 0.0) u[r] = dp / Mc[r];
                else bad = true;
            }
            pU[r * 64] = u[r];
        }
        pU -= step_stride;
        if (mt.y >= 0) {
            __syncthreads();                     // the previous product's operands are read
#pragma unroll
            for (int r = 0; r < 4; ++r) xb[(4 * m + r) * 64 + lane] = u[r];
            __syncthreads();
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ac[kk], xb[kk * 64 + lane], acc, 0, 0, 0);
            if (mt.y == regslot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) regD[r] = acc[r] * Lc[r];
            } else {
                double *os = slot(mt.y);
#pragma unroll
                for (int r = 0; r < 4; ++r) os[r * 64] = acc[r] * Lc[r];
            }
        }
    };
    // (three M / L buffers = two steps ahead, two A buffers, unrolled by six: with four buffers
    // the kernel needs 188 registers -- two workgroups per CU, 625 tiles in two rounds)
    double M0[4], L0[4], M1[4], L1[4], M2[4], L2[4], A0[2 * KP], A1[2 * KP];
    fetch_ml(M0, L0);
    fetch_ml(M1, L1);
    fetch_a(nops - 2, A0);
    for (int i = nops - 2; i >= 0; i -= 6) {
        fetch_ml(M2, L2);
        fetch_a(i - 1, A1);
        step(i, M0, L0, A0);
        if (i - 1 < 0) break;
        fetch_ml(M0, L0);
        fetch_a(i - 2, A0);
        step(i - 1, M1, L1, A1);
        if (i - 2 < 0) break;
        fetch_ml(M1, L1);
        fetch_a(i - 3, A1);
        step(i - 2, M2, L2, A0);
        if (i - 3 < 0) break;
        fetch_ml(M2, L2);
        fetch_a(i - 4, A0);
        step(i - 3, M0, L0, A1);
        if (i - 4 < 0) break;
        fetch_ml(M0, L0);
        fetch_a(i - 5, A1);
        step(i - 4, M1, L1, A0);
        if (i - 5 < 0) break;
        fetch_ml(M1, L1);
        fetch_a(i - 6, A0);
        step(i - 5, M2, L2, A1);
    }
    const long site = blk * 16 + (lane & 15);
    if (bad && site < nsites) atomicOr(&status[site], 2);
}

// W partial of one edge (= step) and one chunk of site tiles.  K = the 16 sites of a tile, four
// per MFMA: k-lane hi of k-step ks is site 4 hi + ks, so that the four k-steps' operands of a lane
// are 32 contiguous bytes of the D-layout arrays (state 16 M + x, site t at (M * 4 + x / 4) * 64 +
// 16 (x % 4) + t): one double4 per array and row tile.
//
// The waves of a workgroup take the chunk's tiles in turn, and each computes ALL NT x NT tile
// pairs of W for its tiles: a tile's U and L are then read once per workgroup, not once per wave
// (L was), a tile is 4 NT^2 MFMAs long, and a wave's next tile is requested two tiles ahead with
// nothing else to wait for.  (History: four scattered doubles per operand and load -> wait -> MFMA
// 780 us at 10 000 codon sites; contiguous operands 369; a wave per row tile with deeper prefetch
// no better -- 2 waves per SIMD at 184 registers, 38 % of the pipe, 44 % of HBM, the waves waiting
// 91 % of their cycles.)  The partials of the waves are added through LDS in wave order.
//
// Fetching: three named buffers, the loop unrolled by three, running pointers, 32-bit counters.
// Rotating buffers through register copies makes every copy wait for the newest loads; an
// address computed into a register a pending load still owns, or a 64-bit loop test that borrows
// one, drains the queue; a conditional load makes the compiler drain it at the back edge.  A
// fetch past the chunk's last tile reads the next tiles or steps of the same scratch block -- U
// is followed by the partials, L by M -- and is never used.
// L of an observed leaf is not in Larr (the upward pass does not store it): it is the leaf's
// observation vector in the batch's resident image, obs[tile][k][pair q][lane][2] = the B-operand
// pairs of states 8 q .. 8 q + 7 (leaf_k[step] = the leaf's stream position k, or -1).  State
// 16 mb + lo of sites 4 hi .. 4 hi + 3 are four doubles at stride 2 there: eight consecutive
// doubles are loaded and the ones of this state's k-step parity picked.
template <int NT, bool WEIGHTS>
__global__ void __launch_bounds__(64 * NT)
expect_wsum_kernel(int nops, const double *__restrict__ Uarr, const double *__restrict__ Larr,
                   const double *__restrict__ weights, long nsites, long nblocks,
                   double *__restrict__ partial, const int *__restrict__ leaf_k,
                   const double *__restrict__ obs, int K, int KP)
{
    __shared__ double redw[NT * NT * 256];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int step = blockIdx.x, chunk = blockIdx.y;
    const int nb = (int)nblocks;
    const int per = (nb + EX_CHUNKS - 1) / EX_CHUNKS;
    const int t0 = chunk * per, t1 = t0 + per < nb ? t0 + per : nb;
    const size_t tile_stride = (size_t)NT * 256;
    const int lo = lane & 15, hi = lane >> 4;
    const int row_off = (lo >> 2) * 64 + 16 * (lo & 3) + 4 * hi;
    double4_t acc[NT][NT];
#pragma unroll
    for (int ma = 0; ma < NT; ++ma)
#pragma unroll
        for (int mb = 0; mb < NT; ++mb) acc[ma][mb] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int first = t0 + wv;                   // this wave's tiles: first, first + NT, ...
    if (first < t1) {
        const size_t base = ((size_t)step * nblocks + first) * tile_stride + row_off;
        const double *pu = Uarr + base, *pl = Larr + base;
        int ftile = first;
        auto fetch = [&](double4_t (&u)[NT], double4_t (&l)[NT], double4_t &w) {
#pragma unroll
            for (int mm = 0; mm < NT; ++mm) {
                u[mm] = *(const double4_t *)(pu + mm * 256);
                l[mm] = *(const double4_t *)(pl + mm * 256);
            }
            const long site = (long)ftile * 16 + 4 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long sc = site + ks < nsites ? site + ks : nsites - 1;
                const double wgt = WEIGHTS ? weights[sc] : 1.0;
                w[ks] = site + ks < nsites ? wgt : 0.0;
            }
            pu += NT * tile_stride;
            pl += NT * tile_stride;
            ftile += NT;
        };
        auto compute = [&](const double4_t (&u)[NT], const double4_t (&l)[NT], const double4_t &w) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ma = 0; ma < NT; ++ma) {
                    const double av = u[ma][ks] * w[ks];           // U[16 ma + lo][site] w
#pragma unroll
                    for (int mb = 0; mb < NT; ++mb)                  // L[16 mb + lo][site]
                        acc[ma][mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, l[mb][ks], acc[ma][mb],
                                                                           0, 0, 0);
                }
        };
        // ---- the same for an observed leaf: L from the observation image
        const int lk = leaf_k[step];                 // (uniform)
        const int par = (lo >> 2) & 1;               // which double of a pair this state is
        const double *po = obs + ((size_t)first * K + (lk >= 0 ? lk : 0)) * KP * 128 +
                           (size_t)(lo >> 3) * 128 + ((lo & 3) * 16 + 4 * hi) * 2;
        const size_t obs_stride = (size_t)NT * K * KP * 128;
        auto fetch_leaf = [&](double4_t (&u)[NT], double4_t (&l0)[NT], double4_t (&l1)[NT], double4_t &w) {
#pragma unroll
            for (int mm = 0; mm < NT; ++mm) {
                u[mm] = *(const double4_t *)(pu + mm * 256);
                // pair q = 2 mm + (lo >> 3); beyond the last pair: padding (zeros, selected below)
                const bool ok = 2 * mm + (lo >> 3) < KP;
                const double *p = po + (ok ? 2 * mm * 128 : 0);
                l0[mm] = *(const double4_t *)p;
                l1[mm] = *(const double4_t *)(p + 4);
            }
            const long site = (long)ftile * 16 + 4 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const long sc = site + ks < nsites ? site + ks : nsites - 1;
                const double wgt = WEIGHTS ? weights[sc] : 1.0;
                w[ks] = site + ks < nsites ? wgt : 0.0;
            }
            pu += NT * tile_stride;
            if (ftile + NT < nb) po += obs_stride;     // (the image ends with the last tile)
            ftile += NT;
        };
        auto compute_leaf = [&](const double4_t (&u)[NT], const double4_t (&l0)[NT],
                                const double4_t (&l1)[NT], const double4_t &w) {
            // the eight doubles are sites 4 hi .. 4 hi + 3 as pairs (even, odd k-step of the
            // pair): site ks of this state = element 2 ks + par
            double lv[NT][4];
#pragma unroll
            for (int mb = 0; mb < NT; ++mb) {
                const bool ok = 2 * mb + (lo >> 3) < KP;
                lv[mb][0] = ok ? (par ? l0[mb][1] : l0[mb][0]) : 0.0;
                lv[mb][1] = ok ? (par ? l0[mb][3] : l0[mb][2]) : 0.0;
                lv[mb][2] = ok ? (par ? l1[mb][1] : l1[mb][0]) : 0.0;
                lv[mb][3] = ok ? (par ? l1[mb][3] : l1[mb][2]) : 0.0;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ma = 0; ma < NT; ++ma) {
                    const double av = u[ma][ks] * w[ks];
#pragma unroll
                    for (int mb = 0; mb < NT; ++mb)
                        acc[ma][mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lv[mb][ks], acc[ma][mb],
                                                                           0, 0, 0);
                }
        };
#define RT_WSUM_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
        if (lk < 0) {
            double4_t uA[NT], lA[NT], wA, uB[NT], lB[NT], wB, uC[NT], lC[NT], wC;
            fetch(uA, lA, wA);
            fetch(uB, lB, wB);
            RT_WSUM_FENCE();
            for (int tile = first; tile < t1; tile += 3 * NT) {
                fetch(uC, lC, wC);
                RT_WSUM_FENCE();
                compute(uA, lA, wA);
                RT_WSUM_FENCE();
                if (tile + NT < t1) {
                    fetch(uA, lA, wA);
                    RT_WSUM_FENCE();
                    compute(uB, lB, wB);
                    RT_WSUM_FENCE();
                }
                if (tile + 2 * NT < t1) {
                    fetch(uB, lB, wB);
                    RT_WSUM_FENCE();
                    compute(uC, lC, wC);
                    RT_WSUM_FENCE();
                }
            }
        } else {
            double4_t uA[NT], pA[NT], qA[NT], wA, uB[NT], pB[NT], qB[NT], wB, uC[NT], pC[NT], qC[NT], wC;
            fetch_leaf(uA, pA, qA, wA);
            fetch_leaf(uB, pB, qB, wB);
            RT_WSUM_FENCE();
            for (int tile = first; tile < t1; tile += 3 * NT) {
                fetch_leaf(uC, pC, qC, wC);
                RT_WSUM_FENCE();
                compute_leaf(uA, pA, qA, wA);
                RT_WSUM_FENCE();
                if (tile + NT < t1) {
                    fetch_leaf(uA, pA, qA, wA);
                    RT_WSUM_FENCE();
                    compute_leaf(uB, pB, qB, wB);
                    RT_WSUM_FENCE();
                }
                if (tile + 2 * NT < t1) {
                    fetch_leaf(uB, pB, qB, wB);
                    RT_WSUM_FENCE();
                    compute_leaf(uC, pC, qC, wC);
                    RT_WSUM_FENCE();
                }
            }
        }
#undef RT_WSUM_FENCE
    }
    // wave 0 += wave 1, 2, ... in that order
    for (int w = 1; w < NT; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int ma = 0; ma < NT; ++ma)
#pragma unroll
                for (int mb = 0; mb < NT; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        redw[((ma * NT + mb) * 4 + r) * 64 + lane] = acc[ma][mb][r];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int ma = 0; ma < NT; ++ma)
#pragma unroll
                for (int mb = 0; mb < NT; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[ma][mb][r] += redw[((ma * NT + mb) * 4 + r) * 64 + lane];
        }
    }
    if (wv == 0) {
        double *out = partial + ((size_t)step * EX_CHUNKS + chunk) * NT * NT * 256;
#pragma unroll
        for (int ma = 0; ma < NT; ++ma)
#pragma unroll
            for (int mb = 0; mb < NT; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[((ma * NT + mb) * 4 + r) * 64 + lane] = acc[ma][mb][r];
    }
}

// chunks summed in order, structural zeros of P masked, reference order [node][a][b]
__global__ void __launch_bounds__(256)
expect_finish_kernel(int n, int NT, int nops, const int *__restrict__ step_node,
                     const double *__restrict__ esd, const double *__restrict__ partial,
                     double *__restrict__ W)
{
    const int step = blockIdx.x;                 // edges: steps 0 .. nops - 2
    const int v = step_node[step];
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int a = e / n, b = e - a * n;
        const int ma = a >> 4, r = (a & 15) >> 2, hi = a & 3;
        const int mb = b >> 4, lo = b & 15;
        double sum = 0.0;
        for (int c = 0; c < EX_CHUNKS; ++c)
            sum += partial[((((size_t)step * EX_CHUNKS + c) * NT + ma) * NT + mb) * 256 + r * 64 +
                           hi * 16 + lo];
        W[((size_t)v * n + a) * n + b] = esd[((size_t)v * n + a) * n + b] != 0.0 ? sum : 0.0;
    }
}

// slot 0, column 0: the weighted sum of the root posteriors -- per chunk of tiles here, the
// chunks are added in order by expect_root_finish_kernel
constexpr int EX_ROOT_CHUNKS = 64;

__global__ void __launch_bounds__(64)
expect_root_kernel(int n, int NT, int root_step, const double *__restrict__ Darr,
                   const double *__restrict__ weights, long nsites, long nblocks,
                   double *__restrict__ rootpart)
{
    const int a = threadIdx.x;                   // state (n <= 64)
    const long per = (nblocks + EX_ROOT_CHUNKS - 1) / EX_ROOT_CHUNKS;
    const long t0 = blockIdx.x * per, t1 = t0 + per < nblocks ? t0 + per : nblocks;
    double sum = 0.0;
    if (a < n) {
        const int ma = a >> 4, r = (a & 15) >> 2, hi = a & 3;
        for (long tile = t0; tile < t1; ++tile) {
            const double *Dt = Darr + ((size_t)root_step * nblocks + tile) * ((size_t)NT * 256) +
                               (ma * 4 + r) * 64 + hi * 16;
            for (int t = 0; t < 16; ++t) {
                const long site = tile * 16 + t;
                if (site < nsites) sum += (weights ? weights[site] : 1.0) * Dt[t];
            }
        }
    }
    rootpart[blockIdx.x * 64 + a] = sum;
}

__global__ void __launch_bounds__(256)
expect_root_finish_kernel(int n, const double *__restrict__ rootpart, double *__restrict__ W)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int a = e / n, b = e - a * n;
    double sum = 0.0;
    if (b == 0)
        for (int c = 0; c < EX_ROOT_CHUNKS; ++c) sum += rootpart[c * 64 + a];
    W[e] = sum;
}

struct dev_free {
    std::vector<void *> ptrs;
    ~dev_free() { for (void *p : ptrs) hipFree(p); }
    template <class T> int alloc(T *&p, size_t count)
    {
        p = nullptr;
        RT_HIP(hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)));
        ptrs.push_back(p);
        return RT_OK;
    }
};

// An EM loop calls with the same tree and the same observations again and again, only the
// transition matrices change: the model (schedule, device tree) and the packed batch of the
// last call stay with the context (one entry; 2.5 of a call's 6 ms at 10 000 codon sites were
// creating and destroying them).  Keys: sizes + FNV-1a of the tree arrays / of the observations.
struct expect_cache_t {
    uint64_t tree_key = 0, data_key = 0;
    int64_t nnodes = 0, n = 0, nsites = 0, nobs = 0, chunk_lo = -1;
    int kind = -1;
    rt_model *model = nullptr;
    rt_sites *sites = nullptr;
    ~expect_cache_t()
    {
        if (sites) rt_sites_destroy(sites);
        if (model) rt_model_destroy(model);
    }
};

uint64_t fnv1a(const void *p, size_t bytes, uint64_t h = 1469598103934665603ull)
{
    // eight bytes a round (the observations of a batch are megabytes), the tail bytewise
    const unsigned char *b = (const unsigned char *)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 1099511628211ull;
        h ^= h >> 29;
    }
    for (; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// Everything of one pass that happens on the device, asynchronously on the context's stream:
// upward pass of the split-M interpreter kernel with L and M of every step stored, downward
// pass, per-edge site sums, W in the reference's [node][a][b] order (slot 0, column 0: the
// weighted sum of the root posteriors).  `s`: a split-M interpreter batch (MFMA layout, one
// row tile per wave, no tree-specialised kernel); d_w device weights or null.
template <int NT, int KS>
int expect_device_passes(rt_ctx *ctx, rt_model *model, rt_sites *s, const double *d_PT,
                         const int *d_step_node, const int *d_parent_step,
                         const unsigned char *d_internal, const double *esd_dev,
                         const double *d_root_w, const double *d_w, double *d_W, int *d_status,
                         double *d_rootpart, bool trace)
{
    hipStream_t st = ctx->stream;
    RT_REQUIRE(s->layout == RT_LAYOUT_MFMA && !s->mfma_solo && !s->jit_fn,
               "unexpected batch layout for the matrix-pipe expectation path");
    const int64_t n = model->n, nsites = s->nsites;
    const int nops = (int)s->ops.size();
    const long nblocks = (long)s->nblocks;
    const size_t arr = (size_t)nops * nblocks * NT * 256;
    // the four per-step arrays and the chunk partials live in the context's grow-only scratch
    // (2.6 GB at 10 000 sites of the codon model: a hipMalloc / hipFree pair per call costs more
    // than the kernels)
    const size_t npart = (size_t)nops * EX_CHUNKS * NT * NT * 256;
    RT_TRY(rt_scratch_reserve(ctx, (4 * arr + npart) * 8));
    double *d_L = (double *)ctx->d_scratch;
    double *d_M = d_L + arr, *d_D = d_M + arr, *d_U = d_D + arr, *d_part = d_U + arr;
    // M of the root step is never written by the upward pass (no product there)
    s->d_Lout = d_L;
    s->d_Mout = d_M;
    const int rc = rt_launch_prune(model, s, false);
    s->d_Lout = s->d_Mout = nullptr;
    RT_TRY(rc);
    if (trace) hipStreamSynchronize(st);
    RT_HIP(hipMemsetAsync(d_status, 0, (size_t)nsites * 4, st));
    // the LDS form of the downward pass when the slots of the tree fit (RAOTEH_EXPECT_DOWN=global:
    // the first form, D through HBM)
    if (!s->d_down_meta) {
        // [nops] {parent slot, own slot} then [nops] the stream position of an observed leaf (-1)
        std::vector<int32_t> meta((size_t)nops * 3);
        int nslots = 0;
        for (int i = 0; i < nops; ++i) {
            const rt_op &op = s->ops[(size_t)i];
            meta[(size_t)i * 2] = op.dst >= 0 ? (op.dst & 255) : 0;
            meta[(size_t)i * 2 + 1] = op.pop;
            meta[(size_t)nops * 2 + i] = (op.pop < 0 && op.obs >= 0) ? op.obs : -1;
            if (op.dst >= 0) nslots = std::max(nslots, (op.dst & 255) + 1);
            if (op.pop >= 0) nslots = std::max(nslots, op.pop + 1);
        }
        RT_HIP(hipMalloc((void **)&s->d_down_meta, meta.size() * 4));
        RT_HIP(hipMemcpyAsync(s->d_down_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, st));
        RT_HIP(hipStreamSynchronize(st));            // (meta is a local)
        s->down_slots = std::max(nslots, 1);
    }
    const int regslot = s->down_slots - 1;           // the deepest slot: registers
    const size_t down_lds = (size_t)(1 + regslot) * NT * 256 * 8 + (size_t)nops * 8;
    const char *dv = getenv("RAOTEH_EXPECT_DOWN");
    if (down_lds <= 120 * 1024 && !(dv && strcmp(dv, "global") == 0)) {
        auto kern = expect_down_lds_kernel<NT, KS>;
        RT_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)down_lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(64 * NT), down_lds, st, d_PT, nops,
                           (const int2 *)s->d_down_meta, d_L, d_M, d_D, d_U, d_root_w, (int)n, d_status,
                           (long)nsites, nblocks, regslot);
    } else {
        hipLaunchKernelGGL((expect_down_kernel<NT, KS>), dim3((unsigned)nblocks), dim3(64 * NT), 0, st,
                           d_PT, nops, d_parent_step, d_internal, d_L, d_M, d_D, d_U, d_root_w, (int)n,
                           d_status, (long)nsites, nblocks);
    }
    const int *d_leaf_k = s->d_down_meta + (size_t)nops * 2;
    const int KPo = (KS + 1) / 2;
    if (d_w)
        hipLaunchKernelGGL((expect_wsum_kernel<NT, true>), dim3((unsigned)(nops - 1), EX_CHUNKS),
                           dim3(64 * NT), 0, st, nops, d_U, d_L, d_w, (long)nsites, nblocks, d_part,
                           d_leaf_k, (const double *)s->d_obs, (int)s->nobs, KPo);
    else
        hipLaunchKernelGGL((expect_wsum_kernel<NT, false>), dim3((unsigned)(nops - 1), EX_CHUNKS),
                           dim3(64 * NT), 0, st, nops, d_U, d_L, d_w, (long)nsites, nblocks, d_part,
                           d_leaf_k, (const double *)s->d_obs, (int)s->nobs, KPo);
    hipLaunchKernelGGL(expect_finish_kernel, dim3((unsigned)(nops - 1)), dim3(256), 0, st, (int)n, NT,
                       nops, d_step_node, esd_dev, d_part, d_W);
    hipLaunchKernelGGL(expect_root_kernel, dim3(EX_ROOT_CHUNKS), dim3(64), 0, st, (int)n, NT, nops - 1,
                       d_D, d_w, (long)nsites, nblocks, d_rootpart);
    hipLaunchKernelGGL(expect_root_finish_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st,
                       (int)n, d_rootpart, d_W);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

template <int NT, int KS>
int run_chunk(rt_ctx *ctx, rt_model *model, int64_t n, int64_t nsites, int64_t nobs,
              const int64_t *obs_nodes, int kind, const void *data, const double *esd_dev,
              const double *d_root_w, const double *site_weights, const std::vector<int> &step_node,
              const std::vector<int> &parent_step, const std::vector<unsigned char> &internal,
              double *d_W, int *d_status, double *d_PT, const int *d_step_node,
              const int *d_parent_step, const unsigned char *d_internal, rt_sites *cached_sites,
              rt_sites **keep_sites)
{
    (void)step_node; (void)parent_step; (void)internal;
    hipStream_t st = ctx->stream;
    // RAOTEH_EXPECT_TRACE=1: host wall clock of the stages (each ends with a stream sync)
    const bool trace = getenv("RAOTEH_EXPECT_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    rt_sites *s = cached_sites;
    struct guard { rt_sites *s; ~guard() { if (s) rt_sites_destroy(s); } } g{nullptr};
    if (!s) {
        RT_TRY(rt_sites_create_interpreter(model, nsites, kind, nobs, obs_nodes, data, &s));
        if (keep_sites) *keep_sites = s;      // the caller's cache owns it now
        else g.s = s;
    }
    if (trace) hipStreamSynchronize(st);
    const auto t1 = now();
    dev_free mem;
    double *d_rootpart, *d_w = nullptr;
    RT_TRY(mem.alloc(d_rootpart, (size_t)EX_ROOT_CHUNKS * 64));
    if (site_weights) {
        RT_TRY(mem.alloc(d_w, (size_t)nsites));
        RT_HIP(hipMemcpyAsync(d_w, site_weights, (size_t)nsites * 8, hipMemcpyHostToDevice, st));
    }
    RT_TRY((expect_device_passes<NT, KS>(ctx, model, s, d_PT, d_step_node, d_parent_step, d_internal,
                                         esd_dev, d_root_w, d_w, d_W, d_status, d_rootpart, trace)));
    const auto t2 = now();
    RT_HIP(hipStreamSynchronize(st));
    if (trace)
        fprintf(stderr, "[raoteh_amd] expectation pass of %lld sites: batch + packing %.2f ms, scratch "
                "+ upward pass %.2f ms, downward pass + site sums %.2f ms\n", (long long)nsites,
                ms(t0, t1), ms(t1, t2), ms(t2, now()));
    return RT_OK;
}

}  // namespace

int rt_expectation_weights_mfma(rt_ctx *ctx, int64_t nnodes, int64_t n, int64_t nsites,
                                const int64_t *idx, const int64_t *ptr, const double *esd,
                                const double *root_distn, int64_t nobs, const int64_t *obs_nodes,
                                int kind, const void *data, const double *site_weights,
                                double *edge_weights, int32_t *status)
{
    if (n <= 8 || n > 64 || nnodes < 2 || getenv("RAOTEH_EXPECT_LEGACY")) return RT_ERR_UNSUPPORTED;
    const auto call_start = std::chrono::steady_clock::now();
    const size_t item = kind == RT_OBS_STATE ? 1 : 8;
    uint64_t tree_key = fnv1a(ptr, (size_t)(nnodes + 1) * 8);
    tree_key = fnv1a(idx, (size_t)(nnodes - 1) * 8, tree_key);
    tree_key = fnv1a(obs_nodes, (size_t)nobs * 8, tree_key);
    expect_cache_t *cache = (expect_cache_t *)ctx->expect_cache;
    if (cache && !(cache->tree_key == tree_key && cache->nnodes == nnodes && cache->n == n &&
                   cache->nobs == nobs && cache->kind == kind)) {
        delete cache;
        cache = nullptr;
        ctx->expect_cache = nullptr;
    }
    if (!cache) {
        cache = new (std::nothrow) expect_cache_t();
        if (!cache) return RT_ERR_NOMEM;
        ctx->expect_cache = cache;
        cache->tree_key = tree_key;
        cache->nnodes = nnodes;
        cache->n = n;
        cache->nobs = nobs;
        cache->kind = kind;
        const int rc = rt_model_create(ctx, nnodes, n, idx, ptr, &cache->model);
        if (rc != RT_OK) {
            rt_expect_cache_release(ctx);
            return rc;
        }
    }
    rt_model *model = cache->model;
    if (model->max_depth > RT_FAST_MAX_DEPTH) return RT_ERR_UNSUPPORTED;     // generic kernel only
    RT_TRY(rt_model_set_transitions(model, esd));
    std::vector<double> ones;
    if (!root_distn) {                       // weights of one (_mjp_dense.py:389-393)
        ones.assign((size_t)n, 1.0);
        root_distn = ones.data();
    }
    RT_TRY(rt_model_set_root_distn(model, root_distn));
    // the schedule: step -> node, its parent's step, whether it has children
    const int nops = (int)model->ops.size();
    std::vector<int> parent((size_t)nnodes, -1), step_of((size_t)nnodes, -1);
    for (int64_t v = 0; v < nnodes; ++v)
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) parent[(size_t)idx[e]] = (int)v;
    std::vector<int> step_node((size_t)nops), parent_step((size_t)nops, 0);
    std::vector<unsigned char> internal((size_t)nops, 0);
    for (int i = 0; i < nops; ++i) {
        step_node[(size_t)i] = model->ops[(size_t)i].node;
        step_of[(size_t)model->ops[(size_t)i].node] = i;
        internal[(size_t)i] = model->ops[(size_t)i].pop >= 0;
    }
    RT_REQUIRE(nops == nnodes && model->ops[(size_t)nops - 1].dst < 0, "unexpected schedule");
    for (int i = 0; i + 1 < nops; ++i) parent_step[(size_t)i] = step_of[(size_t)parent[(size_t)step_node[(size_t)i]]];
    const int NT = (int)((n + 15) / 16), KS = (int)((n + 3) / 4), KP = (KS + 1) / 2;
    hipStream_t st = ctx->stream;
    dev_free mem;
    double *d_W, *d_PT;
    int *d_status, *d_step_node, *d_parent_step;
    unsigned char *d_internal;
    const int64_t CS = 32768;                 // sites per pass: 4 arrays of nops x tiles x 8 KB
    RT_TRY(mem.alloc(d_W, (size_t)nnodes * n * n));
    RT_TRY(mem.alloc(d_PT, (size_t)nops * NT * KP * 128));
    RT_TRY(mem.alloc(d_status, (size_t)std::min<int64_t>(nsites, CS)));
    RT_TRY(mem.alloc(d_step_node, (size_t)nops));
    RT_TRY(mem.alloc(d_parent_step, (size_t)nops));
    RT_TRY(mem.alloc(d_internal, (size_t)nops));
    RT_HIP(hipMemcpyAsync(d_step_node, step_node.data(), (size_t)nops * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_parent_step, parent_step.data(), (size_t)nops * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(d_internal, internal.data(), (size_t)nops, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pack_pt_kernel, dim3(512), dim3(256), 0, st, (int)n, NT, KP, nops, d_step_node,
                       model->d_P, d_PT);
    RT_HIP(hipGetLastError());
    const size_t wcount = (size_t)nnodes * n * n;
    std::vector<double> acc(wcount, 0.0), part(wcount);
    // the packed batch is kept when the whole call is one pass (its observations hashed)
    const bool one_pass = nsites <= CS;
    uint64_t data_key = 0;
    if (one_pass) {
        data_key = fnv1a(data, (size_t)nsites * (size_t)nobs * item);
        if (cache->sites && !(cache->data_key == data_key && cache->nsites == nsites)) {
            rt_sites_destroy(cache->sites);
            cache->sites = nullptr;
        }
    } else if (cache->sites) {
        rt_sites_destroy(cache->sites);
        cache->sites = nullptr;
    }
    for (int64_t lo = 0; lo < nsites; lo += CS) {
        const int64_t cnt = std::min<int64_t>(CS, nsites - lo);
        const void *chunk_data = (const unsigned char *)data + (size_t)lo * (size_t)nobs * item;
        const double *chunk_w = site_weights ? site_weights + lo : nullptr;
        rt_sites *cached = one_pass ? cache->sites : nullptr;
        rt_sites *made = nullptr;
        rt_sites **keep = one_pass && !cached ? &made : nullptr;
        int rc = RT_ERR_UNSUPPORTED;
#define RT_EX(NTV, KSV)                                                                           \
        rc = run_chunk<NTV, KSV>(ctx, model, n, cnt, nobs, obs_nodes, kind, chunk_data, model->d_P, \
                                 model->d_root, chunk_w, step_node, parent_step, internal, d_W,     \
                                 d_status, d_PT, d_step_node, d_parent_step, d_internal, cached, keep)
        switch (KS) {
        case 3: RT_EX(1, 3); break;
        case 4: RT_EX(1, 4); break;
        case 5: RT_EX(2, 5); break;
        case 6: RT_EX(2, 6); break;
        case 7: RT_EX(2, 7); break;
        case 8: RT_EX(2, 8); break;
        case 9: RT_EX(3, 9); break;
        case 10: RT_EX(3, 10); break;
        case 11: RT_EX(3, 11); break;
        case 12: RT_EX(3, 12); break;
        case 13: RT_EX(4, 13); break;
        case 14: RT_EX(4, 14); break;
        case 15: RT_EX(4, 15); break;
        default: RT_EX(4, 16); break;
        }
#undef RT_EX
        if (made) {
            cache->sites = made;
            cache->data_key = data_key;
            cache->nsites = nsites;
        }
        RT_TRY(rc);
        RT_HIP(hipMemcpy(part.data(), d_W, wcount * 8, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < wcount; ++e) acc[e] += part[e];
        if (status) RT_HIP(hipMemcpy(status + lo, d_status, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    }
    memcpy(edge_weights, acc.data(), wcount * 8);
    if (getenv("RAOTEH_EXPECT_TRACE"))
        fprintf(stderr, "[raoteh_amd] expectation call: %.2f ms in all\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() -
                                                          call_start).count());
    return RT_OK;
}

// ---- resident expectation step -----------------------------------------------------------------
// _mjp_dense.get_expected_history_statistics (_mjp_dense.py:410-539) summed over a RESIDENT
// batch, in one call that moves 2 n + n^2 numbers: nothing is marshalled or uploaded per call
// (the reference-shaped entry points above re-upload tree, matrices and observations every
// time: 6.4 ms per call for 2.8 ms of kernels at 10 000 codon sites).  Per model, lazily: the
// schedule arrays of the downward pass, P^T fragments, W, the Frechet block buffers.
namespace {

struct expect_state_t {
    int nops = 0;
    double *d_PT = nullptr, *d_W = nullptr, *d_B = nullptr, *d_E = nullptr, *d_scale = nullptr;
    double *d_ones = nullptr, *d_out = nullptr, *d_rootpart = nullptr;
    int *d_step_node = nullptr, *d_parent_step = nullptr, *d_ident = nullptr, *d_status = nullptr;
    int64_t status_cap = 0;
    unsigned char *d_internal = nullptr;
    ~expect_state_t()
    {
        hipFree(d_PT); hipFree(d_W); hipFree(d_B); hipFree(d_E); hipFree(d_scale); hipFree(d_ones);
        hipFree(d_out); hipFree(d_rootpart); hipFree(d_step_node); hipFree(d_parent_step);
        hipFree(d_ident); hipFree(d_status); hipFree(d_internal);
    }
};

int expect_state_get(rt_model *m, expect_state_t **out)
{
    if (m->expect_state) {
        *out = (expect_state_t *)m->expect_state;
        return RT_OK;
    }
    const int64_t n = m->n, N = m->nnodes;
    const int nops = (int)m->ops.size();
    RT_REQUIRE(nops == N && m->ops[(size_t)nops - 1].dst < 0, "unexpected schedule");
    const bool mfma = n > 4;                   // (n <= 4: the fused lane kernel of passes.hip)
    std::vector<int> step_node((size_t)nops), parent_step((size_t)nops, 0), step_of((size_t)N, -1);
    std::vector<unsigned char> internal((size_t)nops, 0);
    for (int i = 0; i < nops; ++i) {
        step_node[(size_t)i] = m->ops[(size_t)i].node;
        step_of[(size_t)m->ops[(size_t)i].node] = i;
        internal[(size_t)i] = m->ops[(size_t)i].pop >= 0;
    }
    for (int i = 0; i + 1 < nops; ++i)
        parent_step[(size_t)i] = step_of[(size_t)m->parent[(size_t)step_node[(size_t)i]]];
    std::unique_ptr<expect_state_t> st(new (std::nothrow) expect_state_t());
    if (!st) return RT_ERR_NOMEM;
    st->nops = nops;
    const int NT = (int)((n + 15) / 16), KP = ((int)((n + 3) / 4) + 1) / 2;
    const size_t nn = (size_t)n * n, ne = (size_t)(N - 1), mm = 4 * nn;
    if (mfma) RT_HIP(hipMalloc((void **)&st->d_PT, (size_t)nops * NT * KP * 128 * 8));
    RT_HIP(hipMalloc((void **)&st->d_W, (size_t)N * nn * 8));
    RT_HIP(hipMalloc((void **)&st->d_B, ne * mm * 8));
    RT_HIP(hipMalloc((void **)&st->d_E, ne * mm * 8));
    RT_HIP(hipMalloc((void **)&st->d_scale, ne * 8));
    RT_HIP(hipMalloc((void **)&st->d_ones, ne * 8));
    RT_HIP(hipMalloc((void **)&st->d_out, (2 * (size_t)n + nn) * 8));
    RT_HIP(hipMalloc((void **)&st->d_rootpart, (size_t)EX_ROOT_CHUNKS * 64 * 8));
    RT_HIP(hipMalloc((void **)&st->d_step_node, (size_t)nops * 4));
    RT_HIP(hipMalloc((void **)&st->d_parent_step, (size_t)nops * 4));
    RT_HIP(hipMalloc((void **)&st->d_ident, ne * 4));
    RT_HIP(hipMalloc((void **)&st->d_internal, (size_t)nops));
    std::vector<double> ones(ne, 1.0);
    std::vector<int> ident(ne);
    for (size_t e = 0; e < ne; ++e) ident[e] = (int)e;
    RT_HIP(hipMemcpy(st->d_ones, ones.data(), ne * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(st->d_ident, ident.data(), ne * 4, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(st->d_step_node, step_node.data(), (size_t)nops * 4, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(st->d_parent_step, parent_step.data(), (size_t)nops * 4, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(st->d_internal, internal.data(), (size_t)nops, hipMemcpyHostToDevice));
    m->expect_state = st.release();
    *out = (expect_state_t *)m->expect_state;
    return RT_OK;
}

}  // namespace

void rt_expect_state_release(rt_model *m)
{
    if (!m || !m->expect_state) return;
    delete (expect_state_t *)m->expect_state;
    m->expect_state = nullptr;
}

extern "C" int rt_expect_step(rt_model *m, rt_sites *s, int recompute_transitions, double *dwell,
                              double *root_posterior, double *trans, int32_t *status)
{
    RT_REQUIRE(m && s && dwell && root_posterior && trans, "null pointer");
    RT_REQUIRE(s->model == m, "the site batch belongs to another model");
    RT_REQUIRE(m->d_Q && !m->spectral,
               "rt_model_set_rates has not been called (the statistics need the rate matrices)");
    const int64_t n = m->n, N = m->nnodes;
    const bool lane = n <= 4;                  // resident in the lane layout: the fused lane kernel
    if (n > RT_MAX_EXPECT_STATES || N < 2 || s->d_scratch || m->max_depth > RT_FAST_MAX_DEPTH ||
        s->layout != (lane ? RT_LAYOUT_LANE : RT_LAYOUT_MFMA)) {
        rt_set_error("rt_expect_step: resident batches of n <= %d states on trees the fast "
                     "kernels take (n=%lld here)", RT_MAX_EXPECT_STATES, (long long)n);
        return RT_ERR_UNSUPPORTED;
    }
    rt_ctx *ctx = m->ctx;
    RT_HIP(hipSetDevice(ctx->device));
    const int NT = (int)((n + 15) / 16), KS = (int)((n + 3) / 4), KP = (KS + 1) / 2;
    // the four per-step arrays of the passes: nodes x tiles x 8 KB x NT each (n <= 8: one
    // array of nodes x states x sites doubles)
    const double scratch_gb = lane ? (double)N * (double)n * (double)s->nsites * 8.0 / 1e9
                                   : 4.0 * (double)N * (double)((s->nsites + 15) / 16) * NT * 2048.0 / 1e9;
    if (scratch_gb > 96.0) {
        rt_set_error("rt_expect_step: the passes of this batch need %.0f GB of scratch; split the "
                     "batch", scratch_gb);
        return RT_ERR_UNSUPPORTED;
    }
    expect_state_t *st = nullptr;
    RT_TRY(expect_state_get(m, &st));
    const int64_t status_need = (s->nsites + 63) / 64 * 64;
    if (st->status_cap < status_need) {
        hipFree(st->d_status);
        st->d_status = nullptr;
        st->status_cap = 0;
        RT_HIP(hipMalloc((void **)&st->d_status, (size_t)status_need * 4));
        st->status_cap = status_need;
    }
    if (recompute_transitions) RT_TRY(rt_model_recompute_transitions(m));
    RT_REQUIRE(m->have_P, "the model has no transition matrices yet");
    hipStream_t stream = ctx->stream;
    int rc = RT_ERR_UNSUPPORTED;
    rt_sites *x = nullptr;
    if (lane) {
        RT_HIP(hipMemsetAsync(st->d_status, 0, (size_t)status_need * 4, stream));
        RT_TRY(rt_expect_lane_resident(m, s, st->d_W, st->d_status));
        rc = RT_OK;
    } else {
        if (!s->expect_twin) RT_TRY(rt_sites_twin_interpreter(s, &s->expect_twin));
        x = s->expect_twin;
        hipLaunchKernelGGL(pack_pt_kernel, dim3(512), dim3(256), 0, stream, (int)n, NT, KP, st->nops,
                           st->d_step_node, m->d_P, st->d_PT);
        RT_HIP(hipGetLastError());
    }
#define RT_EX(NTV, KSV)                                                                            \
    rc = expect_device_passes<NTV, KSV>(ctx, m, x, st->d_PT, st->d_step_node, st->d_parent_step,   \
                                        st->d_internal, m->d_P, m->d_root, s->d_weights, st->d_W,  \
                                        st->d_status, st->d_rootpart, false)
    if (!lane)
    switch (KS) {
    case 2: RT_EX(1, 2); break;
    case 3: RT_EX(1, 3); break;
    case 4: RT_EX(1, 4); break;
    case 5: RT_EX(2, 5); break;
    case 6: RT_EX(2, 6); break;
    case 7: RT_EX(2, 7); break;
    case 8: RT_EX(2, 8); break;
    case 9: RT_EX(3, 9); break;
    case 10: RT_EX(3, 10); break;
    case 11: RT_EX(3, 11); break;
    case 12: RT_EX(3, 12); break;
    case 13: RT_EX(4, 13); break;
    case 14: RT_EX(4, 14); break;
    case 15: RT_EX(4, 15); break;
    default: RT_EX(4, 16); break;
    }
#undef RT_EX
    RT_TRY(rc);
    // slot 0, column 0 of W: the weighted sum of the root posteriors; edges 1 .. N - 1 follow
    const size_t nn = (size_t)n * n;
    double *d_dwell = st->d_out, *d_rootp = st->d_out + n, *d_trans = st->d_out + 2 * n;
    RT_HIP(hipMemcpy2DAsync(d_rootp, 8, st->d_W, (size_t)n * 8, 8, (size_t)n, hipMemcpyDeviceToDevice,
                            stream));
    RT_TRY(rt_frechet_statistics_device(ctx, n, N - 1, m->d_Q, m->d_qidx + 1, m->d_t + 1, st->d_W + nn,
                                        st->d_B, st->d_E, st->d_scale, st->d_ones, st->d_ident, d_dwell,
                                        d_trans));
    std::vector<double> out(2 * (size_t)n + nn);
    RT_HIP(hipMemcpyAsync(out.data(), st->d_out, out.size() * 8, hipMemcpyDeviceToHost, stream));
    if (status)
        RT_HIP(hipMemcpyAsync(status, st->d_status, (size_t)s->nsites * 4, hipMemcpyDeviceToHost, stream));
    RT_HIP(hipStreamSynchronize(stream));
    memcpy(dwell, out.data(), (size_t)n * 8);
    memcpy(root_posterior, out.data() + n, (size_t)n * 8);
    memcpy(trans, out.data() + 2 * n, nn * 8);
    return RT_OK;
}

void rt_expect_cache_release(rt_ctx *ctx)
{
    if (!ctx || !ctx->expect_cache) return;
    delete (expect_cache_t *)ctx->expect_cache;
    ctx->expect_cache = nullptr;
}

