// The device core of a Rao-Teh sweep: a RAGGED batch of
// trees -- the chunk trees of many independent chains / sites of one sweep, each with
// its own topology (_graph_transform.py:298-375) -- that all share ONE transition
// matrix, the uniformized P = I + Q / omega (_sample_mjp_dense.py:72-114).  Per tree:
//
//   * the two boolean passes of _mcy.py:139-181 / 240-271 (pyfelscore.mcy_get_node_to_pset,
//     pyfelscore.get_node_to_set with a boolean CSR of P; un-accelerated twins
//     _mcy.py:396-470, _mc0.py:89-138): which states keep a positive subtree likelihood,
//     then which are reachable from the root's set -- structural zeros are the normal
//     case with a sparse uniformized P (_sampler.py:615-643);
//   * the upward (Felsenstein) pass with that one P (_mcy.py:611-682):
//         L[v,s] = [s in set(v)] * prod_c sum_s' P[s,s'] L[c,s'];
//   * root-to-leaf sampling of a state for every node from the posterior
//     (_sample_mc0_dense.py:53-98): root ~ root_distn * L[root], child ~ P[parent state] *
//     L[child], with a counter-based generator (Philox-4x32-10 keyed by the caller's
//     seed, counter = (sweep, global node index)): a node's draw does not depend on how
//     the batch is laid out or scheduled, so a sweep is reproducible.
//
// Layout: the trees are concatenated; tree k owns nodes [off[k], off[k+1]) in ITS OWN
// DFS preorder (local node 0 = root), given as the concatenation of the per-tree CSR
// arrays of _density.digraph_to_bool_csr; the library turns that into one parent index
// per node (children come after their parent in preorder, so a reverse sweep over a
// tree's nodes visits children before parents: no schedule, no stack).
// One WAVE per tree, lane = state (n <= 64): allowed-state sets are 64-bit masks, a
// set test over all states is one ballot; P is kept by rows in ELL form (the
// uniformized matrix of a codon model has ~10 entries per row), a message entry is a
// gather from the child's vector in LDS.
#include "common.h"

#include <algorithm>
#include <hipcub/hipcub.hpp>

namespace {

constexpr int FOREST_WAVES = 4;            // waves (trees) per workgroup

struct ell_matrix {
    int width = 0;                         // entries per row (padded with col = row, val = 0)
    int *d_col = nullptr;                  // [width][64]
    double *d_val = nullptr;               // [width][64]
    unsigned long long *d_rowbits = nullptr;   // [64] nonzero pattern of row s
    unsigned long long *d_colbits = nullptr;   // [64] nonzero pattern of column s
    double *d_dense = nullptr;             // [n][n]
};

// ---- Philox-4x32-10 (Salmon et al. 2011), the counter-based generator -----------------

__device__ __forceinline__ void philox_round(unsigned &c0, unsigned &c1, unsigned &c2, unsigned &c3,
                                             unsigned k0, unsigned k1)
{
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// uniform double in [0, 1) with 53 random bits for (seed, sweep, index)
__device__ __forceinline__ double philox_uniform(unsigned long long seed, unsigned long long sweep,
                                                 unsigned long long index)
{
    unsigned c0 = (unsigned)index, c1 = (unsigned)(index >> 32);
    unsigned c2 = (unsigned)sweep, c3 = (unsigned)(sweep >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const unsigned long long bits = ((unsigned long long)c0 << 21) ^ (unsigned long long)(c1 >> 11);
    return (double)(bits & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}

// ---- kernels -------------------------------------------------------------------------------

// A tree's words that one lane writes and the whole wave reads back a few nodes later
// (allowed sets, sampled states) live in a wave-private LDS image when the tree has at
// most FOREST_CAP nodes: LDS operations of one wave execute in order.  Larger trees go
// through global memory with relaxed agent-scope atomics (coherent at L2; memory
// operations of one wave to one address stay in order) -- NOT with __threadfence(): on
// this chip that is a write-back and invalidate of the L2, and with one per node the
// boolean passes of 10 000 chunk trees took 12 ms and the sampling pass 6.4 ms next to
// 0.2 ms for the upward pass that does the arithmetic.
constexpr int FOREST_CAP = 1024;

// (Assumption of the > FOREST_CAP path, stated: a word one lane stores with coherent_store and
// another lane of the SAME wave loads with coherent_load a few instructions later sees the
// store -- same-address accesses of one wave are ordered at L2.  Test:
// test_forest_trees_beyond_the_lds_image.)
__device__ __forceinline__ unsigned long long coherent_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_store(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// backward then forward boolean pass, in place on the allowed-set masks
__global__ void __launch_bounds__(64 * FOREST_WAVES)
forest_sets_kernel(int n, long ntrees, const long *__restrict__ off,
                   const int *__restrict__ parent, const unsigned long long *__restrict__ rowbits,
                   const unsigned long long *__restrict__ colbits,
                   unsigned long long *__restrict__ allowed, int forward)
{
    __shared__ unsigned long long lset[FOREST_WAVES][FOREST_CAP];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long tree = (long)blockIdx.x * FOREST_WAVES + w;
    if (tree >= ntrees) return;
    const long lo = off[tree], hi = off[tree + 1];
    const bool fits = hi - lo <= FOREST_CAP;
    const unsigned long long rb = lane < n ? rowbits[lane] : 0ull;
    const unsigned long long cb = lane < n ? colbits[lane] : 0ull;
    if (fits) {
        for (long i = lane; i < hi - lo; i += 64) lset[w][i] = allowed[lo + i];
        wave_lds_order();
    }
    // backward: a state stays at v only if it has a transition into every child's set
    // (forward: 0 = this pass only, 1 = both, 2 = the forward pass only)
    for (long v = hi - 1; forward != 2 && v > lo; --v) {
        const unsigned long long cset = fits ? lset[w][v - lo] : coherent_load(&allowed[v]);
        const unsigned long long keep = __ballot((rb & cset) != 0ull);
        const long p = lo + parent[v];
        if (fits) {
            if (lane == 0) lset[w][p - lo] &= keep;
            wave_lds_order();
        } else if (lane == 0) {
            coherent_store(&allowed[p], coherent_load(&allowed[p]) & keep);
        }
    }
    // forward: a child state stays only if some state of the parent's set reaches it
    for (long v = lo + 1; forward && v < hi; ++v) {
        const long p = lo + parent[v];
        const unsigned long long pset = fits ? lset[w][p - lo] : coherent_load(&allowed[p]);
        const unsigned long long reach = __ballot((cb & pset) != 0ull);
        if (fits) {
            if (lane == 0) lset[w][v - lo] &= reach;
            wave_lds_order();
        } else if (lane == 0) {
            coherent_store(&allowed[v], coherent_load(&allowed[v]) & reach);
        }
    }
    if (fits)
        for (long i = lane; i < hi - lo; i += 64) allowed[lo + i] = lset[w][i];
}

// upward pass: L[v][s], every entry written
__global__ void __launch_bounds__(64 * FOREST_WAVES)
forest_pmap_kernel(int n, long ntrees, const long *__restrict__ off,
                   const int *__restrict__ parent, int width, const int *__restrict__ ecol,
                   const double *__restrict__ eval,
                   const unsigned long long *__restrict__ allowed, double *__restrict__ L)
{
    __shared__ double vec[FOREST_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long tree = (long)blockIdx.x * FOREST_WAVES + w;
    if (tree >= ntrees) return;
    const long lo = off[tree], hi = off[tree + 1];
    const bool live = lane < n;
    // L[v] starts as the indicator of the node's set; children fold in from the back
    for (long v = lo; v < hi; ++v)
        if (live) L[v * n + lane] = (allowed[v] >> lane) & 1ull ? 1.0 : 0.0;
    for (long v = hi - 1; v > lo; --v) {
        const double mine = live ? L[v * n + lane] : 0.0;
        vec[w][lane] = mine;
        // LDS operations of one wave execute in order; keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double msg = 0.0;
        for (int k = 0; k < width; ++k) {
            const int c = ecol[k * 64 + lane];
            msg = fma(eval[k * 64 + lane], vec[w][c], msg);
        }
        const long p = lo + parent[v];
        if (live) L[p * n + lane] *= msg;      // lane-private address: ordered per lane
        __builtin_amdgcn_wave_barrier();
    }
}

// inclusive prefix sum over the wave (Hillis-Steele on shuffles)
__device__ __forceinline__ double wave_inclusive_sum(double x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

__global__ void __launch_bounds__(64 * FOREST_WAVES)
forest_sample_kernel(int n, long ntrees, const long *__restrict__ off,
                     const int *__restrict__ parent, const double *__restrict__ P,
                     const double *__restrict__ root_distn, const double *__restrict__ L,
                     unsigned long long seed, unsigned long long sweep,
                     int *__restrict__ states, int *__restrict__ status)
{
    __shared__ int lstate[FOREST_WAVES][FOREST_CAP];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long tree = (long)blockIdx.x * FOREST_WAVES + w;
    if (tree >= ntrees) return;
    const long lo = off[tree], hi = off[tree + 1];
    const bool fits = hi - lo <= FOREST_CAP;
    const bool live = lane < n;
    int st = 0;
    for (long v = lo; v < hi; ++v) {
        double prior;
        if (v == lo) prior = live ? (root_distn ? root_distn[lane] : 1.0) : 0.0;
        else {
            const long p = lo + parent[v];
            const int ps = fits ? lstate[w][p - lo]
                                : __hip_atomic_load(&states[p], __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
            prior = (live && ps >= 0) ? P[(long)ps * n + lane] : 0.0;
        }
        const double wgt = live ? fmax(prior * L[v * n + lane], 0.0) : 0.0;
        const double cdf = wave_inclusive_sum(wgt, lane);
        const double total = __shfl(cdf, 63, 64);
        int pick = -1;
        if (total > 0.0 && total < 1e308 * 10.0) {
            const double target = philox_uniform(seed, sweep, (unsigned long long)v) * total;
            // first state whose cumulative weight exceeds the target (a state of weight
            // zero is never picked: its cdf equals its predecessor's)
            const unsigned long long hit = __ballot(wgt > 0.0 && cdf > target);
            pick = hit ? __ffsll((long long)hit) - 1
                       : 63 - __clzll((long long)__ballot(wgt > 0.0));   // rounding at the top
        } else {
            // zero likelihood: at the root the reference raises StructuralZeroProb /
            // NumericalZeroProb (_sample_mc0_dense.py:57-62); below the root it cannot
            // happen with a consistent L, and is reported the same way
            if (st == 0) st = v == lo ? 1 : 2;     // the first failure names the cause
        }
        // every lane holds the same pick; the children read it back from the wave's image
        if (fits) {
            if (lane == 0) {
                lstate[w][v - lo] = pick;
                states[v] = pick;
            }
            wave_lds_order();
        } else if (lane == 0) {
            __hip_atomic_store(&states[v], pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane == 0) status[tree] = st;
}

// ---- small state spaces: several trees per wave ----------------------------------------------
// With a wave per tree and a lane per state, a 4-state model uses 4 of 64 lanes and the three
// passes are chains of per-node round trips (LDS, shuffles) whose latency is the same whatever
// the width.  For n <= 16 a wave takes G = 64 / NP consecutive trees (NP = 4, 8 or 16 lanes
// each): lane = (group, state), ballots are cut into the group's bits, prefix sums run inside
// the group, and the trees' nodes -- consecutive in the forest -- share the wave's LDS image.
// Same arithmetic per tree as the kernels above (same draws: the counter is the node index).
template <int NP>
struct group_geom {
    static constexpr int G = 64 / NP;
    static constexpr unsigned long long mask = NP == 64 ? ~0ull : (1ull << NP) - 1ull;
};

template <int NP>
__global__ void __launch_bounds__(64 * FOREST_WAVES)
forest_sets_grouped_kernel(int n, long ntrees, const long *__restrict__ off,
                           const int *__restrict__ parent,
                           const unsigned long long *__restrict__ rowbits,
                           const unsigned long long *__restrict__ colbits,
                           unsigned long long *__restrict__ allowed, int forward)
{
    constexpr int G = group_geom<NP>::G;
    __shared__ unsigned long long lset[FOREST_WAVES][FOREST_CAP];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int g = lane / NP, s = lane % NP;
    const long t0 = ((long)blockIdx.x * FOREST_WAVES + w) * G;
    if (t0 >= ntrees) return;
    const long tree = t0 + g;
    const bool active = tree < ntrees;
    const long lo = active ? off[tree] : 0, hi = active ? off[tree + 1] : 0;
    const long wlo = off[t0], whi = off[t0 + G < ntrees ? t0 + G : ntrees];
    const bool fits = whi - wlo <= FOREST_CAP;
    const unsigned long long rb = s < n ? rowbits[s] : 0ull;
    const unsigned long long cb = s < n ? colbits[s] : 0ull;
    const int shift = g * NP;
    if (fits) {
        for (long i = lane; i < whi - wlo; i += 64) lset[w][i] = allowed[wlo + i];
        wave_lds_order();
    }
    for (long step = 0; forward != 2; ++step) {
        const long v = hi - 1 - step;
        const bool valid = active && v > lo;
        if (!__any(valid)) break;
        const unsigned long long cset =
            !valid ? 0ull : fits ? lset[w][v - wlo] : coherent_load(&allowed[v]);
        const unsigned long long keep =
            (__ballot(valid && (rb & cset) != 0ull) >> shift) & group_geom<NP>::mask;
        if (valid && s == 0) {
            const long p = lo + parent[v];
            if (fits) lset[w][p - wlo] &= keep;
            else coherent_store(&allowed[p], coherent_load(&allowed[p]) & keep);
        }
        wave_lds_order();
    }
    for (long step = 0; forward; ++step) {
        const long v = lo + 1 + step;
        const bool valid = active && v < hi;
        if (!__any(valid)) break;
        const long p = valid ? lo + parent[v] : 0;
        const unsigned long long pset =
            !valid ? 0ull : fits ? lset[w][p - wlo] : coherent_load(&allowed[p]);
        const unsigned long long reach =
            (__ballot(valid && (cb & pset) != 0ull) >> shift) & group_geom<NP>::mask;
        if (valid && s == 0) {
            if (fits) lset[w][v - wlo] &= reach;
            else coherent_store(&allowed[v], coherent_load(&allowed[v]) & reach);
        }
        wave_lds_order();
    }
    if (fits)
        for (long i = lane; i < whi - wlo; i += 64) allowed[wlo + i] = lset[w][i];
}

template <int NP>
__global__ void __launch_bounds__(64 * FOREST_WAVES)
forest_pmap_grouped_kernel(int n, long ntrees, const long *__restrict__ off,
                           const int *__restrict__ parent, int width,
                           const int *__restrict__ ecol, const double *__restrict__ eval,
                           const unsigned long long *__restrict__ allowed, double *__restrict__ L)
{
    constexpr int G = group_geom<NP>::G;
    __shared__ double vec[FOREST_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int g = lane / NP, s = lane % NP;
    const long t0 = ((long)blockIdx.x * FOREST_WAVES + w) * G;
    if (t0 >= ntrees) return;
    const long tree = t0 + g;
    const bool active = tree < ntrees;
    const long lo = active ? off[tree] : 0, hi = active ? off[tree + 1] : 0;
    const bool live = active && s < n;
    for (long v = lo; v < hi; ++v)
        if (live) L[v * n + s] = (allowed[v] >> s) & 1ull ? 1.0 : 0.0;
    for (long step = 0;; ++step) {
        const long v = hi - 1 - step;
        const bool valid = active && v > lo;
        if (!__any(valid)) break;
        vec[w][lane] = (valid && live) ? L[v * n + s] : 0.0;
        wave_lds_order();
        double msg = 0.0;
        if (s < n)
            for (int k = 0; k < width; ++k)
                msg = fma(eval[k * 64 + s], vec[w][g * NP + ecol[k * 64 + s]], msg);
        if (valid && live) L[(lo + parent[v]) * n + s] *= msg;
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NP>
__global__ void __launch_bounds__(64 * FOREST_WAVES)
forest_sample_grouped_kernel(int n, long ntrees, const long *__restrict__ off,
                             const int *__restrict__ parent, const double *__restrict__ P,
                             const double *__restrict__ root_distn, const double *__restrict__ L,
                             unsigned long long seed, unsigned long long sweep,
                             int *__restrict__ states, int *__restrict__ status)
{
    constexpr int G = group_geom<NP>::G;
    __shared__ int lstate[FOREST_WAVES][FOREST_CAP];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int g = lane / NP, s = lane % NP;
    const long t0 = ((long)blockIdx.x * FOREST_WAVES + w) * G;
    if (t0 >= ntrees) return;
    const long tree = t0 + g;
    const bool active = tree < ntrees;
    const long lo = active ? off[tree] : 0, hi = active ? off[tree + 1] : 0;
    const long wlo = off[t0], whi = off[t0 + G < ntrees ? t0 + G : ntrees];
    const bool fits = whi - wlo <= FOREST_CAP;
    const bool live = active && s < n;
    const int shift = g * NP;
    int st = 0;
    for (long step = 0;; ++step) {
        const long v = lo + step;
        const bool valid = active && v < hi;
        if (!__any(valid)) break;
        double prior = 0.0;
        if (valid && live) {
            if (v == lo) prior = root_distn ? root_distn[s] : 1.0;
            else {
                const long p = lo + parent[v];
                const int ps = fits ? lstate[w][p - wlo]
                                    : __hip_atomic_load(&states[p], __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                prior = ps >= 0 ? P[(long)ps * n + s] : 0.0;
            }
        }
        const double wgt = (valid && live) ? fmax(prior * L[v * n + s], 0.0) : 0.0;
        double cdf = wgt;                         // inclusive prefix sum inside the group
#pragma unroll
        for (int d = 1; d < NP; d <<= 1) {
            const double y = __shfl_up(cdf, d, 64);
            if (s >= d) cdf += y;
        }
        const double total = __shfl(cdf, shift + NP - 1, 64);
        const bool ok = total > 0.0 && total < 1e308 * 10.0;
        const double target = (valid && ok) ? philox_uniform(seed, sweep, (unsigned long long)v) * total
                                            : 0.0;
        const unsigned long long hit =
            (__ballot(valid && ok && wgt > 0.0 && cdf > target) >> shift) & group_geom<NP>::mask;
        const unsigned long long pos =
            (__ballot(valid && wgt > 0.0) >> shift) & group_geom<NP>::mask;
        int pick = -1;
        if (valid) {
            if (ok) pick = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)pos);
            else if (st == 0) st = v == lo ? 1 : 2;
            if (s == 0) {
                if (fits) {
                    lstate[w][v - wlo] = pick;
                    states[v] = pick;
                } else {
                    __hip_atomic_store(&states[v], pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        wave_lds_order();
    }
    if (active && s == 0) status[tree] = st;
}

// the three passes, by state count
unsigned forest_grid_grouped(int64_t ntrees, int G)
{
    const int64_t waves = (ntrees + G - 1) / G;
    return (unsigned)((waves + FOREST_WAVES - 1) / FOREST_WAVES);
}

void launch_forest_sets(int n, long ntrees, const long *off, const int *parent,
                        const unsigned long long *rowbits, const unsigned long long *colbits,
                        unsigned long long *allowed, int forward, hipStream_t st)
{
    const dim3 block(64 * FOREST_WAVES);
    const bool wide = getenv("RAOTEH_FOREST_WAVE_PER_TREE") != nullptr;
    if (n <= 4 && !wide)
        hipLaunchKernelGGL(forest_sets_grouped_kernel<4>, dim3(forest_grid_grouped(ntrees, 16)), block,
                           0, st, n, ntrees, off, parent, rowbits, colbits, allowed, forward);
    else if (n <= 8 && !wide)
        hipLaunchKernelGGL(forest_sets_grouped_kernel<8>, dim3(forest_grid_grouped(ntrees, 8)), block,
                           0, st, n, ntrees, off, parent, rowbits, colbits, allowed, forward);
    else if (n <= 16 && !wide)
        hipLaunchKernelGGL(forest_sets_grouped_kernel<16>, dim3(forest_grid_grouped(ntrees, 4)), block,
                           0, st, n, ntrees, off, parent, rowbits, colbits, allowed, forward);
    else
        hipLaunchKernelGGL(forest_sets_kernel, dim3(forest_grid_grouped(ntrees, 1)), block, 0, st, n,
                           ntrees, off, parent, rowbits, colbits, allowed, forward);
}

void launch_forest_pmap(int n, long ntrees, const long *off, const int *parent, int width,
                        const int *ecol, const double *eval, const unsigned long long *allowed,
                        double *L, hipStream_t st)
{
    const dim3 block(64 * FOREST_WAVES);
    const bool wide = getenv("RAOTEH_FOREST_WAVE_PER_TREE") != nullptr;
    if (n <= 4 && !wide)
        hipLaunchKernelGGL(forest_pmap_grouped_kernel<4>, dim3(forest_grid_grouped(ntrees, 16)), block,
                           0, st, n, ntrees, off, parent, width, ecol, eval, allowed, L);
    else if (n <= 8 && !wide)
        hipLaunchKernelGGL(forest_pmap_grouped_kernel<8>, dim3(forest_grid_grouped(ntrees, 8)), block,
                           0, st, n, ntrees, off, parent, width, ecol, eval, allowed, L);
    else if (n <= 16 && !wide)
        hipLaunchKernelGGL(forest_pmap_grouped_kernel<16>, dim3(forest_grid_grouped(ntrees, 4)), block,
                           0, st, n, ntrees, off, parent, width, ecol, eval, allowed, L);
    else
        hipLaunchKernelGGL(forest_pmap_kernel, dim3(forest_grid_grouped(ntrees, 1)), block, 0, st, n,
                           ntrees, off, parent, width, ecol, eval, allowed, L);
}

void launch_forest_sample(int n, long ntrees, const long *off, const int *parent, const double *P,
                          const double *root_distn, const double *L, unsigned long long seed,
                          unsigned long long sweep, int *states, int *status, hipStream_t st)
{
    const dim3 block(64 * FOREST_WAVES);
    const bool wide = getenv("RAOTEH_FOREST_WAVE_PER_TREE") != nullptr;
    if (n <= 4 && !wide)
        hipLaunchKernelGGL(forest_sample_grouped_kernel<4>, dim3(forest_grid_grouped(ntrees, 16)),
                           block, 0, st, n, ntrees, off, parent, P, root_distn, L, seed, sweep, states,
                           status);
    else if (n <= 8 && !wide)
        hipLaunchKernelGGL(forest_sample_grouped_kernel<8>, dim3(forest_grid_grouped(ntrees, 8)),
                           block, 0, st, n, ntrees, off, parent, P, root_distn, L, seed, sweep, states,
                           status);
    else if (n <= 16 && !wide)
        hipLaunchKernelGGL(forest_sample_grouped_kernel<16>, dim3(forest_grid_grouped(ntrees, 4)),
                           block, 0, st, n, ntrees, off, parent, P, root_distn, L, seed, sweep, states,
                           status);
    else
        hipLaunchKernelGGL(forest_sample_kernel, dim3(forest_grid_grouped(ntrees, 1)), block, 0, st, n,
                           ntrees, off, parent, P, root_distn, L, seed, sweep, states, status);
}

// ---- host side -------------------------------------------------------------------------------

struct forest_dev {
    long *d_off = nullptr;
    int *d_parent = nullptr;
    unsigned long long *d_allowed = nullptr;
    double *d_L = nullptr;
    double *d_root = nullptr;
    int *d_states = nullptr;
    int *d_status = nullptr;
    ell_matrix P;
    ~forest_dev()
    {
        hipFree(d_off); hipFree(d_parent); hipFree(d_allowed); hipFree(d_L); hipFree(d_root);
        hipFree(d_states); hipFree(d_status);
        hipFree(P.d_col); hipFree(P.d_val); hipFree(P.d_rowbits); hipFree(P.d_colbits);
        hipFree(P.d_dense);
    }
};

// concatenated per-tree CSR -> one local parent index per node; validates the layout
int forest_parents(int64_t ntrees, const int64_t *off, const int64_t *idx, const int64_t *ptr,
                   std::vector<int> &parent)
{
    RT_REQUIRE(ntrees >= 1 && off && ptr, "null forest arrays");
    RT_REQUIRE(off[0] == 0, "tree_node_offset[0] must be 0");
    const int64_t total = off[ntrees];
    parent.assign((size_t)total, -1);
    for (int64_t k = 0; k < ntrees; ++k) {
        const int64_t lo = off[k], nn = off[k + 1] - off[k];
        RT_REQUIRE(nn >= 1 && nn < (1ll << 30), "tree %lld has %lld nodes", (long long)k, (long long)nn);
        const int64_t *p = ptr + lo + k;            // nn + 1 entries
        const int64_t *ix = idx ? idx + lo - k : nullptr;   // nn - 1 entries
        RT_REQUIRE(p[0] == 0 && p[nn] == nn - 1, "tree %lld: indptr does not describe a tree",
                   (long long)k);
        for (int64_t v = 0; v < nn; ++v) {
            RT_REQUIRE(p[v + 1] >= p[v], "tree %lld: indptr not monotone", (long long)k);
            for (int64_t e = p[v]; e < p[v + 1]; ++e) {
                const int64_t c = ix[e];
                RT_REQUIRE(c > v && c < nn, "tree %lld: child %lld of node %lld not in preorder",
                           (long long)k, (long long)c, (long long)v);
                RT_REQUIRE(parent[(size_t)(lo + c)] < 0, "tree %lld: node %lld has two parents",
                           (long long)k, (long long)c);
                parent[(size_t)(lo + c)] = (int)v;
            }
        }
    }
    return RT_OK;
}

int upload_matrix(int64_t n, const double *P, ell_matrix &M)
{
    std::vector<unsigned long long> rowbits(64, 0), colbits(64, 0);
    int width = 1;
    for (int64_t r = 0; r < n; ++r) {
        int cnt = 0;
        for (int64_t c = 0; c < n; ++c)
            if (P[r * n + c] != 0.0) {
                rowbits[(size_t)r] |= 1ull << c;
                colbits[(size_t)c] |= 1ull << r;
                ++cnt;
            }
        width = std::max(width, cnt);
    }
    std::vector<int> col((size_t)width * 64);
    std::vector<double> val((size_t)width * 64, 0.0);
    for (int r = 0; r < 64; ++r) {
        int k = 0;
        if (r < n)
            for (int64_t c = 0; c < n; ++c)
                if (P[r * n + c] != 0.0) {
                    col[(size_t)k * 64 + r] = (int)c;
                    val[(size_t)k * 64 + r] = P[r * n + c];
                    ++k;
                }
        for (; k < width; ++k) col[(size_t)k * 64 + r] = r < n ? r : 0;
    }
    M.width = width;
    RT_HIP(hipMalloc((void **)&M.d_col, col.size() * 4));
    RT_HIP(hipMalloc((void **)&M.d_val, val.size() * 8));
    RT_HIP(hipMalloc((void **)&M.d_rowbits, 64 * 8));
    RT_HIP(hipMalloc((void **)&M.d_colbits, 64 * 8));
    RT_HIP(hipMalloc((void **)&M.d_dense, (size_t)n * n * 8));
    RT_HIP(hipMemcpy(M.d_col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(M.d_val, val.data(), val.size() * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(M.d_rowbits, rowbits.data(), 64 * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(M.d_colbits, colbits.data(), 64 * 8, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(M.d_dense, P, (size_t)n * n * 8, hipMemcpyHostToDevice));
    return RT_OK;
}

// the same layout given as one local parent index per node (root: -1, parent < child)
int forest_check_parents(int64_t ntrees, const int64_t *off, const int32_t *parent)
{
    RT_REQUIRE(ntrees >= 1 && off && parent, "null forest arrays");
    RT_REQUIRE(off[0] == 0, "tree_node_offset[0] must be 0");
    for (int64_t k = 0; k < ntrees; ++k) {
        const int64_t lo = off[k], nn = off[k + 1] - off[k];
        RT_REQUIRE(nn >= 1 && nn < (1ll << 30), "tree %lld has %lld nodes", (long long)k, (long long)nn);
        RT_REQUIRE(parent[lo] == -1, "tree %lld: node 0 must be the root (parent -1)", (long long)k);
        for (int64_t v = 1; v < nn; ++v)
            RT_REQUIRE(parent[lo + v] >= 0 && parent[lo + v] < v,
                       "tree %lld: parent %d of node %lld is not before it", (long long)k,
                       parent[lo + v], (long long)v);
    }
    return RT_OK;
}

int forest_upload(rt_ctx *ctx, int64_t n, int64_t ntrees, const int64_t *off, const int64_t *idx,
                  const int64_t *ptr, const double *P, forest_dev &f,
                  const int32_t *given_parent = nullptr)
{
    RT_REQUIRE(ctx, "null context");
    RT_REQUIRE(n >= 1 && n <= 64, "the forest passes hold a state per lane: n <= 64");
    RT_REQUIRE(P, "null transition matrix");
    std::vector<int> parent;
    if (given_parent) {
        RT_TRY(forest_check_parents(ntrees, off, given_parent));
        parent.assign(given_parent, given_parent + off[ntrees]);
    } else {
        RT_TRY(forest_parents(ntrees, off, idx, ptr, parent));
    }
    const int64_t total = off[ntrees];
    RT_HIP(hipSetDevice(ctx->device));
    std::vector<long> loff((size_t)ntrees + 1);
    for (int64_t k = 0; k <= ntrees; ++k) loff[(size_t)k] = (long)off[k];
    RT_HIP(hipMalloc((void **)&f.d_off, (ntrees + 1) * sizeof(long)));
    RT_HIP(hipMalloc((void **)&f.d_parent, total * 4));
    RT_HIP(hipMemcpy(f.d_off, loff.data(), (ntrees + 1) * sizeof(long), hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(f.d_parent, parent.data(), total * 4, hipMemcpyHostToDevice));
    return upload_matrix(n, P, f.P);
}

unsigned forest_grid(int64_t ntrees) { return (unsigned)((ntrees + FOREST_WAVES - 1) / FOREST_WAVES); }

}  // namespace

extern "C" int rt_forest_passes(rt_ctx *ctx, int64_t n, int64_t ntrees,
                                const int64_t *tree_node_offset, const int64_t *tree_csr_indices,
                                const int64_t *tree_csr_indptr, const double *P,
                                uint64_t *allowed_sets, double *subtree_probability)
{
    RT_REQUIRE(allowed_sets, "null allowed_sets");
    forest_dev f;
    RT_TRY(forest_upload(ctx, n, ntrees, tree_node_offset, tree_csr_indices, tree_csr_indptr, P, f));
    const int64_t total = tree_node_offset[ntrees];
    hipStream_t st = ctx->stream;
    RT_HIP(hipMalloc((void **)&f.d_allowed, total * 8));
    RT_HIP(hipMemcpyAsync(f.d_allowed, allowed_sets, total * 8, hipMemcpyHostToDevice, st));
    launch_forest_sets((int)n, (long)ntrees, f.d_off, f.d_parent, f.P.d_rowbits, f.P.d_colbits,
                       (unsigned long long *)f.d_allowed, 1, st);
    RT_HIP(hipGetLastError());
    if (subtree_probability) {
        RT_HIP(hipMalloc((void **)&f.d_L, total * n * 8));
        launch_forest_pmap((int)n, (long)ntrees, f.d_off, f.d_parent, f.P.width, f.P.d_col, f.P.d_val,
                           (const unsigned long long *)f.d_allowed, f.d_L, st);
        RT_HIP(hipGetLastError());
        RT_HIP(hipMemcpyAsync(subtree_probability, f.d_L, total * n * 8, hipMemcpyDeviceToHost, st));
    }
    RT_HIP(hipMemcpyAsync(allowed_sets, f.d_allowed, total * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

// pyfelscore.mcy_get_node_to_pset (_mcy.py:158,259; twin _mcy.py:396-470) and
// pyfelscore.get_node_to_set (_mcy.py:168; twin _mc0.py:89-138) on the reference's own arrays:
// ONE tree, the transition matrix as a boolean CSR shared by every edge, the int64 0/1 state
// mask updated in place.  The forest kernels above on a forest of one tree.
static int shared_matrix_mask_pass(rt_ctx *ctx, int mode, int64_t nnodes, int64_t n,
                                   const int64_t *idx, const int64_t *ptr, const int64_t *tidx,
                                   const int64_t *tptr, int64_t *state_mask)
{
    RT_REQUIRE(ctx, "null context");
    RT_REQUIRE(nnodes >= 1 && n >= 1 && ptr && tptr && state_mask, "bad arguments");
    RT_REQUIRE(n <= 64, "the forest passes hold a state per lane: n <= 64");
    RT_REQUIRE(tptr[0] == 0 && tptr[n] >= 0 && (tidx || tptr[n] == 0), "bad transition CSR");
    std::vector<double> P((size_t)n * n, 0.0);
    for (int64_t a = 0; a < n; ++a) {
        RT_REQUIRE(tptr[a + 1] >= tptr[a], "trans_csr_indptr not monotone");
        for (int64_t e = tptr[a]; e < tptr[a + 1]; ++e) {
            RT_REQUIRE(tidx[e] >= 0 && tidx[e] < n, "trans_csr_indices out of range");
            P[(size_t)(a * n + tidx[e])] = 1.0;
        }
    }
    const int64_t off[2] = {0, nnodes};
    forest_dev f;
    RT_TRY(forest_upload(ctx, n, 1, off, idx, ptr, P.data(), f));
    std::vector<unsigned long long> sets((size_t)nnodes, 0ull);
    for (int64_t v = 0; v < nnodes; ++v)
        for (int64_t a = 0; a < n; ++a)
            if (state_mask[v * n + a]) sets[(size_t)v] |= 1ull << a;
    hipStream_t st = ctx->stream;
    RT_HIP(hipMalloc((void **)&f.d_allowed, nnodes * 8));
    RT_HIP(hipMemcpyAsync(f.d_allowed, sets.data(), nnodes * 8, hipMemcpyHostToDevice, st));
    launch_forest_sets((int)n, 1, f.d_off, f.d_parent, f.P.d_rowbits, f.P.d_colbits,
                       (unsigned long long *)f.d_allowed, mode, st);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(sets.data(), f.d_allowed, nnodes * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    for (int64_t v = 0; v < nnodes; ++v)
        for (int64_t a = 0; a < n; ++a) state_mask[v * n + a] = (int64_t)((sets[(size_t)v] >> a) & 1ull);
    return RT_OK;
}

extern "C" int rt_mcy_get_node_to_pset(rt_ctx *ctx, int64_t nnodes, int64_t n,
        const int64_t *tree_csr_indices, const int64_t *tree_csr_indptr,
        const int64_t *trans_csr_indices, const int64_t *trans_csr_indptr, int64_t *state_mask)
{
    return shared_matrix_mask_pass(ctx, 0, nnodes, n, tree_csr_indices, tree_csr_indptr,
                                   trans_csr_indices, trans_csr_indptr, state_mask);
}

extern "C" int rt_get_node_to_set(rt_ctx *ctx, int64_t nnodes, int64_t n,
        const int64_t *tree_csr_indices, const int64_t *tree_csr_indptr,
        const int64_t *trans_csr_indices, const int64_t *trans_csr_indptr, int64_t *state_mask,
        int64_t *tmp_state_mask)
{
    // tmp_state_mask int64[n]: the scratch row pyfelscore asks its caller for (_mcy.py:166-174);
    // left zeroed, as the caller hands it over
    if (tmp_state_mask) memset(tmp_state_mask, 0, (size_t)n * 8);
    return shared_matrix_mask_pass(ctx, 2, nnodes, n, tree_csr_indices, tree_csr_indptr,
                                   trans_csr_indices, trans_csr_indptr, state_mask);
}

static int forest_resample_impl(rt_ctx *ctx, int64_t n, int64_t ntrees,
                                const int64_t *tree_node_offset, const int64_t *tree_csr_indices,
                                const int64_t *tree_csr_indptr, const int32_t *tree_parent,
                                const double *P, const double *root_distn, uint64_t *allowed_sets,
                                uint64_t seed, uint64_t sweep, int32_t *states, int32_t *status,
                                double *subtree_probability)
{
    RT_REQUIRE(allowed_sets && states && status, "null output arrays");
    forest_dev f;
    RT_TRY(forest_upload(ctx, n, ntrees, tree_node_offset, tree_csr_indices, tree_csr_indptr, P, f,
                         tree_parent));
    const int64_t total = tree_node_offset[ntrees];
    hipStream_t st = ctx->stream;
    RT_HIP(hipMalloc((void **)&f.d_allowed, total * 8));
    RT_HIP(hipMalloc((void **)&f.d_L, total * n * 8));
    RT_HIP(hipMalloc((void **)&f.d_states, total * 4));
    RT_HIP(hipMalloc((void **)&f.d_status, ntrees * 4));
    if (root_distn) {
        RT_HIP(hipMalloc((void **)&f.d_root, n * 8));
        RT_HIP(hipMemcpyAsync(f.d_root, root_distn, n * 8, hipMemcpyHostToDevice, st));
    }
    RT_HIP(hipMemcpyAsync(f.d_allowed, allowed_sets, total * 8, hipMemcpyHostToDevice, st));
    // one sweep's worth of passes back to back on the stream: sets, pmap, sampling
    launch_forest_sets((int)n, (long)ntrees, f.d_off, f.d_parent, f.P.d_rowbits, f.P.d_colbits,
                       (unsigned long long *)f.d_allowed, 1, st);
    launch_forest_pmap((int)n, (long)ntrees, f.d_off, f.d_parent, f.P.width, f.P.d_col, f.P.d_val,
                       (const unsigned long long *)f.d_allowed, f.d_L, st);
    launch_forest_sample((int)n, (long)ntrees, f.d_off, f.d_parent, f.P.d_dense, f.d_root, f.d_L,
                         (unsigned long long)seed, (unsigned long long)sweep, f.d_states, f.d_status,
                         st);
    RT_HIP(hipGetLastError());
    RT_HIP(hipMemcpyAsync(states, f.d_states, total * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(hipMemcpyAsync(status, f.d_status, ntrees * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(hipMemcpyAsync(allowed_sets, f.d_allowed, total * 8, hipMemcpyDeviceToHost, st));
    if (subtree_probability)
        RT_HIP(hipMemcpyAsync(subtree_probability, f.d_L, total * n * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

extern "C" int rt_forest_resample_states(rt_ctx *ctx, int64_t n, int64_t ntrees,
                                         const int64_t *tree_node_offset,
                                         const int64_t *tree_csr_indices,
                                         const int64_t *tree_csr_indptr, const double *P,
                                         const double *root_distn, uint64_t *allowed_sets,
                                         uint64_t seed, uint64_t sweep, int32_t *states,
                                         int32_t *status, double *subtree_probability)
{
    return forest_resample_impl(ctx, n, ntrees, tree_node_offset, tree_csr_indices, tree_csr_indptr,
                                nullptr, P, root_distn, allowed_sets, seed, sweep, states, status,
                                subtree_probability);
}

extern "C" int rt_forest_resample_states_parents(rt_ctx *ctx, int64_t n, int64_t ntrees,
                                                 const int64_t *tree_node_offset,
                                                 const int32_t *tree_parent, const double *P,
                                                 const double *root_distn, uint64_t *allowed_sets,
                                                 uint64_t seed, uint64_t sweep, int32_t *states,
                                                 int32_t *status)
{
    RT_REQUIRE(tree_parent, "null parent array");
    return forest_resample_impl(ctx, n, ntrees, tree_node_offset, nullptr, nullptr, tree_parent, P,
                                root_distn, allowed_sets, seed, sweep, states, status, nullptr);
}

// =================================================================================================
// Device-resident Rao-Teh sweeps: the histories of a batch of chains stay in HBM
// =================================================================================================
// raoteh_amd/_sampler.py runs steps 1, 2, 3 and 5 of a sweep (_sampler.py:366-390) as numpy
// passes over the segment rows around the kernels above: 108 ms per sweep of 10 000 chains on
// a 127-node tree, 1.7 ms of it on the device.  Here the rows never leave the device.  A chain's
// history is a run of rows (edge = preorder index of the edge's lower node, length, state) sorted
// by (edge, position along the edge); two row buffers alternate.  One sweep is
//
//   count   one wave per chain: Poisson events of every row (exponential gaps at rate
//           omega - q(state) until they pass the row's end, _sample_mjp_dense.py:47-61; Philox
//           counter = (chain, row, gap), so the second pass regenerates the same gaps)
//   scan    exclusive prefix sum of the new row counts over the chains (= where each chain's
//           rows and, shifted, its chunks start; rocPRIM's device scan)
//   split   one wave per chain: write the new rows, count the events per edge (LDS), number the
//           chunks -- the first piece of an edge lies in the chunk of its upper node, the last in
//           the chunk of its lower node, pieces in between are chunks of their own
//           (_graph_transform.py:298-375) -- in preorder of the base edges so that a chunk's
//           parent precedes it, AND the allowed sets of the base nodes of a chunk together
//   sets / pmap / sample   the three forest kernels above on the chunk trees
//   merge   one wave per chain: rows take their chunk's state; neighbours of one edge with equal
//           states fuse (the event between them was a self transition, _graph_transform.py:55-83)
//
// All sums are taken in a fixed order: a (seed, batch) pair gives the same histories every run.
namespace {

constexpr int SWEEP_WAVES = 4;             // chains per workgroup
constexpr int SWEEP_MAX_NODES = 1024;      // base tree nodes (LDS tables per wave)
// Poisson events per row: the count of a row travels from the count kernel to the split
// kernel in 16 bits, the draw of gap k has the counter (chain, row, k & 255) in the stream
// `stream + ((k >> 8) << 40)`.  A row that reaches the cap (rate x length of the order of
// 60 000: not a branch of a tree anybody samples) fails the sweep with an error -- it is
// never silently emitted as one event-free piece (the reference, _sample_mjp_dense.py:47-61,
// has no cap at all).
constexpr int SWEEP_MAX_GAPS = 65535;

struct chains_tables {                     // per wave, in dynamic LDS
    int *rows;          // [N] new rows on the edge above node v
    int *first;         // [N] index of the edge's first new row within the chain
    int *node;          // [N] local chunk of base node v
    unsigned long long *acc;   // [N] allowed sets ANDed over the event-free subtree below v
};

__device__ __forceinline__ int wave_inclusive_sum_int(int x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// events of one row: gaps ~ Exp(rate) until the row's end; returns their number and, through
// `emit`, the piece lengths (k + 1 of them)
template <class Emit>
__device__ __forceinline__ int row_events(double rate, double len, unsigned long long seed,
                                          unsigned long long stream, unsigned long long chain,
                                          unsigned long long row, int per, Emit emit)
{
    if (per > 0) {                         // bisection mode: per equal pieces
        for (int j = 0; j < per; ++j) emit(j, len / per);
        return per - 1;
    }
    int k = 0;
    double t = 0.0;
    if (rate > 0.0) {
        while (k < SWEEP_MAX_GAPS) {
            const double u = philox_uniform(seed, stream + ((unsigned long long)(k >> 8) << 40),
                                            (chain << 40) | (row << 8) | (unsigned long long)(k & 255));
            const double gap = -log1p(-u) / rate;
            if (!(t + gap < len)) break;
            emit(k, gap);
            t += gap;
            ++k;
        }
    }
    emit(k, len - t);
    return k;
}

__global__ void __launch_bounds__(64 * SWEEP_WAVES)
sweep_count_kernel(long nchains, const long *__restrict__ start, const int *__restrict__ cnt,
                   const double *__restrict__ len, const int *__restrict__ state,
                   const double *__restrict__ rates, int per, unsigned long long seed,
                   unsigned long long stream, long *__restrict__ newcnt,
                   unsigned short *__restrict__ row_k, int *__restrict__ overflow)
{
    const int lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * SWEEP_WAVES + (threadIdx.x >> 6);
    if (c >= nchains) return;
    const long lo = start[c];
    int total = 0;
    for (int i = lane; i < cnt[c]; i += 64) {
        const int k = row_events(rates[state[lo + i]], len[lo + i], seed, stream,
                                 (unsigned long long)c, (unsigned long long)i, per,
                                 [](int, double) {});
        row_k[lo + i] = (unsigned short)k;       // <= SWEEP_MAX_GAPS: the split pass reads it back
        if (per == 0 && k >= SWEEP_MAX_GAPS) atomicOr(overflow, 1);     // the cap cut this row short
        total += k + 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if (lane == 0) newcnt[c] = total;
}

__global__ void __launch_bounds__(64 * SWEEP_WAVES)
sweep_split_kernel(long nchains, int N, int nbits, const int *__restrict__ parent,
                   const long *__restrict__ start, const int *__restrict__ cnt,
                   const int *__restrict__ edge, const double *__restrict__ len,
                   const int *__restrict__ state, const double *__restrict__ rates, int per,
                   unsigned long long seed, unsigned long long stream,
                   const unsigned short *__restrict__ row_k,
                   const unsigned long long *__restrict__ node_masks,
                   const long *__restrict__ newstart, int *__restrict__ edge_out,
                   double *__restrict__ len_out, int *__restrict__ row_chunk,
                   long *__restrict__ choff, int *__restrict__ cparent,
                   unsigned long long *__restrict__ cmask, int *__restrict__ node_chunk)
{
    extern __shared__ unsigned long long sweep_lds[];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * SWEEP_WAVES + w;
    if (c >= nchains) return;
    // tables of this wave: acc (8 B) first, then three int arrays
    unsigned long long *acc = sweep_lds + (size_t)w * N * 3;           // N * 8 B
    int *rows = (int *)(acc + N);                                       // 3 N ints = N * 12 B
    int *first = rows + N;
    int *node = first + N;
    const unsigned long long full = nbits >= 64 ? ~0ull : (1ull << nbits) - 1ull;
    const long lo = start[c], out = newstart[c];
    const long ch = out - c * (long)(N - 2);        // chunks of a chain = its rows - (N - 2)
    const int total = (int)(newstart[c + 1] - out);
    if (lane == 0) choff[c] = ch;
    if (c == nchains - 1 && lane == 0) choff[nchains] = newstart[nchains] - nchains * (long)(N - 2);
    for (int v = lane; v < N; v += 64) rows[v] = 0;
    wave_lds_order();
    // pass 1: the new rows, in order; rows per edge
    int running = 0;
    const int n_old = cnt[c];
    for (int base = 0; base < n_old; base += 64) {
        const int i = base + lane;
        const bool valid = i < n_old;
        const int e = valid ? edge[lo + i] : 0;
        const double l = valid ? len[lo + i] : 0.0;
        const double r = valid ? rates[state[lo + i]] : 0.0;
        const int k = valid ? (int)row_k[lo + i] : -1;      // counted by sweep_count_kernel
        const int incl = wave_inclusive_sum_int(k + 1, lane);
        const long dst = out + running + incl - (k + 1);
        if (valid) {
            row_events(r, l, seed, stream, (unsigned long long)c, (unsigned long long)i, per,
                       [&](int j, double piece) {
                           // (read back by other lanes of this wave in pass 2)
                           __hip_atomic_store(&edge_out[dst + j], e, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
                           len_out[dst + j] = piece;
                       });
            atomicAdd(&rows[e], k + 1);
        }
        running += __shfl(incl, 63, 64);
    }
    wave_lds_order();
    // first new row of every edge (exclusive prefix over the edges in preorder)
    int carry = 0;
    for (int base = 0; base < N; base += 64) {
        const int v = base + lane;
        const int x = v < N ? rows[v] : 0;
        const int incl = wave_inclusive_sum_int(x, lane);
        if (v < N) first[v] = carry + incl - x;
        carry += __shfl(incl, 63, 64);
    }
    wave_lds_order();
    // chunk of every base node (top-down) and the allowed sets of each event-free subtree
    // (bottom-up): chains of dependent LDS accesses, done by the whole wave in step
    for (int v = lane; v < N; v += 64) acc[v] = node_masks[c * N + v] & full;
    wave_lds_order();
    // the chunk of a base node is the chunk of its nearest ancestor-or-self whose edge carries
    // events (or the root): pointer jumping, ceil(log2 N) rounds of two LDS reads per node (a
    // walk by one lane was 2 N dependent LDS round trips: 1.06 ms of a 2.4 ms sweep of 100 000
    // chains); a jump may read an already updated entry -- still an ancestor on the path
    for (int v = lane; v < N; v += 64) node[v] = (v == 0 || rows[v] > 1) ? v : parent[v];
    wave_lds_order();
    for (int span = 1; span < N; span <<= 1) {
        for (int v = lane; v < N; v += 64) node[v] = node[node[v]];
        wave_lds_order();
    }
    // allowed sets of the event-free region below each such node, ANDed into it
    for (int v = lane; v < N; v += 64) {
        const int top = node[v];
        if (top != v) atomicAnd(&acc[top], acc[v]);
    }
    wave_lds_order();
    {
        int cid[(SWEEP_MAX_NODES + 63) / 64];
#pragma unroll
        for (int q = 0; q < (SWEEP_MAX_NODES + 63) / 64; ++q) {
            const int v = q * 64 + lane;
            if (v < N) {
                const int top = node[v];
                cid[q] = top == 0 ? 0 : first[top] - top + 2 + (rows[top] - 1) - 1;
            }
        }
        wave_lds_order();
#pragma unroll
        for (int q = 0; q < (SWEEP_MAX_NODES + 63) / 64; ++q) {
            const int v = q * 64 + lane;
            if (v < N) node[v] = cid[q];
        }
    }
    wave_lds_order();
    for (int v = lane; v < N; v += 64) node_chunk[c * N + v] = node[v];
    if (lane == 0) {
        cparent[ch] = -1;
        cmask[ch] = acc[0];
    }
    // pass 2: chunk of every new row; parents and masks of the chunks the rows open
    // (the rows were written by other lanes of this wave a moment ago, and a neighbouring
    // chain's wave on this CU may have pulled the shared cache line into the vector L1 before
    // that: read them at L2).  Both sides are relaxed agent-scope atomics on the same words
    // (pass 1 stores edge_out with coherent_store_int), ordered across the lanes of the wave
    // by the wavefront-scope release of wave_lds_order() above.  (An agent-scope fence here
    // instead is an L2 write-back on this chip: the sweep of 100 000 chains went from 1.5 to
    // 5.2 ms with one per chain.)
    for (int j = lane; j < total; j += 64) {
        const int v = __hip_atomic_load(&edge_out[out + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int k = j - first[v];
        const int m = rows[v] - 1;
        const int base_id = first[v] - v + 2;                // 1 + events on the edges before v
        const int up = node[parent[v]];
        const int local = k == 0 ? up : base_id + k - 1;
        row_chunk[out + j] = local;            // local index; global = choff[c] + local
        if (k >= 1) {
            cparent[ch + local] = k == 1 ? up : local - 1;
            cmask[ch + local] = k == m ? acc[v] : full;
        }
    }
}

__global__ void __launch_bounds__(64 * SWEEP_WAVES)
sweep_merge_kernel(long nchains, int N, const long *__restrict__ newstart,
                   const long *__restrict__ choff, const int *__restrict__ edge_in,
                   const double *__restrict__ len_in, const int *__restrict__ row_chunk,
                   const int *__restrict__ cstate, const int *__restrict__ node_chunk,
                   int *__restrict__ edge_out, double *__restrict__ len_out,
                   int *__restrict__ state_out, long *__restrict__ start, int *__restrict__ cnt,
                   int *__restrict__ node_state)
{
    const int lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * SWEEP_WAVES + (threadIdx.x >> 6);
    if (c >= nchains) return;
    const long lo = newstart[c], ch = choff[c];
    const int total = (int)(newstart[c + 1] - lo);
    auto st = [&](int j) { return cstate[ch + row_chunk[lo + j]]; };
    auto is_start = [&](int j) {
        return j == 0 || edge_in[lo + j] != edge_in[lo + j - 1] || st(j) != st(j - 1);
    };
    int written = 0;
    for (int base = 0; base < total; base += 64) {
        const int j = base + lane;
        const bool head = j < total && is_start(j);
        const int incl = wave_inclusive_sum_int(head ? 1 : 0, lane);
        if (head) {
            double sum = len_in[lo + j];
            for (int jj = j + 1; jj < total && !is_start(jj); ++jj) sum += len_in[lo + jj];
            const long dst = lo + written + incl - 1;
            edge_out[dst] = edge_in[lo + j];
            len_out[dst] = sum;
            state_out[dst] = st(j);
        }
        written += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
        start[c] = lo;
        cnt[c] = written;
    }
    for (int v = lane; v < N; v += 64) node_state[c * N + v] = cstate[ch + node_chunk[c * N + v]];
}

// status of the forest pass -> one flag; per-chain statistics on request
__global__ void __launch_bounds__(256)
sweep_flag_kernel(long nchains, const int *__restrict__ status, int *__restrict__ flag)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c < nchains && status[c] != 0) atomicMax(flag, status[c]);
}

__global__ void __launch_bounds__(64 * SWEEP_WAVES)
sweep_stats_kernel(long nchains, int n, const long *__restrict__ start, const int *__restrict__ cnt,
                   const int *__restrict__ edge, const double *__restrict__ len,
                   const int *__restrict__ state, double *__restrict__ dwell,
                   long long *__restrict__ trans)
{
    const int lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * SWEEP_WAVES + (threadIdx.x >> 6);
    if (c >= nchains) return;
    const long lo = start[c];
    const int m = cnt[c];
    if (dwell) {                 // lane = state: every lane walks the rows in order
        double sum = 0.0;
        for (int i = 0; i < m; ++i)
            if (state[lo + i] == lane) sum += len[lo + i];
        if (lane < n) dwell[c * n + lane] = sum;
    }
    if (trans) {                 // zeroed by the host; integer adds: order does not matter
        for (int i = 1 + lane; i < m; i += 64)
            if (edge[lo + i] == edge[lo + i - 1])
                atomicAdd((unsigned long long *)&trans[(c * n + state[lo + i - 1]) * (long)n +
                                                       state[lo + i]], 1ull);
    }
}

// Metropolis-Hastings: the chains flagged in `reject` get the rows of the snapshot back.  A
// chain's current run has room for them: the sweep since the snapshot only added rows
// before the merge, and the run keeps the room of the split rows.
__global__ void __launch_bounds__(64 * SWEEP_WAVES)
sweep_restore_kernel(long nchains, int N, const unsigned char *__restrict__ reject,
                     const long *__restrict__ start_k, const int *__restrict__ cnt_k,
                     const int *__restrict__ edge_k, const double *__restrict__ len_k,
                     const int *__restrict__ state_k, const int *__restrict__ node_state_k,
                     const long *__restrict__ start, int *__restrict__ cnt,
                     int *__restrict__ edge, double *__restrict__ len, int *__restrict__ state,
                     int *__restrict__ node_state)
{
    const int lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * SWEEP_WAVES + (threadIdx.x >> 6);
    if (c >= nchains || !reject[c]) return;
    const long src = start_k[c], dst = start[c];
    const int m = cnt_k[c];
    for (int i = lane; i < m; i += 64) {
        edge[dst + i] = edge_k[src + i];
        len[dst + i] = len_k[src + i];
        state[dst + i] = state_k[src + i];
    }
    for (int v = lane; v < N; v += 64) node_state[c * N + v] = node_state_k[c * N + v];
    if (lane == 0) cnt[c] = m;
}

}  // namespace

struct rt_chains {
    rt_ctx *ctx = nullptr;
    int64_t nchains = 0, N = 0, n = 0;
    uint64_t seed = 0, nsweeps = 0;
    ell_matrix P;
    int *d_parent = nullptr;               // base tree
    double *d_branch = nullptr, *d_rates = nullptr, *d_root = nullptr;
    unsigned long long *d_node_masks = nullptr;
    // rows: buffer A holds the current histories, B the freshly split ones
    int64_t cap_rows = 0, cap_chunks = 0;
    int *d_edge_a = nullptr, *d_state_a = nullptr, *d_edge_b = nullptr, *d_row_chunk = nullptr;
    unsigned short *d_row_k = nullptr;     // events of each current row (count -> split)
    double *d_len_a = nullptr, *d_len_b = nullptr;
    long *d_start = nullptr, *d_newcnt = nullptr, *d_newstart = nullptr, *d_choff = nullptr;
    int *d_cnt = nullptr, *d_node_chunk = nullptr, *d_node_state = nullptr;
    int *d_cparent = nullptr, *d_cstate = nullptr, *d_status = nullptr, *d_flag = nullptr;
    unsigned long long *d_cmask = nullptr;
    double *d_L = nullptr;
    int64_t last_rows = 0, last_chunks = 0;
    // rt_chains_snapshot: the histories a Metropolis-Hastings step may return to
    int64_t cap_backup = 0;
    bool has_backup = false;
    uint64_t backup_step = 0;              // nsweeps when the snapshot was taken
    int *d_edge_k = nullptr, *d_state_k = nullptr, *d_cnt_k = nullptr, *d_node_state_k = nullptr;
    double *d_len_k = nullptr;
    long *d_start_k = nullptr;
    unsigned char *d_reject = nullptr;
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    ~rt_chains()
    {
        hipFree(d_edge_k); hipFree(d_state_k); hipFree(d_cnt_k); hipFree(d_node_state_k);
        hipFree(d_len_k); hipFree(d_start_k); hipFree(d_reject); hipFree(d_scan_tmp);
        hipFree(P.d_col); hipFree(P.d_val); hipFree(P.d_rowbits); hipFree(P.d_colbits);
        hipFree(P.d_dense);
        hipFree(d_parent); hipFree(d_branch); hipFree(d_rates); hipFree(d_root);
        hipFree(d_node_masks); hipFree(d_edge_a); hipFree(d_state_a); hipFree(d_edge_b);
        hipFree(d_row_chunk); hipFree(d_row_k); hipFree(d_len_a); hipFree(d_len_b); hipFree(d_start);
        hipFree(d_newcnt); hipFree(d_newstart); hipFree(d_choff); hipFree(d_cnt);
        hipFree(d_node_chunk); hipFree(d_node_state); hipFree(d_cparent); hipFree(d_cstate);
        hipFree(d_status); hipFree(d_flag); hipFree(d_cmask); hipFree(d_L);
    }
};

namespace {

template <class T>
int grow(T *&p, int64_t count, int64_t keep, hipStream_t st)
{
    T *q = nullptr;
    RT_HIP(hipMalloc((void **)&q, (size_t)std::max<int64_t>(count, 1) * sizeof(T)));
    if (p && keep > 0)
        RT_HIP(hipMemcpyAsync(q, p, (size_t)keep * sizeof(T), hipMemcpyDeviceToDevice, st));
    RT_HIP(hipStreamSynchronize(st));
    hipFree(p);
    p = q;
    return RT_OK;
}

unsigned sweep_grid(int64_t nchains) { return (unsigned)((nchains + SWEEP_WAVES - 1) / SWEEP_WAVES); }

// one sweep (per = 0) or one bisection attempt of the start-up (per = pieces per row);
// `commit` false leaves buffer A as it was when the chunk trees turn out infeasible
int chains_step(rt_chains *h, int per, int *flag_out)
{
    rt_ctx *ctx = h->ctx;
    hipStream_t st = ctx->stream;
    const int64_t C = h->nchains, N = h->N;
    const dim3 grid(sweep_grid(C)), block(64 * SWEEP_WAVES);
    const unsigned long long stream_events = 2 * h->nsweeps + 1, stream_states = 2 * h->nsweeps;
    RT_HIP(hipMemsetAsync(h->d_flag, 0, 2 * sizeof(int), st));      // [0] status, [1] row overflow
    hipLaunchKernelGGL(sweep_count_kernel, grid, block, 0, st, (long)C, h->d_start, h->d_cnt,
                       h->d_len_a, h->d_state_a, h->d_rates, per, (unsigned long long)h->seed,
                       stream_events, h->d_newcnt, h->d_row_k, h->d_flag + 1);
    // where each chain's new rows start: exclusive prefix sum over C + 1 counts (the last is 0,
    // so the last output is the total); rocPRIM's scan (a single-workgroup scan took 186 us at
    // 100 000 chains)
    {
        size_t bytes = h->scan_tmp_bytes;
        RT_HIP(hipcub::DeviceScan::ExclusiveSum(h->d_scan_tmp, bytes, h->d_newcnt, h->d_newstart,
                                                (int)(C + 1), st));
    }
    RT_HIP(hipGetLastError());
    long total_rows = 0;
    int overflow = 0;
    RT_HIP(hipMemcpyAsync(&total_rows, h->d_newstart + C, sizeof(long), hipMemcpyDeviceToHost, st));
    RT_HIP(hipMemcpyAsync(&overflow, h->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    if (overflow) {
        rt_set_error("a history row would carry %d or more virtual events in one sweep (Poisson "
                     "rate x segment length too large for the device sampler); nothing was "
                     "changed", SWEEP_MAX_GAPS);
        return RT_ERR_UNSUPPORTED;
    }
    const int64_t total_chunks = total_rows - C * (N - 2);
    RT_REQUIRE(total_rows < (1ll << 31) && total_chunks >= C, "row count out of range");
    if (total_rows > h->cap_rows) {
        const int64_t cap = total_rows + total_rows / 4 + 1024;
        RT_TRY(grow(h->d_edge_a, cap, h->cap_rows, st));
        RT_TRY(grow(h->d_len_a, cap, h->cap_rows, st));
        RT_TRY(grow(h->d_state_a, cap, h->cap_rows, st));
        RT_TRY(grow(h->d_row_k, cap, h->cap_rows, st));
        RT_TRY(grow(h->d_edge_b, cap, 0, st));
        RT_TRY(grow(h->d_len_b, cap, 0, st));
        RT_TRY(grow(h->d_row_chunk, cap, 0, st));
        h->cap_rows = cap;
    }
    if (total_chunks > h->cap_chunks) {
        const int64_t cap = total_chunks + total_chunks / 4 + 1024;
        RT_TRY(grow(h->d_cparent, cap, 0, st));
        RT_TRY(grow(h->d_cstate, cap, 0, st));
        RT_TRY(grow(h->d_cmask, cap, 0, st));
        RT_TRY(grow(h->d_L, cap * h->n, 0, st));
        h->cap_chunks = cap;
    }
    const size_t lds = (size_t)SWEEP_WAVES * N * 24;      // acc (8 B) + three int tables
    hipLaunchKernelGGL(sweep_split_kernel, grid, block, lds, st, (long)C, (int)N, (int)h->n,
                       h->d_parent, h->d_start, h->d_cnt, h->d_edge_a, h->d_len_a, h->d_state_a,
                       h->d_rates, per, (unsigned long long)h->seed, stream_events, h->d_row_k,
                       h->d_node_masks,
                       h->d_newstart, h->d_edge_b, h->d_len_b, h->d_row_chunk, h->d_choff,
                       h->d_cparent, h->d_cmask, h->d_node_chunk);
    launch_forest_sets((int)h->n, (long)C, h->d_choff, h->d_cparent, h->P.d_rowbits, h->P.d_colbits,
                       h->d_cmask, 1, st);
    launch_forest_pmap((int)h->n, (long)C, h->d_choff, h->d_cparent, h->P.width, h->P.d_col,
                       h->P.d_val, h->d_cmask, h->d_L, st);
    launch_forest_sample((int)h->n, (long)C, h->d_choff, h->d_cparent, h->P.d_dense, h->d_root,
                         h->d_L, (unsigned long long)h->seed, stream_states, h->d_cstate,
                         h->d_status, st);
    hipLaunchKernelGGL(sweep_flag_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st,
                       (long)C, h->d_status, h->d_flag);
    RT_HIP(hipGetLastError());
    int flag = 0;
    RT_HIP(hipMemcpyAsync(&flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    *flag_out = flag;
    ++h->nsweeps;
    if (flag != 0) return RT_OK;               // nothing committed: buffer A is untouched
    hipLaunchKernelGGL(sweep_merge_kernel, grid, block, 0, st, (long)C, (int)N, h->d_newstart,
                       h->d_choff, h->d_edge_b, h->d_len_b, h->d_row_chunk, h->d_cstate,
                       h->d_node_chunk, h->d_edge_a, h->d_len_a, h->d_state_a, h->d_start,
                       h->d_cnt, h->d_node_state);
    RT_HIP(hipGetLastError());
    h->last_rows = total_rows;
    h->last_chunks = total_chunks;
    return RT_OK;
}

}  // namespace

extern "C" int rt_chains_create(rt_ctx *ctx, int64_t nnodes, const int32_t *parent,
                                const double *branch_lengths, int64_t n, const double *P,
                                const double *poisson_rates, const double *root_distn,
                                int64_t nchains, const uint64_t *node_masks, uint64_t seed,
                                rt_chains **out)
{
    RT_REQUIRE(ctx && out, "null pointer");
    *out = nullptr;
    RT_REQUIRE(nnodes >= 2 && nnodes <= SWEEP_MAX_NODES, "the base tree needs 2..%d nodes",
               SWEEP_MAX_NODES);
    RT_REQUIRE(n >= 1 && n <= 64, "the forest passes hold a state per lane: n <= 64");
    RT_REQUIRE(parent && branch_lengths && P && poisson_rates && node_masks && nchains >= 1,
               "null array or no chains");
    RT_REQUIRE(nchains < (1ll << 23), "at most 2^23 chains per batch");
    RT_REQUIRE(parent[0] == -1, "node 0 must be the root (parent -1)");
    for (int64_t v = 1; v < nnodes; ++v) {
        RT_REQUIRE(parent[v] >= 0 && parent[v] < v, "parent %d of node %lld is not before it",
                   parent[v], (long long)v);
        RT_REQUIRE(branch_lengths[v] > 0.0, "branch length of node %lld is not positive",
                   (long long)v);
    }
    RT_HIP(hipSetDevice(ctx->device));
    rt_chains *h = new (std::nothrow) rt_chains();
    if (!h) return RT_ERR_NOMEM;
    h->ctx = ctx;
    h->nchains = nchains;
    h->N = nnodes;
    h->n = n;
    h->seed = seed;
    hipStream_t st = ctx->stream;
    const int64_t C = nchains, N = nnodes, E = N - 1;
    int rc = upload_matrix(n, P, h->P);
    auto fail = [&](int code) { delete h; return code; };
    if (rc != RT_OK) return fail(rc);
#define RT_CH(call) do { if ((call) != hipSuccess) { rt_set_error("HIP error in rt_chains_create: %s", hipGetErrorString(hipGetLastError())); return fail(RT_ERR_HIP); } } while (0)
    RT_CH(hipMalloc((void **)&h->d_parent, N * 4));
    RT_CH(hipMalloc((void **)&h->d_branch, N * 8));
    RT_CH(hipMalloc((void **)&h->d_rates, n * 8));
    RT_CH(hipMalloc((void **)&h->d_node_masks, C * N * 8));
    RT_CH(hipMalloc((void **)&h->d_start, C * sizeof(long)));
    RT_CH(hipMalloc((void **)&h->d_newcnt, (C + 1) * sizeof(long)));
    RT_CH(hipMemsetAsync(h->d_newcnt, 0, (C + 1) * sizeof(long), st));
    {
        size_t bytes = 0;
        RT_CH(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, h->d_newcnt, h->d_newcnt, (int)(C + 1), st));
        RT_CH(hipMalloc(&h->d_scan_tmp, std::max<size_t>(bytes, 16)));
        h->scan_tmp_bytes = bytes;
    }
    RT_CH(hipMalloc((void **)&h->d_newstart, (C + 1) * sizeof(long)));
    RT_CH(hipMalloc((void **)&h->d_choff, (C + 1) * sizeof(long)));
    RT_CH(hipMalloc((void **)&h->d_cnt, C * 4));
    RT_CH(hipMalloc((void **)&h->d_node_chunk, C * N * 4));
    RT_CH(hipMalloc((void **)&h->d_node_state, C * N * 4));
    RT_CH(hipMalloc((void **)&h->d_status, C * 4));
    RT_CH(hipMalloc((void **)&h->d_flag, 8));
    RT_CH(hipMemcpyAsync(h->d_parent, parent, N * 4, hipMemcpyHostToDevice, st));
    RT_CH(hipMemcpyAsync(h->d_branch, branch_lengths, N * 8, hipMemcpyHostToDevice, st));
    RT_CH(hipMemcpyAsync(h->d_rates, poisson_rates, n * 8, hipMemcpyHostToDevice, st));
    RT_CH(hipMemcpyAsync(h->d_node_masks, node_masks, C * N * 8, hipMemcpyHostToDevice, st));
    if (root_distn) {
        RT_CH(hipMalloc((void **)&h->d_root, n * 8));
        RT_CH(hipMemcpyAsync(h->d_root, root_distn, n * 8, hipMemcpyHostToDevice, st));
    }
    // start-up rows: one per edge and chain
    {
        std::vector<int> edge((size_t)(C * E)), state((size_t)(C * E), 0), cnt((size_t)C, (int)E);
        std::vector<double> len((size_t)(C * E));
        std::vector<long> start((size_t)C);
        for (int64_t c = 0; c < C; ++c) {
            start[(size_t)c] = (long)(c * E);
            for (int64_t v = 1; v < N; ++v) {
                edge[(size_t)(c * E + v - 1)] = (int)v;
                len[(size_t)(c * E + v - 1)] = branch_lengths[v];
            }
        }
        h->cap_rows = C * E;
        RT_CH(hipMalloc((void **)&h->d_edge_a, h->cap_rows * 4));
        RT_CH(hipMalloc((void **)&h->d_state_a, h->cap_rows * 4));
        RT_CH(hipMalloc((void **)&h->d_len_a, h->cap_rows * 8));
        RT_CH(hipMalloc((void **)&h->d_edge_b, h->cap_rows * 4));
        RT_CH(hipMalloc((void **)&h->d_len_b, h->cap_rows * 8));
        RT_CH(hipMalloc((void **)&h->d_row_chunk, h->cap_rows * 4));
        RT_CH(hipMalloc((void **)&h->d_row_k, h->cap_rows * sizeof(unsigned short)));
        RT_CH(hipMemcpyAsync(h->d_edge_a, edge.data(), edge.size() * 4, hipMemcpyHostToDevice, st));
        RT_CH(hipMemcpyAsync(h->d_state_a, state.data(), state.size() * 4, hipMemcpyHostToDevice, st));
        RT_CH(hipMemcpyAsync(h->d_len_a, len.data(), len.size() * 8, hipMemcpyHostToDevice, st));
        RT_CH(hipMemcpyAsync(h->d_start, start.data(), start.size() * sizeof(long), hipMemcpyHostToDevice, st));
        RT_CH(hipMemcpyAsync(h->d_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, st));
        RT_CH(hipStreamSynchronize(st));
    }
#undef RT_CH
    {
        const size_t lds = (size_t)SWEEP_WAVES * N * 24;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)sweep_split_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                rt_set_error("cannot reserve %zu bytes of LDS for the split kernel", lds);
                return fail(RT_ERR_HIP);
            }
        }
    }
    // a first feasible history: bisect the edges until every chunk tree has positive
    // likelihood (_sampler.py:563-648); more events per edge than states cannot help
    int flag = 1;
    for (int per = 1; flag != 0; per *= 2) {
        if (per > 1 && per - 1 > n) {
            rt_set_error("failed to find a feasible history for some chain (zero likelihood)");
            return fail(RT_ERR_ZERO_PROB);
        }
        rc = chains_step(h, per, &flag);
        if (rc != RT_OK) return fail(rc);
    }
    ctx->live_chains += 1;
    *out = h;
    return RT_OK;
}

extern "C" int rt_chains_sweep(rt_chains *h, int64_t nsweeps)
{
    RT_REQUIRE(h && nsweeps >= 0, "bad arguments");
    RT_HIP(hipSetDevice(h->ctx->device));
    for (int64_t i = 0; i < nsweeps; ++i) {
        int flag = 0;
        RT_TRY(chains_step(h, 0, &flag));
        if (flag != 0) {
            // the reference raises StructuralZeroProb / NumericalZeroProb here
            // (_sample_mc0_dense.py:57-62); nothing was committed, and the sweep does not count
            --h->nsweeps;
            rt_set_error("a chunk tree has zero likelihood (status %d: %s)", flag,
                         flag == 1 ? "at its root" : "below its root");
            return RT_ERR_ZERO_PROB;
        }
    }
    return RT_OK;
}

extern "C" int rt_chains_get_sizes(const rt_chains *h, int64_t *rows, int64_t *chunks,
                                   int64_t *sweeps)
{
    RT_REQUIRE(h, "null pointer");
    RT_HIP(hipSetDevice(h->ctx->device));
    RT_HIP(hipStreamSynchronize(h->ctx->stream));     // the last sweep's merge kernel
    if (rows) {
        std::vector<int> cnt((size_t)h->nchains);
        RT_HIP(hipMemcpy(cnt.data(), h->d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
        int64_t total = 0;
        for (int x : cnt) total += x;
        *rows = total;
    }
    if (chunks) *chunks = h->last_chunks;
    if (sweeps) *sweeps = (int64_t)h->nsweeps;
    return RT_OK;
}

extern "C" int rt_chains_get_statistics(rt_chains *h, double *dwell, int64_t *transitions,
                                        int32_t *node_states)
{
    RT_REQUIRE(h, "null pointer");
    RT_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int64_t C = h->nchains, n = h->n, N = h->N;
    double *d_dwell = nullptr;
    long long *d_trans = nullptr;
    struct release {                      // also on the error returns below
        double *&a;
        long long *&b;
        ~release() { hipFree(a); hipFree(b); }
    } guard{d_dwell, d_trans};
    if (dwell) RT_HIP(hipMalloc((void **)&d_dwell, C * n * 8));
    if (transitions) {
        RT_HIP(hipMalloc((void **)&d_trans, C * n * n * 8));
        RT_HIP(hipMemsetAsync(d_trans, 0, C * n * n * 8, st));
    }
    if (dwell || transitions) {
        hipLaunchKernelGGL(sweep_stats_kernel, dim3(sweep_grid(C)), dim3(64 * SWEEP_WAVES), 0, st,
                           (long)C, (int)n, h->d_start, h->d_cnt, h->d_edge_a, h->d_len_a,
                           h->d_state_a, d_dwell, d_trans);
        RT_HIP(hipGetLastError());
    }
    if (dwell) RT_HIP(hipMemcpyAsync(dwell, d_dwell, C * n * 8, hipMemcpyDeviceToHost, st));
    if (transitions)
        RT_HIP(hipMemcpyAsync(transitions, d_trans, C * n * n * 8, hipMemcpyDeviceToHost, st));
    if (node_states)
        RT_HIP(hipMemcpyAsync(node_states, h->d_node_state, C * N * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
}

extern "C" int rt_chains_get_rows(rt_chains *h, int64_t capacity, int64_t *chain_offset,
                                  int32_t *edge, double *length, int32_t *state)
{
    RT_REQUIRE(h && chain_offset && edge && length && state, "null pointer");
    RT_HIP(hipSetDevice(h->ctx->device));
    RT_HIP(hipStreamSynchronize(h->ctx->stream));     // the last sweep's merge kernel
    const int64_t C = h->nchains;
    std::vector<long> start((size_t)C);
    std::vector<int> cnt((size_t)C);
    RT_HIP(hipMemcpy(start.data(), h->d_start, (size_t)C * sizeof(long), hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(cnt.data(), h->d_cnt, (size_t)C * 4, hipMemcpyDeviceToHost));
    int64_t total = 0;
    for (int64_t c = 0; c < C; ++c) {
        chain_offset[c] = total;
        total += cnt[(size_t)c];
    }
    chain_offset[C] = total;
    RT_REQUIRE(total <= capacity, "the batch holds %lld rows, the arrays %lld", (long long)total,
               (long long)capacity);
    // the chains' runs are not contiguous on the device (each keeps the room of its split rows)
    const int64_t span = h->cap_rows;
    std::vector<int> e((size_t)span), s((size_t)span);
    std::vector<double> l((size_t)span);
    RT_HIP(hipMemcpy(e.data(), h->d_edge_a, (size_t)span * 4, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(s.data(), h->d_state_a, (size_t)span * 4, hipMemcpyDeviceToHost));
    RT_HIP(hipMemcpy(l.data(), h->d_len_a, (size_t)span * 8, hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < C; ++c)
        for (int i = 0; i < cnt[(size_t)c]; ++i) {
            const size_t src = (size_t)(start[(size_t)c] + i), dst = (size_t)(chain_offset[c] + i);
            edge[dst] = e[src];
            length[dst] = l[src];
            state[dst] = s[src];
        }
    return RT_OK;
}

extern "C" int rt_chains_snapshot(rt_chains *h)
{
    RT_REQUIRE(h, "null pointer");
    RT_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int64_t C = h->nchains, N = h->N;
    if (!h->d_start_k) {
        RT_HIP(hipMalloc((void **)&h->d_start_k, C * sizeof(long)));
        RT_HIP(hipMalloc((void **)&h->d_cnt_k, C * 4));
        RT_HIP(hipMalloc((void **)&h->d_node_state_k, C * N * 4));
        RT_HIP(hipMalloc((void **)&h->d_reject, C));
    }
    if (h->cap_backup < h->cap_rows) {
        RT_HIP(hipStreamSynchronize(st));
        hipFree(h->d_edge_k); hipFree(h->d_state_k); hipFree(h->d_len_k);
        h->d_edge_k = h->d_state_k = nullptr;
        h->d_len_k = nullptr;
        RT_HIP(hipMalloc((void **)&h->d_edge_k, h->cap_rows * 4));
        RT_HIP(hipMalloc((void **)&h->d_state_k, h->cap_rows * 4));
        RT_HIP(hipMalloc((void **)&h->d_len_k, h->cap_rows * 8));
        h->cap_backup = h->cap_rows;
    }
    const int64_t span = h->cap_rows;
    RT_HIP(hipMemcpyAsync(h->d_edge_k, h->d_edge_a, span * 4, hipMemcpyDeviceToDevice, st));
    RT_HIP(hipMemcpyAsync(h->d_state_k, h->d_state_a, span * 4, hipMemcpyDeviceToDevice, st));
    RT_HIP(hipMemcpyAsync(h->d_len_k, h->d_len_a, span * 8, hipMemcpyDeviceToDevice, st));
    RT_HIP(hipMemcpyAsync(h->d_start_k, h->d_start, C * sizeof(long), hipMemcpyDeviceToDevice, st));
    RT_HIP(hipMemcpyAsync(h->d_cnt_k, h->d_cnt, C * 4, hipMemcpyDeviceToDevice, st));
    RT_HIP(hipMemcpyAsync(h->d_node_state_k, h->d_node_state, C * N * 4, hipMemcpyDeviceToDevice, st));
    h->has_backup = true;
    h->backup_step = h->nsweeps;
    return RT_OK;
}

extern "C" int rt_chains_restore(rt_chains *h, const uint8_t *reject)
{
    RT_REQUIRE(h && reject, "null pointer");
    RT_REQUIRE(h->has_backup, "no snapshot to return to");
    // a chain's run has room for its snapshot only right after the sweep that followed it
    // (the run keeps the room of that sweep's split rows, which include every snapshot row)
    RT_REQUIRE(h->nsweeps == h->backup_step + 1,
               "rt_chains_restore undoes exactly one sweep (%llu since the snapshot)",
               (unsigned long long)(h->nsweeps - h->backup_step));
    RT_HIP(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int64_t C = h->nchains;
    RT_HIP(hipMemcpyAsync(h->d_reject, reject, C, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sweep_restore_kernel, dim3(sweep_grid(C)), dim3(64 * SWEEP_WAVES), 0, st,
                       (long)C, (int)h->N, h->d_reject, h->d_start_k, h->d_cnt_k, h->d_edge_k,
                       h->d_len_k, h->d_state_k, h->d_node_state_k, h->d_start, h->d_cnt,
                       h->d_edge_a, h->d_len_a, h->d_state_a, h->d_node_state);
    RT_HIP(hipGetLastError());
    RT_HIP(hipStreamSynchronize(st));           // `reject` is the caller's buffer
    return RT_OK;
}

extern "C" int rt_chains_destroy(rt_chains *h)
{
    if (!h) return RT_OK;
    h->ctx->live_chains -= 1;
    hipSetDevice(h->ctx->device);
    hipStreamSynchronize(h->ctx->stream);
    delete h;
    return RT_OK;
}
