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
    for (long v = hi - 1; v > lo; --v) {
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
    hipLaunchKernelGGL(forest_sets_kernel, dim3(forest_grid(ntrees)), dim3(64 * FOREST_WAVES), 0, st,
                       (int)n, (long)ntrees, f.d_off, f.d_parent, f.P.d_rowbits, f.P.d_colbits,
                       (unsigned long long *)f.d_allowed, 1);
    RT_HIP(hipGetLastError());
    if (subtree_probability) {
        RT_HIP(hipMalloc((void **)&f.d_L, total * n * 8));
        hipLaunchKernelGGL(forest_pmap_kernel, dim3(forest_grid(ntrees)), dim3(64 * FOREST_WAVES), 0,
                           st, (int)n, (long)ntrees, f.d_off, f.d_parent, f.P.width, f.P.d_col,
                           f.P.d_val, (const unsigned long long *)f.d_allowed, f.d_L);
        RT_HIP(hipGetLastError());
        RT_HIP(hipMemcpyAsync(subtree_probability, f.d_L, total * n * 8, hipMemcpyDeviceToHost, st));
    }
    RT_HIP(hipMemcpyAsync(allowed_sets, f.d_allowed, total * 8, hipMemcpyDeviceToHost, st));
    RT_HIP(hipStreamSynchronize(st));
    return RT_OK;
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
    const dim3 grid(forest_grid(ntrees)), block(64 * FOREST_WAVES);
    // one sweep's worth of passes back to back on the stream: sets, pmap, sampling
    hipLaunchKernelGGL(forest_sets_kernel, grid, block, 0, st, (int)n, (long)ntrees, f.d_off,
                       f.d_parent, f.P.d_rowbits, f.P.d_colbits, (unsigned long long *)f.d_allowed, 1);
    hipLaunchKernelGGL(forest_pmap_kernel, grid, block, 0, st, (int)n, (long)ntrees, f.d_off,
                       f.d_parent, f.P.width, f.P.d_col, f.P.d_val,
                       (const unsigned long long *)f.d_allowed, f.d_L);
    hipLaunchKernelGGL(forest_sample_kernel, grid, block, 0, st, (int)n, (long)ntrees, f.d_off,
                       f.d_parent, f.P.d_dense, f.d_root, f.d_L, (unsigned long long)seed,
                       (unsigned long long)sweep, f.d_states, f.d_status);
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
