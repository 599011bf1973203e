// Felsenstein upward pass (leaf -> root), batched over independent sites.
//
// Replaces pyfelscore.mcy_esd_get_node_to_pmap + _mc0_dense.get_likelihood of
// the reference (raoteh/sampler/_mcy_dense.py:286, _mc0_dense.py:147-212) for
// the log-likelihood hot path:
//
//   L[v,s] = obs[v,s] * prod_{c child of v} sum_{s'} P_c[s,s'] * L[c,s']
//   lik    = sum_s w[s] * max(L[root,s], 0);  loglik = log(lik)
//
// Three kernel families, all f64, all driven by the same post-order schedule
// (rt_op, common.h):
//
//   prune_lane_kernel   n <= 4   one lane = one site.  Leaf vectors stream
//                       HBM -> LDS through an LDS-DMA ring (global_load_lds,
//                       1 KiB per wave-instruction, R slots in flight per wave),
//                       P_e comes through the scalar cache as FMA operands,
//                       pending accumulators live in a register stack.
//                       HBM-bound (config 2).
//   prune_mfma_kernel   4 < n <= 128  one wave = 16 sites, one workgroup = 64
//                       sites (n <= 64; above that NT = 5..8 waves share ONE tile).  t = P_e * L as v_mfma_f64_16x16x4_f64 tiles:
//                       the D tile of one edge IS the B operand of the next
//                       edge (same lane/register map), so messages never leave
//                       registers.  P_e is staged once per workgroup in LDS
//                       (LDS-DMA, double buffered), leaf vectors are loaded in
//                       B-operand order straight from HBM one step ahead.
//   prune_generic_kernel  any n <= 128, any stack depth: one lane = one site,
//                       accumulators in a global scratch stack.  Fallback only.
#include "common.h"
#include "reduce.h"

#include <algorithm>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define RT_CONST_AS __attribute__((address_space(4)))

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef int int4_t __attribute__((ext_vector_type(4)));

// schedule entries are wave-uniform: read them through the scalar cache
__device__ __forceinline__ rt_op load_op(const RT_CONST_AS int4_t *ops, int i)
{
    const int4_t o = ops[i];
    rt_op op;
    op.node = o.x;
    op.obs = o.y;
    op.pop = o.z;
    op.dst = o.w;
    return op;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// escale: the site's messages were multiplied by 2^-escale on the way up (opt-in rescaling:
// exact powers of two, so the mantissa of the likelihood is what it would have been)
__device__ __forceinline__ void finish_site(double lik, bool negative,
                                            bool valid, double *loglik,
                                            int *status, long site,
                                            double &sum, double &nzero, int escale = 0)
{
    const bool ok = lik > 0.0;
    if (valid) {
        double ll = ok ? log(lik) : -INFINITY;
        if (ok && escale != 0) {
            // in range: undo the scaling exactly and take the logarithm of the true value;
            // out of range: log(m 2^e) = log(m) + e ln 2
            const double back = ldexp(lik, escale);
            ll = (back > 0x1p-1000 && back < 0x1p1000) ? log(back)
                                                       : log(lik) + (double)escale * 0.6931471805599453094;
        }
        loglik[site] = ll;
        status[site] = (ok ? RT_SITE_OK : RT_SITE_ZERO_PROB) |
                       (negative ? RT_SITE_NEGATIVE : 0);
        sum = ok ? ll : 0.0;
        nzero = ok ? 0.0 : 1.0;
    } else {
        sum = 0.0;
        nzero = 0.0;
    }
}

// Opt-in rescaling ("rescale", interpreter kernels): when the largest entry of a site's
// message drops below 2^-256 the message is multiplied by the power of two that brings it
// back to [1, 2) and the exponent is remembered per site.  Powers of two are exact: a batch
// that never comes near the threshold gets the numbers of the plain kernels bit for bit, and
// a 1 000-leaf tree (e^-6000: zero in f64 without this; the reference has no rescaling
// either) gets its log-likelihood.  -> the exponent to ADD to the site's tally (<= 0), 0: none
__device__ __forceinline__ int rescale_exponent(double mx)
{
    if (!(mx < 0x1p-256) || !(mx > 0.0)) return 0;
    int e;
    frexp(mx, &e);              // mx = f 2^e, f in [0.5, 1)
    return e - 1;               // 2^-(e-1) mx in [1, 2)
}

// ---------------------------------------------------------------------------
// How the fast kernels walk the tree
// ---------------------------------------------------------------------------
//
// One flat loop over the post-order schedule (rt_op).  Pending accumulators
// live in an LDS stack indexed by the slot numbers the host assigned
// (build_schedule, api.hip): with its Sethi-Ullman child order the stack needs
// at most floor(log2(#leaves)) + 1 slots for ANY tree shape.  The LDS stack is
// laid out [slot][component][lane], so every access is a conflict-free
// ds_read_b64 / ds_write_b64 and no register array is ever indexed at run time.
// Every branch on the schedule is wave-uniform (scalar).

#define RT_OP_IS_INTERNAL(op) ((op).pop >= 0)
#define RT_OP_IS_ROOT(op) ((op).dst < 0)
#define RT_OP_IS_FIRST(op) (((op).dst >> 8) != 0)

// ---------------------------------------------------------------------------
// n <= 4: one wave per workgroup, lane per site
// ---------------------------------------------------------------------------
//
// HBM -> VGPR streaming, no LDS staging: the observation stream of a wave (the
// leaf vectors of its 64 sites, in schedule order) is consumed strictly in
// order whatever the tree shape, so the kernel is an outer loop over the
// stream, unrolled R times, with a register ring of R slots whose indices are
// all static; between two stream elements an inner loop runs the steps that
// carry no observation (internal nodes).  Each slot is NP/2 fully coalesced
// 16-byte loads per lane (layout [block][slot][pair][lane][2]), R - 1 slots
// ((R - 1) * NP * 512 B per wave) stay in flight, and the compiler's own
// counted vmcnt waits retire them in order.  P_e arrives through the scalar
// cache as FMA operands; pending accumulators sit in a conflict-free LDS stack
// ([slot][state][lane], 512*N bytes per slot).

// Lane-kernel step encoding (built on the host by rt_lane_program, api.hip).
// The top accumulator is cached in registers ("cur"); the host simulates that
// cache and tells every step where its operands live:
//   x = flags & LOP_INTERNAL ? (flags & LOP_X_CUR ? cur : lds[pop_off]) : 1
//   t = P * (x * obs)
//   FIRST: (SPILL ? lds[spill_off] = cur : -) ; cur = t
//   else : DST_CUR ? cur *= t : lds[dst_off] *= t
// In a cherry (two leaves + parent) nothing touches LDS at all.
enum {
    LOP_INTERNAL = 1, LOP_X_CUR = 2, LOP_FIRST = 4, LOP_ROOT = 8, LOP_SPILL = 16,
    LOP_DST_CUR = 32, LOP_OBS = 64,
    // fast path (set by the host for the common steps): x is the observation
    // (leaf) or the register cache (internal node), and the result stays in the
    // register cache: cur = t * {1 | cur | lds[dst_off]}
    LOP_FAST = 128,
    // fused step of the LDS-DMA lane kernel: two observed leaves a, b and their
    // parent p in one go, t = P_p * ((P_a o_a) * (P_b o_b)); three consecutive P
    // records; spill_off as for a first child, dst flags/offset are the parent's
    LOP_CHERRY = 256,
    LOP_STOP = 512      // sentinel entry after the last step (lane kernels)
};

// B = site blocks per wave: B = 2 gives every lane two independent sites, which
// (i) halves the scalar work, the schedule fetches and the P reads per site and
// (ii) gives the scheduler two independent dependency chains to interleave.
template <int N, bool PLDS, int B = 1, bool RESC = false>
struct LaneCtx {
    const RT_CONST_AS int4_t *ops_c;
    const RT_CONST_AS double *P_c;       // step-ordered P through the scalar cache
    const double *P_l;                   // ... or the copy of it in LDS (PLDS)
    const RT_CONST_AS double *w_c;
    unsigned char *stack[B];       // LDS, + lane * 8 bytes
    int nops;                      // program entries
    int nrec;                      // P records (>= nops: a fused entry spans 3)
    int pidx;                      // P record of the current step
    int i;                         // index of the current program entry
    int4_t op;                     // current step: {flags, pop_off, dst_off, spill_off}
    double p[N * N];               // transition matrix of the current step
    double cur[B][N];              // register-cached top accumulator
    double lik[B];
    bool negative[B];
    int escale[B];                 // RESC: exponent tally of the lane's site

    // RESC: t = P x is about to be deposited; x's largest entry decides (rescale_exponent)
    __device__ __forceinline__ void rescale_result(const double (&x)[B][N], double (&t)[B][N])
    {
        if constexpr (RESC) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                double mx = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) mx = fmax(mx, x[b][j]);
                const int e = rescale_exponent(mx);
                if (e != 0) {
#pragma unroll
                    for (int r = 0; r < N; ++r) t[b][r] = ldexp(t[b][r], -e);
                    escale[b] += e;
                }
            }
        }
    }

    __device__ __forceinline__ void init()
    {
        pidx = 0;
        i = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            escale[b] = 0;
            lik[b] = 0.0;
            negative[b] = false;
#pragma unroll
            for (int j = 0; j < N; ++j) cur[b][j] = 1.0;
        }
    }

    __device__ __forceinline__ void load_p(int ii)
    {
        if (PLDS) {
            // wave-uniform address: every ds_read is a broadcast
#pragma unroll
            for (int j = 0; j < N * N; ++j) p[j] = P_l[ii * N * N + j];
        } else {
#pragma unroll
            for (int j = 0; j < N * N; ++j) p[j] = P_c[(long)ii * N * N + j];
        }
    }

    // The program carries a STOP sentinel after its last entry and the P table
    // one extra (zero) record, so fetching "the next" entry / matrix never needs
    // a bounds clamp.
    __device__ __forceinline__ void load_current()
    {
        op = ops_c[i];
        load_p(pidx);
    }

    __device__ __forceinline__ double lds_get(int b, int off, int j) const
    {
        return *(const double *)(stack[b] + off + j * 512);
    }
    __device__ __forceinline__ void lds_put(int b, int off, int j, double v)
    {
        *(double *)(stack[b] + off + j * 512) = v;
    }

    // t = P * x for every block
    __device__ __forceinline__ void matvec(const double (&q)[N * N],
                                           const double (&x)[B][N], double (&t)[B][N])
    {
#pragma unroll
        for (int r = 0; r < N; ++r) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                double sacc = q[r * N] * x[b][0];
#pragma unroll
                for (int j = 1; j < N; ++j) sacc = fma(q[r * N + j], x[b][j], sacc);
                t[b][r] = sacc;
            }
        }
    }

    // where the result of a step goes (flags are wave-uniform)
    __device__ __forceinline__ void deposit(int flags, const double (&t)[B][N],
                                            bool unspill_is_fast)
    {
        if (flags & LOP_FIRST) {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int r = 0; r < N; ++r) cur[b][r] = t[b][r];
        } else if (flags & LOP_DST_CUR) {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int r = 0; r < N; ++r) cur[b][r] *= t[b][r];
        } else if (unspill_is_fast) {      // un-spill: the parent's step is next
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int r = 0; r < N; ++r) cur[b][r] = lds_get(b, op.z, r) * t[b][r];
        } else {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int r = 0; r < N; ++r)
                    lds_put(b, op.z, r, lds_get(b, op.z, r) * t[b][r]);
        }
    }

    __device__ __forceinline__ void spill()
    {
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int r = 0; r < N; ++r) lds_put(b, op.w, r, cur[b][r]);
    }

    // Execute the current step with observation vector o (ignored unless
    // HAS_OBS) and move to the next step.
    template <bool HAS_OBS>
    __device__ __forceinline__ void compute(const double (&o)[B][N])
    {
        const int flags = op.x;
        double t[B][N];
        if (flags & LOP_FAST) {
            if (flags & LOP_SPILL) spill();
            if constexpr (HAS_OBS) { matvec(p, o, t); rescale_result(o, t); }
            else { matvec(p, cur, t); rescale_result(cur, t); }
            deposit(flags, t, true);
            return;
        }
        double x[B][N];
        if (flags & LOP_INTERNAL) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (flags & LOP_X_CUR) {
#pragma unroll
                    for (int j = 0; j < N; ++j) x[b][j] = cur[b][j];
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) x[b][j] = lds_get(b, op.y, j);
                }
                if (HAS_OBS) {
#pragma unroll
                    for (int j = 0; j < N; ++j) x[b][j] *= o[b][j];
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int j = 0; j < N; ++j) x[b][j] = HAS_OBS ? o[b][j] : 1.0;
        }
        if (flags & LOP_ROOT) {
            // root reduction (_mc0_dense.py:184-209)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                double sacc = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    negative[b] |= x[b][j] < 0.0;
                    sacc += w_c[j] * fmax(x[b][j], 0.0);
                }
                lik[b] = sacc;
            }
        } else {
            matvec(p, x, t);
            rescale_result(x, t);
            if ((flags & LOP_FIRST) && (flags & LOP_SPILL)) spill();
            deposit(flags, t, false);
        }
    }

    // Fetch ahead (call after compute() and after anything that must not wait
    // for these requests, e.g. the ring refill).  p is a single buffer: the FMAs
    // of compute() have read it, so the next step's matrix can be requested now.
    __device__ __forceinline__ void advance(int records = 1)
    {
        i += 1;
        pidx += records;
        op = ops_c[i];
        load_p(pidx);
    }

    // Fused cherry (LOP_CHERRY, PLDS only): p already holds P_a.
    __device__ __forceinline__ void compute_cherry(const double (&oa)[B][N],
                                                   const double (&ob)[B][N])
    {
        const int flags = op.x;
        double pb[N * N], pp[N * N];
        const double *rec = P_l + (pidx + 1) * N * N;
#pragma unroll
        for (int j = 0; j < N * N; ++j) pb[j] = rec[j];
#pragma unroll
        for (int j = 0; j < N * N; ++j) pp[j] = rec[N * N + j];
        if (flags & LOP_SPILL) spill();
        double ya[B][N], yb[B][N], t[B][N];
        matvec(p, oa, ya);
        matvec(pb, ob, yb);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int r = 0; r < N; ++r) ya[b][r] *= yb[b][r];
        matvec(pp, ya, t);
        deposit(flags, t, (flags & LOP_FAST) != 0);
    }

    template <bool HAS_OBS>
    __device__ __forceinline__ void step(const double (&o)[B][N])
    {
        compute<HAS_OBS>(o);
        advance();
    }

    // run the steps that carry no observation
    __device__ __forceinline__ void run_plain()
    {
        const double none[B][N] = {};
        while (!(op.x & (LOP_OBS | LOP_STOP))) step<false>(none);
    }
};

// PLDS = true: four waves per workgroup share one copy of the whole step-ordered
// P table in LDS (one barrier at kernel start, none afterwards); the scalar
// cache cannot hold P for a 64-leaf tree (16 KB + schedule), and an L2 round
// trip per step is what the wave then waits for.  PLDS = false: one wave per
// workgroup, P through the scalar cache (trees whose P table does not fit LDS).
template <int N, int R, bool PLDS, bool RESC = false>
__global__ void __launch_bounds__(PLDS ? 256 : 64)
prune_lane_kernel(const double *__restrict__ Pord,   // [nops][N][N]
                  const int4_t *__restrict__ ops, int nops,   // lane program
                  const double *__restrict__ obs, int K,  // [blk][K][NP/2][64][2]
                  const double *__restrict__ root_w, int depth_arg,
                  double *__restrict__ loglik, int *__restrict__ status,
                  double *__restrict__ partial, long nsites, long nblocks)
{
    constexpr int NP = (N + 1) & ~1;
    constexpr int HP = NP / 2;                // 16-byte pairs per site
    // timing experiment only (RAOTEH_LANE_NOLOAD): skip the HBM stream
    const bool noload = depth_arg < 0;
    const int depth = noload ? -depth_arg : depth_arg;
    constexpr int WPB = PLDS ? 4 : 1;         // waves per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long gw = (long)blockIdx.x * WPB + wave;    // site block of this wave

    LaneCtx<N, PLDS, 1, RESC> C;
    unsigned char *stack_base = smem;
    if (PLDS) {
        double *pl = (double *)smem;
        for (int e = threadIdx.x; e < (nops + 1) * N * N; e += 256) pl[e] = Pord[e];
        stack_base = smem + (((size_t)(nops + 1) * N * N * 8 + 15) & ~(size_t)15);
        __syncthreads();
        if (gw >= nblocks) return;            // wave-uniform, after the only barrier
        C.P_l = pl;
    } else {
        C.P_l = nullptr;
    }
    const double2 *g = (const double2 *)obs + (size_t)gw * K * HP * 64 + lane;
    C.ops_c = (const RT_CONST_AS int4_t *)ops;
    C.P_c = (const RT_CONST_AS double *)Pord;
    C.w_c = (const RT_CONST_AS double *)root_w;
    C.stack[0] = stack_base + (size_t)wave * depth * N * 512 + lane * 8;
    C.nops = nops;
    C.nrec = nops;
    C.init();

    // prologue: R slots in flight
    double2 ring[R][HP];
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
        for (int h = 0; h < HP; ++h) ring[j][h] = make_double2(1.0, 1.0);
        if (j < K && !noload) {
#pragma unroll
            for (int h = 0; h < HP; ++h) ring[j][h] = g[((size_t)j * HP + h) * 64];
        }
    }

    C.load_current();
    C.run_plain();
    for (int k0 = 0; k0 < K; k0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int k = k0 + j;
            if (k < K) {              // wave-uniform; the current step consumes slot k
                double o[1][N];
#pragma unroll
                for (int q = 0; q < N; ++q)
                    o[0][q] = (q & 1) ? ring[j][q >> 1].y : ring[j][q >> 1].x;
                C.template step<true>(o);
                if (k + R < K && !noload) {
#pragma unroll
                    for (int h = 0; h < HP; ++h)
                        ring[j][h] = g[((size_t)(k + R) * HP + h) * 64];
                }
                C.run_plain();
            }
        }
    }

    const long site = gw * 64 + lane;
    double sum, nzero;
    finish_site(C.lik[0], C.negative[0], site < nsites, loglik, status, site, sum, nzero,
                C.escale[0]);
    sum = wave_sum(sum);
    nzero = wave_sum(nzero);
    if (lane == 0) {
        partial[gw * 2] = sum;
        partial[gw * 2 + 1] = nzero;
    }
}

// ---------------------------------------------------------------------------
// n <= 4, variant B: the same walk (LaneCtx) with the leaf vectors landing in an
// LDS ring by LDS-DMA (global_load_lds, 1 KiB per wave-instruction, data layout
// [block][slot][pair][lane][2], the same as the VGPR-ring variant).  Data in flight lives in LDS, not in VGPRs, so the
// kernel owns its vmcnt waits: exactly the slots younger than the one being
// consumed stay outstanding ((R-1)*IPS instructions), nothing is ever drained.
// ---------------------------------------------------------------------------

// B site blocks per wave (see LaneCtx), WPB waves per workgroup.
// cache policy of the leaf-vector stream: non-temporal (read exactly once)
#define RT_AUX_NT 2

template <int N, int R, int B, int WPB>
__global__ void __launch_bounds__(64 * WPB)
prune_lanedma_kernel(const double *__restrict__ Pord,   // [nrec][N][N]
                     const int4_t *__restrict__ ops, int nops, int nrec,   // program
                     const double *__restrict__ obs, int K,  // [blk][K][NP/2][64][2]
                     const double *__restrict__ root_w, int depth,
                     double *__restrict__ loglik, int *__restrict__ status,
                     double *__restrict__ partial, long nsites, long nwaves)
{
    constexpr int NP = (N + 1) & ~1;
    constexpr int SLOT = 64 * NP * 8;         // bytes of one obs slot of one block
    constexpr int IPS = SLOT / 1024;          // LDS-DMA instructions per slot
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long gw = (long)blockIdx.x * WPB + wave;    // wave = B consecutive site blocks

    LaneCtx<N, true, B> C;
    double *pl = (double *)smem;
    for (int e = threadIdx.x; e < (nrec + 1) * N * N; e += 64 * WPB) pl[e] = Pord[e];
    const int stack_bytes = depth * N * 512;
    // per wave: [B][R] observation slots, then [B] accumulator stacks
    unsigned char *ring = smem + (((size_t)(nrec + 1) * N * N * 8 + 15) & ~(size_t)15) +
                          (size_t)wave * B * (R * SLOT + stack_bytes);
    __syncthreads();
    if (gw >= nwaves) return;                 // wave-uniform, after the only barrier
    C.P_l = pl;
    C.ops_c = (const RT_CONST_AS int4_t *)ops;
    C.P_c = (const RT_CONST_AS double *)Pord;
    C.w_c = (const RT_CONST_AS double *)root_w;
#pragma unroll
    for (int b = 0; b < B; ++b) C.stack[b] = ring + B * R * SLOT + b * stack_bytes + lane * 8;
    C.nops = nops;
    C.nrec = nrec;
    C.init();

    const size_t bstride = (size_t)K * SLOT;  // bytes between consecutive blocks
    const unsigned char *g =
        (const unsigned char *)obs + (size_t)gw * B * bstride + lane * 16;
    // stream position kk of every block -> ring slot r
    auto fetch = [&](int kk, int r) {
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < IPS; ++j)
                __builtin_amdgcn_global_load_lds(
                    (glb_void *)(g + b * bstride + (size_t)kk * SLOT + j * 1024),
                    (lds_void *)(ring + (b * R + r) * SLOT + j * 1024), 16, 0, RT_AUX_NT);
    };
    // prologue: fill the ring
#pragma unroll
    for (int k = 0; k < R; ++k)
        if (k < K) fetch(k, k);

    C.load_current();
    C.run_plain();
    int rs = 0;                               // ring slot of stream position k
    int k = 0;
    while (k < K) {
        if ((C.op.x & LOP_CHERRY) && R >= 2) {
            // two leaves + parent: stream positions k, k+1
            if (k + 1 + R <= K)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R >= 2 ? R - 2 : 0) * IPS * B) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int rs1 = (rs + 1 == R) ? 0 : rs + 1;
            double oa[B][N], ob[B][N];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                // slot image [pair][lane][2]: conflict-free ds_read_b128
                const double *oa_l = (const double *)(ring + (b * R + rs) * SLOT) + lane * 2;
                const double *ob_l = (const double *)(ring + (b * R + rs1) * SLOT) + lane * 2;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    oa[b][j] = oa_l[(j >> 1) * 128 + (j & 1)];
                    ob[b][j] = ob_l[(j >> 1) * 128 + (j & 1)];
                }
            }
            C.compute_cherry(oa, ob);
            if (k + R < K) fetch(k + R, rs);
            if (k + 1 + R < K) fetch(k + 1 + R, rs1);
            C.advance(3);
            rs = (rs1 + 1 == R) ? 0 : rs1 + 1;
            k += 2;
        } else {
            if (k + R <= K)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * IPS * B) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            double o[B][N];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const double *o_l = (const double *)(ring + (b * R + rs) * SLOT) + lane * 2;
#pragma unroll
                for (int j = 0; j < N; ++j) o[b][j] = o_l[(j >> 1) * 128 + (j & 1)];
            }
            C.template compute<true>(o);
            // refill BEFORE the fetch-ahead: the LDS-DMA must wait for the reads
            // of the slot it overwrites (lgkmcnt(0)), which are long done here,
            // and must not wait for the P / schedule requests advance() issues
            if (k + R < K) fetch(k + R, rs);
            C.advance();
            rs = (rs + 1 == R) ? 0 : rs + 1;
            k += 1;
        }
        C.run_plain();
    }

#pragma unroll
    for (int b = 0; b < B; ++b) {
        const long blk = gw * B + b;
        const long site = blk * 64 + lane;
        double sum, nzero;
        finish_site(C.lik[b], C.negative[b], site < nsites, loglik, status, site, sum, nzero);
        sum = wave_sum(sum);
        nzero = wave_sum(nzero);
        if (lane == 0) {
            partial[blk * 2] = sum;
            partial[blk * 2 + 1] = nzero;
        }
    }
}

// ---------------------------------------------------------------------------
// 4 < n <= 128: v_mfma_f64_16x16x4_f64
// ---------------------------------------------------------------------------
//
// Register maps (guide: f64 MFMA 16x16x4): A lane l holds A[l&15][l>>4], B lane
// l holds B[l>>4][l&15], D lane l reg r holds D[(l>>4) + 4r][l&15].
//
// A tile of 16 sites is owned by NT waves (NT = ceil(n/16) row tiles); wave m
// computes rows 16m..16m+15 of t = P_e * x: KS = ceil(n/4) MFMAs with
// A = its own 16 x 4 slices of P_e (loaded from HBM/L2 straight into registers,
// one step ahead, in fragment order) and B = x (k-step kk = states 4kk..4kk+3
// of the 16 sites).  The D registers of wave m are states 16m + 4r + (l>>4),
// i.e. exactly rows 4m..4m+3 of the B operand of the next edge, so the waves of
// a tile publish their four rows of x to an LDS exchange buffer (double
// buffered, one workgroup barrier per step) and keep only their own rows of
// the pending accumulators (LDS stack, 2 KB per slot per wave).  A workgroup is
// 4 waves (3 when NT = 3) = 4/NT site tiles; nothing but x crosses waves.

// Root halves of the interpreter kernel (the cut of jit.hip's split_at_root, for trees that
// have no tree-specialised kernel yet): even workgroups run the program of the subtrees of
// all children of the root but the last, odd ones the program of the last child's subtree;
// each ends at ITS root step by storing its share of the root's accumulator (own rows) to
// halfbuf[tile][half][k-step][lane]; prune_mfma_combine_kernel finishes the sites.  The
// second program's steps are records rec1.. of the P table and its leaves positions kobs1..
// of the observation stream (both programs are contiguous runs of the post-order schedule).
struct rt_interp_halves {
    const int4_t *prog1 = nullptr;   // program of the last child's subtree
    int nops0 = 0, nops1 = 0;        // steps of the two programs (root step included)
    int rec1 = 0, kobs1 = 0;
    double *halfbuf = nullptr;       // null: the whole tree in one program
};

// waves of a split-M workgroup: whole tiles of NT row-tile waves (NT <= 4: up to four
// waves; 4 < NT <= 8, i.e. 64 < n <= 128 states: one tile of NT waves)
__host__ __device__ constexpr int rt_split_waves(int NT) { return NT == 3 ? 3 : (NT < 4 ? 4 : NT); }

template <int NT, int KS, bool STORE, bool SPARSE = false>
__global__ void __launch_bounds__(rt_split_waves(NT) * 64)
prune_mfma_kernel(const double *__restrict__ Pfrag,  // [nops][NT][KP][64][2]
                  const int4_t *__restrict__ prog, int nops,   // LOP_* program
                  const double *__restrict__ obs, int K,  // [blk16][K][KP][64][2]
                  const double *__restrict__ root_w, int n, int lds_slots,
                  double *__restrict__ loglik, int *__restrict__ status,
                  double *__restrict__ partial, long nsites, long nblocks16,
                  double *__restrict__ Lout, double *__restrict__ Mout, rt_interp_halves hv,
                  int rescale, const unsigned *__restrict__ leafw, const double *__restrict__ Pcol,
                  int sparse_mode)
{
    // (a compile-time switch: as a run-time one the leaf path cost the dense instance 35 %)
    const int sparse = SPARSE ? sparse_mode : 0;
    // sparse (1: one observed state per leaf, 2: allowed sets of one or two; batches whose
    // `sparse_ok` holds): a leaf step is a gathered column of P (or two, added) from the model's
    // leaf-column table instead of an x exchange and KS MFMAs -- what the tree-specialised
    // kernels do (jit.hip), for the trees that have none: more than 600 steps, or not compiled yet.
    // Lout / Mout (optional): own rows of L_v and of the message M_v = P_v L_v of every step,
    // [step][tile][m][r][lane] -- what the downward pass and the site sums of the expectation
    // path read back (csrc/expect_mfma.hip)
    constexpr int WAVES = rt_split_waves(NT);
    constexpr int TILES = WAVES / NT;
    constexpr int KP = (KS + 1) / 2;           // k-step pairs
    constexpr int XB = NT * 4 * 64;            // doubles of one x exchange buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave / NT;
    const int m = wave - tile * NT;
    const bool halves = !STORE && hv.halfbuf != nullptr;
    const int half = halves ? (int)(blockIdx.x & 1) : 0;
    const long bidx = halves ? (long)(blockIdx.x >> 1) : (long)blockIdx.x;
    const long gt = bidx * TILES + tile;                   // global site tile
    const bool live = gt < nblocks16;
    const long blk = live ? gt : nblocks16 - 1;           // keep barriers uniform
    if (halves) {
        nops = half ? hv.nops1 : hv.nops0;
        if (half) prog = hv.prog1;
    }
    const int rec0 = half ? hv.rec1 : 0;                   // P record of this program's step 0
    const int kobs0 = half ? hv.kobs1 : 0;                 // stream position of its first leaf

    double *xb = (double *)smem + tile * XB;               // [TILES][XB], single buffer
    double *red = (double *)smem + TILES * XB;             // [TILES][NT][16]
    // this wave's rows of the spilled accumulators: [slot][4][64], + lane
    unsigned char *stack = (unsigned char *)(red + TILES * NT * 16) +
                           (size_t)wave * lds_slots * 2048 + lane * 8;

    const RT_CONST_AS int4_t *prog_c = (const RT_CONST_AS int4_t *)prog;
    // A fragments of this wave: [op][m][q][lane][2]
    constexpr size_t ASTRIDE = (size_t)NT * KP * 128;
    const double *ag = Pfrag + (size_t)rec0 * ASTRIDE + ((size_t)m * KP * 64 + lane) * 2;
    // observation pairs holding this wave's own rows 4m..4m+3: q = 2m, 2m+1
    const double *og = obs + (size_t)blk * K * (KP * 128) + lane * 2;

    double an[2 * KP];            // A fragments of the NEXT step
    double on[4];                 // own rows of the next observation in the stream
#pragma unroll
    for (int j = 0; j < 4; ++j) on[j] = 1.0;

    int4_t op = prog_c[0];
    int knext = kobs0;            // stream position `on` holds / will hold
#pragma unroll
    for (int q = 0; q < KP; ++q) {
        const double2 v = *(const double2 *)(ag + q * 128);
        an[2 * q] = v.x;
        an[2 * q + 1] = v.y;
    }
    if (!SPARSE && knext < K) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 2 * m + h;
            double2 v = {0.0, 0.0};
            if (q < KP) v = *(const double2 *)(og + ((size_t)knext * KP + q) * 128);
            on[2 * h] = v.x;
            on[2 * h + 1] = v.y;
        }
    }

    double lik = 0.0;
    bool negative = false;
    double cur[4] = {1.0, 1.0, 1.0, 1.0};      // register-cached top accumulator
    int escale = 0;               // rescaling: exponent tally of this lane's site (lane & 15)

    // ---- sparse leaves: the state words (two in flight) and the gathered column of the NEXT
    // leaf step, requested when the step before it starts its products
    const int per_word = sparse == 2 ? 2 : 4;
    const int KW = (K + per_word - 1) / per_word;
    const unsigned *lwp = SPARSE ? leafw + (size_t)blk * KW * 16 + (lane & 15) : nullptr;
    const double4_t *pcm = (const double4_t *)Pcol + (4 * m + (lane >> 4));
    int lwidx = SPARSE ? kobs0 / per_word : 0;
    unsigned lwcur = 0, lwnext = 0;
    if constexpr (SPARSE) {
        lwcur = lwp[(size_t)lwidx * 16];
        lwnext = lwp[(size_t)(lwidx + 1 < KW ? lwidx + 1 : lwidx) * 16];
    }
    double4_t pcn = {0.0, 0.0, 0.0, 0.0}, pcq = {0.0, 0.0, 0.0, 0.0};
    bool pair_on = false;
    // the column(s) of the leaf whose step is program index `step`, stream position kpos
    auto gather = [&](int step, int kpos) {
        const int wq = kpos / per_word;
        if (wq != lwidx) {                       // (uniform) the next word becomes the current one
            lwcur = lwnext;
            lwidx = wq;
            lwnext = lwp[(size_t)(wq + 1 < KW ? wq + 1 : wq) * 16];
        }
        const int sh = sparse == 2 ? 16 * (kpos & 1) : 8 * (kpos & 3);
        const int sa = (int)((lwcur >> sh) & 255u);
        pcn = pcm[((size_t)(rec0 + step) * n + sa) * (4 * NT)];
        if (sparse == 2) {
            const int sb = (int)((lwcur >> (sh + 8)) & 255u);
            pair_on = sb != 255;
            pcq = pcm[((size_t)(rec0 + step) * n + (sb != 255 ? sb : sa)) * (4 * NT)];
        }
    };
    auto is_sleaf = [&](int f) { return SPARSE && !(f & LOP_INTERNAL) && (f & LOP_OBS); };
    if constexpr (SPARSE) {
        if (is_sleaf(op.x)) gather(0, kobs0);
    }

    for (int i = 0; i < nops; ++i) {
        const int flags = op.x;
        const bool sleaf = SPARSE && is_sleaf(flags);      // (uniform)
        // own rows of L_v = (accumulator of v) * (observation at v)
        double x[4];
        if (flags & LOP_INTERNAL) {
            if (flags & LOP_X_CUR) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = cur[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = *(const double *)(stack + op.y + r * 512);
            }
            if (flags & LOP_OBS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] *= on[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (flags & LOP_OBS) ? on[r] : 1.0;
        }
        if (flags & LOP_OBS) knext += 1;
        // (L of an observed leaf is its observation vector, which is resident already: the site
        // sums read it from there, expect_mfma.hip -- a quarter of this pass's stores)
        if (STORE && live && ((flags & LOP_INTERNAL) || !(flags & LOP_OBS))) {
            double *lo = Lout + ((size_t)i * nblocks16 + gt) * (NT * 256) + (m * 4) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[r * 64] = x[r];
        }
        if (flags & LOP_ROOT) {
            if (halves) {
                // this program's share of the root's accumulator (the root's own observation
                // is the combine kernel's: the half programs' root step carries none)
                if (live) {
                    double *hb = hv.halfbuf + ((size_t)gt * 2 + half) * (NT * 256) + (m * 4) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hb[r * 64] = x[r];
                }
                return;
            }
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * m + 4 * r + (lane >> 4);
                const double w = row < n ? root_w[row] : 0.0;
                negative |= (row < n) && (x[r] < 0.0);
                s += w * fmax(x[r], 0.0);
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) red[(tile * NT + m) * 16 + lane] = s;
            __syncthreads();
            if (m == 0 && lane < 16) {
                double tot = 0.0;
#pragma unroll
                for (int mm = 0; mm < NT; ++mm) tot += red[(tile * NT + mm) * 16 + lane];
                lik = tot;
            }
            break;
        }
        double a[2 * KP];
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        if (SPARSE && sleaf) {
            // the message of this leaf: its column(s), requested one step ago (one rounding for
            // two columns, as the chain of the matrix pipe adds two non-zero terms)
            acc = pcn;
            if (sparse == 2 && pair_on) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = pcn[r] + pcq[r];
            }
        } else {
            __syncthreads();      // every wave is done reading the previous step's x
#pragma unroll
            for (int r = 0; r < 4; ++r) xb[((4 * m + r) * 64) + lane] = x[r];
#pragma unroll
            for (int j = 0; j < 2 * KP; ++j) a[j] = an[j];
            __syncthreads();      // x of every wave of the tile is in LDS
        }

        // start everything the next step needs
        const int inext = (i + 1 < nops) ? i + 1 : i;
        const int4_t opn = prog_c[inext];
        if (SPARSE && is_sleaf(opn.x)) {
            if constexpr (SPARSE) {
                if (inext != i) gather(inext, knext);
            }
        } else if (!(opn.x & LOP_ROOT)) {
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                const double2 v = *(const double2 *)(ag + (size_t)inext * ASTRIDE + q * 128);
                an[2 * q] = v.x;
                an[2 * q + 1] = v.y;
            }
        }
        if (!SPARSE && (flags & LOP_OBS) && knext < K) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = 2 * m + h;
                double2 v = {0.0, 0.0};
                if (q < KP) v = *(const double2 *)(og + ((size_t)knext * KP + q) * 128);
                on[2 * h] = v.x;
                on[2 * h + 1] = v.y;
            }
        }

        if (!sleaf) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const double b = xb[kk * 64 + lane];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b, acc, 0, 0, 0);
            }
        }
        if (rescale && !sleaf) {
            // every wave of the tile sees all of x as its B operands: the site's largest
            // entry, the same number in every lane of the site and in every wave
            double xmax = 0.0;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) xmax = fmax(xmax, xb[kk * 64 + lane]);
            xmax = fmax(xmax, __shfl_xor(xmax, 16, 64));
            xmax = fmax(xmax, __shfl_xor(xmax, 32, 64));
            const int e = rescale_exponent(xmax);
            if (e != 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = ldexp(acc[r], -e);
                escale += e;
            }
        }
        if (STORE && live) {
            double *mo = Mout + ((size_t)i * nblocks16 + gt) * (NT * 256) + (m * 4) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) mo[r * 64] = acc[r];
        }

        if (flags & LOP_FIRST) {
            if (flags & LOP_SPILL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) *(double *)(stack + op.w + r * 512) = cur[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] = acc[r];
        } else if (flags & LOP_DST_CUR) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] *= acc[r];
        } else if (flags & LOP_FAST) {     // un-spill: the parent's step is next
#pragma unroll
            for (int r = 0; r < 4; ++r)
                cur[r] = *(const double *)(stack + op.z + r * 512) * acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *d = (double *)(stack + op.z + r * 512);
                *d = *d * acc[r];
            }
        }
        op = opn;
    }

    // lanes 0..15 of wave m == 0 own the 16 sites of the tile
    const long site = blk * 16 + (lane & 15);
    const bool valid = live && m == 0 && lane < 16 && site < nsites;
    double sum, nzero;
    finish_site(lik, negative, valid, loglik, status, site, sum, nzero, escale);
    sum = wave_sum(sum);
    nzero = wave_sum(nzero);
    if (lane == 0) {
        const long gwv = (long)blockIdx.x * WAVES + wave;
        partial[gwv * 2] = sum;
        partial[gwv * 2 + 1] = nzero;
    }
}

// Second kernel of a root-halves interpreter launch: the root's accumulator is the product of
// the two shares (in child order: the interpreter's own product bit for bit), times the root's
// observation if it has one (stream position kroot), then the root step and the site epilogue
// of prune_mfma_kernel unchanged -- same workgroup shape, same partial sums per wave.
template <int NT>
__global__ void __launch_bounds__(rt_split_waves(NT) * 64)
prune_mfma_combine_kernel(const double *__restrict__ halfbuf, const double *__restrict__ obs,
                          int K, int KP, int kroot, const double *__restrict__ root_w, int n,
                          double *__restrict__ loglik, int *__restrict__ status,
                          double *__restrict__ partial, long nsites, long nblocks16)
{
    constexpr int WAVES = rt_split_waves(NT);
    constexpr int TILES = WAVES / NT;
    __shared__ double red[TILES * NT * 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = wave / NT;
    const int m = wave - tile * NT;
    const long gt = (long)blockIdx.x * TILES + tile;
    const bool live = gt < nblocks16;
    const long blk = live ? gt : nblocks16 - 1;
    const double *ha = halfbuf + (size_t)blk * 2 * (NT * 256) + (m * 4) * 64 + lane;
    double x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = ha[r * 64] * ha[NT * 256 + r * 64];
    if (kroot >= 0) {
        const double *og = obs + (size_t)blk * K * (KP * 128) + lane * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 2 * m + h;
            double2 v = {0.0, 0.0};
            if (q < KP) v = *(const double2 *)(og + ((size_t)kroot * KP + q) * 128);
            x[2 * h] *= v.x;
            x[2 * h + 1] *= v.y;
        }
    }
    double lik = 0.0;
    bool negative = false;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * m + 4 * r + (lane >> 4);
        const double w = row < n ? root_w[row] : 0.0;
        negative |= (row < n) && (x[r] < 0.0);
        s += w * fmax(x[r], 0.0);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[(tile * NT + m) * 16 + lane] = s;
    __syncthreads();
    if (m == 0 && lane < 16) {
        double tot = 0.0;
#pragma unroll
        for (int mm = 0; mm < NT; ++mm) tot += red[(tile * NT + mm) * 16 + lane];
        lik = tot;
    }
    const long site = blk * 16 + (lane & 15);
    const bool valid = live && m == 0 && lane < 16 && site < nsites;
    double sum, nzero;
    finish_site(lik, negative, valid, loglik, status, site, sum, nzero);
    sum = wave_sum(sum);
    nzero = wave_sum(nzero);
    if (lane == 0) {
        const long gwv = (long)blockIdx.x * WAVES + wave;
        partial[gwv * 2] = sum;
        partial[gwv * 2 + 1] = nzero;
    }
}

// ---------------------------------------------------------------------------
// 4 < n <= 32: v_mfma_f64_16x16x4_f64, ONE wave owns a 16-site tile outright
// ---------------------------------------------------------------------------
//
// With at most two row tiles the whole message (NT*4 doubles per lane) and the
// A fragments of an edge (NT*KS doubles per lane) fit comfortably in registers,
// so a wave computes all rows itself: the D registers ARE the next B operand
// (same lane, same register), nothing is exchanged between waves, there is no
// barrier anywhere, and four independent waves share a workgroup only to share
// the launch.  Stack: [slot][NT*4][lane] doubles per wave in LDS, top cached in
// registers (same LOP_* program as the other kernels).

template <int NT, int KS>
__global__ void __launch_bounds__(256)
prune_mfma_solo_kernel(const double *__restrict__ Pfrag,  // [nops][NT][KP][64][2]
                       const int4_t *__restrict__ prog, int nops,
                       const double *__restrict__ obs, int K,  // [blk16][K][KP][64][2]
                       const double *__restrict__ root_w, int n, int lds_slots,
                       double *__restrict__ loglik, int *__restrict__ status,
                       double *__restrict__ partial, long nsites, long nblocks16, int rescale)
{
    constexpr int MW = NT * 4;                 // message doubles per lane
    constexpr int KP = (KS + 1) / 2;           // k-step pairs
    constexpr int NA = NT * 2 * KP;            // A-fragment doubles per lane and step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long gw = (long)blockIdx.x * 4 + wave;           // site tile of this wave
    if (gw >= nblocks16) {                                  // wave-uniform; no barriers
        if (lane == 0) { partial[gw * 2] = 0.0; partial[gw * 2 + 1] = 0.0; }
        return;
    }
    unsigned char *stack = smem + (size_t)wave * lds_slots * (MW * 512) + lane * 8;
    const RT_CONST_AS int4_t *prog_c = (const RT_CONST_AS int4_t *)prog;
    const double *ag = Pfrag + lane * 2;                    // [op][m][q][lane][2]
    constexpr size_t ASTRIDE = (size_t)NT * KP * 128;
    const double *og = obs + (size_t)gw * K * (KP * 128) + lane * 2;

    double an[NA];                // A fragments of the NEXT step
    double on[2 * KP];            // next observation of the stream (B-operand order)
#pragma unroll
    for (int j = 0; j < 2 * KP; ++j) on[j] = 1.0;
    int4_t op = prog_c[0];
    int knext = 0;
#pragma unroll
    for (int q = 0; q < NT * KP; ++q) {
        const double2 v = *(const double2 *)(ag + q * 128);
        an[2 * q] = v.x;
        an[2 * q + 1] = v.y;
    }
    if (K > 0) {
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            const double2 v = *(const double2 *)(og + (size_t)q * 128);
            on[2 * q] = v.x;
            on[2 * q + 1] = v.y;
        }
    }

    double lik = 0.0;
    bool negative = false;
    int escale = 0;               // rescaling: exponent tally of this lane's site
    double cur[MW];
#pragma unroll
    for (int j = 0; j < MW; ++j) cur[j] = 1.0;

    for (int i = 0; i < nops; ++i) {
        const int flags = op.x;
        double x[MW];
        if (flags & LOP_INTERNAL) {
            if (flags & LOP_X_CUR) {
#pragma unroll
                for (int j = 0; j < MW; ++j) x[j] = cur[j];
            } else {
#pragma unroll
                for (int j = 0; j < MW; ++j) x[j] = *(const double *)(stack + op.y + j * 512);
            }
            if (flags & LOP_OBS) {
#pragma unroll
                for (int j = 0; j < KS; ++j) x[j] *= on[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < MW; ++j)
                x[j] = (flags & LOP_OBS) ? (j < 2 * KP ? on[j] : 0.0) : 1.0;
        }
        if (flags & LOP_OBS) knext += 1;
        if (flags & LOP_ROOT) {
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int row = 4 * j + (lane >> 4);
                const double w = row < n ? root_w[row] : 0.0;
                negative |= (row < n) && (x[j] < 0.0);
                sacc += w * fmax(x[j], 0.0);
            }
            sacc += __shfl_xor(sacc, 16, 64);
            sacc += __shfl_xor(sacc, 32, 64);
            lik = sacc;
            break;
        }
        double a[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) a[j] = an[j];

        // start everything the next step needs
        const int inext = (i + 1 < nops) ? i + 1 : i;
        const int4_t opn = prog_c[inext];
        if (!(opn.x & LOP_ROOT)) {
#pragma unroll
            for (int q = 0; q < NT * KP; ++q) {
                const double2 v = *(const double2 *)(ag + (size_t)inext * ASTRIDE + q * 128);
                an[2 * q] = v.x;
                an[2 * q + 1] = v.y;
            }
        }
        if ((flags & LOP_OBS) && knext < K) {
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                const double2 v = *(const double2 *)(og + ((size_t)knext * KP + q) * 128);
                on[2 * q] = v.x;
                on[2 * q + 1] = v.y;
            }
        }

        double4_t acc[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[m] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int m = 0; m < NT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m * 2 * KP + kk], x[kk],
                                                              acc[m], 0, 0, 0);
        }
        double t[MW];
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[m * 4 + r] = acc[m][r];
        if (rescale) {
            double xmax = 0.0;
#pragma unroll
            for (int j = 0; j < KS; ++j) xmax = fmax(xmax, x[j]);
            xmax = fmax(xmax, __shfl_xor(xmax, 16, 64));
            xmax = fmax(xmax, __shfl_xor(xmax, 32, 64));
            const int e = rescale_exponent(xmax);
            if (e != 0) {
#pragma unroll
                for (int j = 0; j < MW; ++j) t[j] = ldexp(t[j], -e);
                escale += e;
            }
        }

        if (flags & LOP_FIRST) {
            if (flags & LOP_SPILL) {
#pragma unroll
                for (int j = 0; j < MW; ++j) *(double *)(stack + op.w + j * 512) = cur[j];
            }
#pragma unroll
            for (int j = 0; j < MW; ++j) cur[j] = t[j];
        } else if (flags & LOP_DST_CUR) {
#pragma unroll
            for (int j = 0; j < MW; ++j) cur[j] *= t[j];
        } else if (flags & LOP_FAST) {     // un-spill: the parent's step is next
#pragma unroll
            for (int j = 0; j < MW; ++j)
                cur[j] = *(const double *)(stack + op.z + j * 512) * t[j];
        } else {
#pragma unroll
            for (int j = 0; j < MW; ++j) {
                double *d = (double *)(stack + op.z + j * 512);
                *d = *d * t[j];
            }
        }
        op = opn;
    }

    // lanes 0..15 own the 16 sites of the tile
    const long site = gw * 16 + (lane & 15);
    const bool valid = lane < 16 && site < nsites;
    double sum, nzero;
    finish_site(lik, negative, valid, loglik, status, site, sum, nzero, escale);
    sum = wave_sum(sum);
    nzero = wave_sum(nzero);
    if (lane == 0) {
        partial[gw * 2] = sum;
        partial[gw * 2 + 1] = nzero;
    }
}

// ---------------------------------------------------------------------------
// generic fallback: lane per site, accumulator stack in global scratch
// ---------------------------------------------------------------------------

template <int NMAX>
__global__ void __launch_bounds__(64)
prune_generic_kernel(const double *__restrict__ P,   // [nnodes][n][n]
                     const rt_op *__restrict__ ops, int nops,
                     const double *__restrict__ obs, int K, int n, int np,
                     const double *__restrict__ root_w,
                     double *__restrict__ loglik, int *__restrict__ status,
                     double *__restrict__ partial, double *__restrict__ scratch,
                     long nsp, long nsites, int rescale)
{
    const int lane = threadIdx.x;
    const long blk = blockIdx.x;
    const long site = blk * 64 + lane;
    double lik = 0.0;
    bool negative = false;
    int escale = 0;
    for (int i = 0; i < nops; ++i) {
        const rt_op op = ops[i];
        double x[NMAX];
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            x[j] = 1.0;
            if (j < n) {
                if (op.pop >= 0) x[j] = scratch[((long)op.pop * n + j) * nsp + site];
                if (op.obs >= 0)
                    x[j] *= obs[(((size_t)blk * K + op.obs) * 64 + lane) * np + j];
            }
        }
        if (op.dst < 0) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                if (j < n) {
                    negative |= x[j] < 0.0;
                    s += root_w[j] * fmax(x[j], 0.0);
                }
            }
            lik = s;
        } else {
            const int dslot = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            const double *Pe = P + (long)op.node * n * n;
            int e = 0;
            if (rescale) {
                double mx = 0.0;
#pragma unroll
                for (int j = 0; j < NMAX; ++j)
                    if (j < n) mx = fmax(mx, x[j]);
                e = rescale_exponent(mx);
                escale += e;
            }
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < NMAX; ++j)
                    if (j < n) s = fma(Pe[r * n + j], x[j], s);
                if (e != 0) s = ldexp(s, -e);
                double *d = scratch + ((long)dslot * n + r) * nsp + site;
                *d = first ? s : *d * s;
            }
        }
    }
    double sum, nzero;
    finish_site(lik, negative, site < nsites, loglik, status, site, sum, nzero, escale);
    sum = wave_sum(sum);
    nzero = wave_sum(nzero);
    if (lane == 0) {
        partial[blk * 2] = sum;
        partial[blk * 2 + 1] = nzero;
    }
}

// ---------------------------------------------------------------------------
// fixed-order reduction of the per-wave partial sums -> totals[3]
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
reduce_partials_kernel(const double *__restrict__ partial, long npartials,
                       double *__restrict__ totals, double nsites)
{
    rt_reduce_partials_body(partial, npartials, totals, nsites);
}

// ---------------------------------------------------------------------------
// P repack: esd_transitions [node][n][n] -> kernel-native, step-ordered
// ---------------------------------------------------------------------------

// lane family: Pord[i][r][c] = P[node_i][r][c]
__global__ void pack_p_lane_kernel(const double *__restrict__ P,
                                   const rt_op *__restrict__ ops, int nops, int n,
                                   double *__restrict__ Pord)
{
    const int i = blockIdx.x;
    const rt_op op = ops[i];
    for (int e = threadIdx.x; e < n * n; e += blockDim.x)
        Pord[(long)i * n * n + e] = op.dst < 0 ? 0.0 : P[(long)op.node * n * n + e];
}

// mfma family: Pfrag[i][m][q][lane][e] = P[node_i][16m + (lane&15)][4(2q+e) + (lane>>4)]
__global__ void pack_p_mfma_kernel(const double *__restrict__ P,
                                   const rt_op *__restrict__ ops, int nops, int n,
                                   int NT, int KP, double *__restrict__ Pfrag)
{
    const int i = blockIdx.x;
    const rt_op op = ops[i];
    const int total = NT * KP * 128;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int e2 = e & 1;
        const int lane = (e >> 1) & 63;
        const int q = (e >> 7) % KP;
        const int m = (e >> 7) / KP;
        const int row = 16 * m + (lane & 15);
        const int col = 4 * (2 * q + e2) + (lane >> 4);
        double v = 0.0;
        if (op.dst >= 0 && row < n && col < n)
            v = P[(long)op.node * n * n + row * n + col];
        Pfrag[(long)i * total + e] = v;
    }
}

// quad blocks: Pquad[i][rq][kk][k][r] = P[node_i][4 rq + r][4 kk + k]
__global__ void pack_p_quad_kernel(const double *__restrict__ P,
                                   const rt_op *__restrict__ ops, int nops, int n, int KS,
                                   double *__restrict__ Pquad)
{
    const int i = blockIdx.x;
    const rt_op op = ops[i];
    const int total = KS * KS * 16;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int r = e & 3, k = (e >> 2) & 3, blk = e >> 4;
        const int row = 4 * (blk / KS) + r, col = 4 * (blk % KS) + k;
        double v = 0.0;
        if (op.dst >= 0 && row < n && col < n) v = P[(long)op.node * n * n + row * n + col];
        Pquad[(long)i * rt_quad_stride(n) + e] = v;
    }
}

// ---------------------------------------------------------------------------
// site packing: user data -> kernel-native HBM layout
// ---------------------------------------------------------------------------

__device__ __forceinline__ double obs_value(int kind, const void *data, long site,
                                            long nobs, long j, int n, int s)
{
    if (kind == RT_OBS_DENSE)
        return ((const double *)data)[((size_t)site * nobs + j) * n + s];
    if (kind == RT_OBS_STATE) {
        const unsigned char st = ((const unsigned char *)data)[(size_t)site * nobs + j];
        return (st == 255 || st == s) ? 1.0 : 0.0;
    }
    // allowed-set masks: ceil(n / 64) words per (site, node), bit s % 64 of word s / 64
    const int words = (n + 63) >> 6;
    const unsigned long long m =
        ((const unsigned long long *)data)[((size_t)site * nobs + j) * words + (s >> 6)];
    return ((m >> (s & 63)) & 1ull) ? 1.0 : 0.0;
}

// lane family: [blk][k][pair][lane][2] with S sites per block (64 unless the batch
// runs a tree-specialised kernel, jit.hip); the generic fallback keeps the
// simpler [blk64][k][lane][np]
__global__ void pack_sites_lane_kernel(int kind, const void *__restrict__ data,
                                       const int *__restrict__ src_of_k, long nsites,
                                       long nobs, int K, int n, int np, int paired, int S,
                                       double *__restrict__ out, size_t total)
{
    const int hp = np / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        int s, lane;
        size_t r2;
        if (paired) {
            const int e2 = e & 1;
            lane = (int)((e >> 1) % S);
            const size_t r = (e >> 1) / S;
            s = 2 * (int)(r % hp) + e2;
            r2 = r / hp;
        } else {
            s = e % np;
            const size_t r = e / np;
            lane = (int)(r % S);
            r2 = r / S;
        }
        const int k = r2 % K;
        const long blk = r2 / K;
        const long site = blk * S + lane;
        double v;
        if (s >= n) v = 0.0;
        else if (site >= nsites) v = 1.0;
        else v = obs_value(kind, data, site, nobs, src_of_k[k], n, s);
        out[e] = v;
    }
}

// lane family, compact: uint8 states stay states, [blk][word][lane], byte j of word q =
// stream position 4q + j (255: unobserved, padding positions and padding sites)
__global__ void pack_sites_lane_state_kernel(int kind, int n, const void *__restrict__ data,
                                             const int *__restrict__ src_of_k, long nsites,
                                             long nobs, int K, int S,
                                             unsigned *__restrict__ out, size_t total)
{
    // states: 255 = unobserved; masks (n <= 4): the low n bits, all set = unobserved
    const unsigned unobserved = kind == RT_OBS_MASK ? (1u << n) - 1u : 255u;
    const int KQ = (K + 3) / 4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e % S);
        const size_t r = e / S;
        const int q = (int)(r % KQ);
        const long site = (long)(r / KQ) * S + lane;
        unsigned w = 0;
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
            unsigned st = unobserved;
            if (k < K && site < nsites) {
                const size_t at = (size_t)site * nobs + src_of_k[k];
                st = kind == RT_OBS_MASK
                         ? (unsigned)(((const unsigned long long *)data)[at] & unobserved)
                         : ((const unsigned char *)data)[at];
            }
            w |= st << (8 * j);
        }
        out[e] = w;
    }
}

// [blk16][k][kkpair][lane][2]: element e2 of pair q is k-step 2q+e2, state
// 4(2q+e2) + (lane>>4), site blk*16 + (lane&15)
__global__ void pack_sites_mfma_kernel(int kind, const void *__restrict__ data,
                                       const int *__restrict__ src_of_k, long nsites,
                                       long nobs, int K, int n, int KP,
                                       double *__restrict__ out, size_t total)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int e2 = e & 1;
        const int lane = (e >> 1) & 63;
        const size_t r = e >> 7;
        const int q = r % KP;
        const size_t r2 = r / KP;
        const int k = r2 % K;
        const long blk = r2 / K;
        const int s = 4 * (2 * q + e2) + (lane >> 4);
        const long site = blk * 16 + (lane & 15);
        double v;
        if (s >= n) v = 0.0;
        else if (site >= nsites) v = 1.0;
        else v = obs_value(kind, data, site, nobs, src_of_k[k], n, s);
        out[e] = v;
    }
}

// leaf states as bytes for the column-gathering kernels (jit.hip, `sparse`): word w of
// (tile, site) holds stream positions 4w .. 4w+3; sites past the batch read state 0
__global__ void pack_leaf_words_kernel(const unsigned char *__restrict__ data,
                                       const int *__restrict__ src_of_k, long nsites, long nobs,
                                       int K, long nblocks16, unsigned *__restrict__ out)
{
    const int KW = (K + 3) / 4;
    const size_t total = (size_t)nblocks16 * KW * 16;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 15);
        const size_t r = e >> 4;
        const int w = (int)(r % KW);
        const long site = (long)(r / KW) * 16 + lane;
        unsigned word = 0;
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * w + j;
            unsigned st = 0;
            if (k < K && site < nsites) st = data[(size_t)site * nobs + src_of_k[k]];
            word |= (st & 255u) << (8 * j);
        }
        out[e] = word;
    }
}

// ... and as allowed sets of one or two states (masks): 16 bits per leaf, the lower state in the
// low byte, the other one (255: none) in the high byte; word w holds stream positions 2w, 2w+1
__global__ void pack_leaf_pairs_kernel(const unsigned long long *__restrict__ data,
                                       const int *__restrict__ src_of_k, long nsites, long nobs,
                                       int K, int nwords, long nblocks16, unsigned *__restrict__ out)
{
    const int KW = (K + 1) / 2;
    const size_t total = (size_t)nblocks16 * KW * 16;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 15);
        const size_t r = e >> 4;
        const int w = (int)(r % KW);
        const long site = (long)(r / KW) * 16 + lane;
        unsigned word = 0;
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * w + j;
            unsigned a = 0, b = 255;
            if (k < K && site < nsites) {
                const unsigned long long *mk = data + ((size_t)site * nobs + src_of_k[k]) * nwords;
                int found = 0;
                for (int q = 0; q < nwords; ++q) {
                    unsigned long long v = mk[q];
                    while (v && found < 2) {
                        const int bit = __ffsll((long long)v) - 1;
                        if (found == 0) a = (unsigned)(64 * q + bit);
                        else b = (unsigned)(64 * q + bit);
                        ++found;
                        v &= v - 1;
                    }
                }
            }
            word |= ((a & 255u) | ((b & 255u) << 8)) << (16 * j);
        }
        out[e] = word;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

static int ks_of(int64_t n) { return (int)((n + 3) / 4); }
static int nt_of(int64_t n) { return (int)((n + 15) / 16); }

// leaf-column table (rt_model::d_Pcol): Pcol[step][s][m][q][r] = P[node][16 m + 4 r + q][s].
// A transpose: 32 columns at a time through LDS, so that both the reads of P (rows) and the
// writes of the table (columns) are contiguous runs (strided on one side it cost 9.8 us per
// step of the 61-state model).
__global__ void __launch_bounds__(256)
pack_pcol_kernel(const double *__restrict__ P, const rt_op *__restrict__ ops, int nops, int n,
                 int NT, double *__restrict__ out)
{
    __shared__ double tile[128][33];
    const int i = blockIdx.x;
    const rt_op op = ops[i];
    if (op.pop >= 0 || op.dst < 0) return;      // only the leaves' records are ever read
    const int RN = 16 * NT;
    const double *Pn = P + (long)op.node * n * n;
    double *o = out + (long)i * n * RN;
    const bool live = op.dst >= 0;
    for (int c0 = 0; c0 < n; c0 += 32) {
        // (all of a thread's loads of the chunk in flight together: n <= 128 rows are at most 16)
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int row = e >> 5, c = e & 31;
            v[u] = (live && row < n && c0 + c < n) ? Pn[(long)row * n + c0 + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = threadIdx.x + 256 * u;
            if ((e >> 5) < n) tile[e >> 5][e & 31] = v[u];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * RN; e += 256) {
            const int c = e / RN, x = e - c * RN;
            const int row = 16 * (x >> 4) + 4 * (x & 3) + ((x >> 2) & 3);
            if (c0 + c < n) o[(long)(c0 + c) * RN + x] = row < n ? tile[row][c] : 0.0;
        }
        __syncthreads();
    }
}

int rt_model_pack_pcol(rt_model *m)
{
    if (!m->d_Pcol || m->n <= 4) return RT_OK;
    hipLaunchKernelGGL(pack_pcol_kernel, dim3((unsigned)m->ops.size()), dim3(256), 0, m->ctx->stream,
                       m->d_P, m->d_ops, (int)m->ops.size(), (int)m->n, (int)((m->n + 15) / 16),
                       m->d_Pcol);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

int rt_model_need_pcol(rt_model *m)
{
    if (m->d_Pcol || m->n <= 4) return RT_OK;
    const size_t rn = (size_t)16 * ((m->n + 15) / 16);
    RT_HIP(hipMalloc((void **)&m->d_Pcol, m->ops.size() * (size_t)m->n * rn * 8));
    return m->have_P ? rt_model_pack_pcol(m) : RT_OK;
}

int rt_launch_pfrag(rt_model *m)
{
    if (!m->frag_dirty) return RT_OK;
    const int nops = (int)m->ops.size();
    const int n = (int)m->n;
    hipStream_t st = m->ctx->stream;
    if (n <= 4) {
        hipLaunchKernelGGL(pack_p_lane_kernel, dim3(nops), dim3(64), 0, st, m->d_P,
                           m->d_ops, nops, n, m->d_Pfrag);
    } else {
        hipLaunchKernelGGL(pack_p_mfma_kernel, dim3(nops), dim3(256), 0, st, m->d_P,
                           m->d_ops, nops, n, nt_of(n), (ks_of(n) + 1) / 2, m->d_Pfrag);
        if (m->d_Pquad)
            hipLaunchKernelGGL(pack_p_quad_kernel, dim3(nops), dim3(256), 0, st, m->d_P,
                               m->d_ops, nops, n, ks_of(n), m->d_Pquad);
    }
    RT_HIP(hipGetLastError());
    RT_TRY(rt_model_pack_pcol(m));
    m->frag_dirty = false;
    return RT_OK;
}

// the user's observations (already on the device) -> the batch's resident layout
int rt_sites_pack_device(rt_sites *s, int kind, const void *d_in, const int *d_src)
{
    rt_model *m = s->model;
    hipStream_t st = m->ctx->stream;
    const int n = (int)m->n;
    const int K = (int)s->nobs;
    if (K > 0) {
        const size_t total = (size_t)s->obs_bytes / 8;
        size_t blocks = (total + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        if (s->layout == RT_LAYOUT_LANE && s->compact_states) {
            const size_t words = (size_t)s->obs_bytes / 4;
            hipLaunchKernelGGL(pack_sites_lane_state_kernel,
                               dim3((unsigned)std::min<size_t>((words + 255) / 256, 65536)),
                               dim3(256), 0, st, kind, n, d_in, d_src,
                               (long)s->nsites, (long)K, K, s->block_sites,
                               (unsigned *)s->d_obs, words);
        } else if (s->layout == RT_LAYOUT_LANE) {
            const int np = (n + 1) & ~1;
            const int paired = s->d_scratch == nullptr;
            hipLaunchKernelGGL(pack_sites_lane_kernel, dim3((unsigned)blocks), dim3(256),
                               0, st, kind, d_in, d_src, (long)s->nsites, (long)K, K, n,
                               np, paired, s->block_sites, s->d_obs, total);
        } else {
            const int KP = (ks_of(n) + 1) / 2;
            hipLaunchKernelGGL(pack_sites_mfma_kernel, dim3((unsigned)blocks), dim3(256),
                               0, st, kind, d_in, d_src, (long)s->nsites, (long)K, K, n,
                               KP, s->d_obs, total);
            if (s->d_leafw && kind == RT_OBS_STATE)
                hipLaunchKernelGGL(pack_leaf_words_kernel, dim3(1024), dim3(256), 0, st,
                                   (const unsigned char *)d_in, d_src, (long)s->nsites, (long)K, K,
                                   (long)s->nblocks, s->d_leafw);
            if (s->d_leafw && kind == RT_OBS_MASK && s->sparse_pairs)
                hipLaunchKernelGGL(pack_leaf_pairs_kernel, dim3(1024), dim3(256), 0, st,
                                   (const unsigned long long *)d_in, d_src, (long)s->nsites, (long)K, K,
                                   (int)((n + 63) / 64), (long)s->nblocks, s->d_leafw);
        }
        RT_HIP(hipGetLastError());
    }
    return RT_OK;
}

int rt_sites_pack(rt_sites *s, int kind, const int64_t *src_of_k, const void *data)
{
    rt_model *m = s->model;
    hipStream_t st = m->ctx->stream;
    const int n = (int)m->n;
    const int K = (int)s->nobs;
    size_t in_bytes;
    if (kind == RT_OBS_DENSE) in_bytes = (size_t)s->nsites * K * n * 8;
    else if (kind == RT_OBS_STATE) in_bytes = (size_t)s->nsites * K;
    else in_bytes = (size_t)s->nsites * K * 8 * (size_t)((n + 63) / 64);
    void *d_in = nullptr;
    int *d_src = nullptr;
    std::vector<int> src(K);
    for (int k = 0; k < K; ++k) src[k] = (int)src_of_k[k];
    if (K > 0) {
        RT_HIP(hipMalloc(&d_in, in_bytes));
        RT_HIP(hipMalloc((void **)&d_src, sizeof(int) * K));
        RT_HIP(hipMemcpyAsync(d_in, data, in_bytes, hipMemcpyHostToDevice, st));
        RT_HIP(hipMemcpyAsync(d_src, src.data(), sizeof(int) * K,
                              hipMemcpyHostToDevice, st));
        const int rc = rt_sites_pack_device(s, kind, d_in, d_src);
        RT_HIP(hipStreamSynchronize(st));
        if (rc == RT_OK && s->keep_raw) {
            // a lane-family batch whose tree-specialised kernel is still compiling: the
            // kernel's resident layout differs (sites per wave, compact states), so the
            // observations are packed again when it arrives (rt_sites_jit_poll)
            s->d_raw = d_in;
            s->d_raw_src = d_src;
        } else {
            RT_HIP(hipFree(d_in));
            RT_HIP(hipFree(d_src));
        }
        return rc;
    }
    return RT_OK;
}

template <int N, int R, bool RESC = false>
static int launch_lane_reg(rt_model *m, rt_sites *s, bool *plds_out)
{
    // the deepest accumulator never leaves the register cache
    const int depth = std::max(1, m->max_depth - 1);
    const int nops = (int)s->ops.size();
    const int stack = depth * N * 512;                        // per wave
    const int ptab = ((nops + 1) * N * N * 8 + 15) & ~15;
    // P table in LDS when two 4-wave workgroups still fit on a CU
    const bool plds = ptab + 4 * stack <= 80 * 1024 && !getenv("RAOTEH_LANE_NO_PLDS");
    *plds_out = plds;
    const int depth_arg = getenv("RAOTEH_LANE_NOLOAD") ? -depth : depth;
    if (plds) {
        const int lds = ptab + 4 * stack;
        auto kern = prune_lane_kernel<N, R, true, RESC>;
        RT_HIP(hipFuncSetAttribute((const void *)kern,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        RT_LAUNCH_TIMED(m->ctx, kern, dim3((unsigned)((s->nblocks + 3) / 4)), dim3(256), lds,
                           m->d_Pfrag, (const int4_t *)s->d_lane_ops, nops,
                           s->d_obs, (int)s->nobs, m->d_root, depth_arg, s->d_loglik,
                           s->d_status, s->d_partial, (long)s->nsites, (long)s->nblocks);
    } else {
        auto kern = prune_lane_kernel<N, R, false, RESC>;
        RT_HIP(hipFuncSetAttribute((const void *)kern,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, stack));
        RT_LAUNCH_TIMED(m->ctx, kern, dim3((unsigned)s->nblocks), dim3(64), stack,
                           m->d_Pfrag, (const int4_t *)s->d_lane_ops, nops,
                           s->d_obs, (int)s->nobs, m->d_root, depth, s->d_loglik,
                           s->d_status, s->d_partial, (long)s->nsites, (long)s->nblocks);
    }
    return RT_OK;
}

static int lane_dma_lds(int n, int R, int B, int wpb, int nrec, int slots)
{
    const int np = (n + 1) & ~1;
    const int ptab = ((nrec + 1) * n * n * 8 + 15) & ~15;
    return ptab + wpb * B * (R * 64 * np * 8 + slots * n * 512);
}

template <int N, int R, int B, int WPB>
static int launch_lane_dma(rt_model *m, rt_sites *s)
{
    const int depth = s->lane_stack_slots;
    const int nrec = (int)s->ops.size();          // P records = schedule steps
    const int nops = (int)s->lane_nprog;          // program entries (cherries fused)
    const int lds = lane_dma_lds(N, R, B, WPB, nrec, depth);
    if (lds > 160 * 1024) {
        rt_set_error("LDS-DMA lane kernel: %d bytes of LDS needed (tree too large); "
                     "unset RAOTEH_LANE_VARIANT", lds);
        return RT_ERR_UNSUPPORTED;
    }
    // rt_sites pads the lane layout to an even number of blocks
    const long nwaves = (s->nblocks + B - 1) / B;
    auto kern = prune_lanedma_kernel<N, R, B, WPB>;
    RT_HIP(hipFuncSetAttribute((const void *)kern,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    RT_LAUNCH_TIMED(m->ctx, kern, dim3((unsigned)((nwaves + WPB - 1) / WPB)), dim3(64 * WPB), lds,
                       m->d_Pfrag, (const int4_t *)s->d_lane_ops, nops, nrec,
                       s->d_obs, (int)s->nobs, m->d_root, depth, s->d_loglik, s->d_status,
                       s->d_partial, (long)s->nsites, nwaves);
    return RT_OK;
}

template <int N>
static int launch_lane(rt_model *m, rt_sites *s, const char **name)
{
    int R = s->lane_ring;
    int B = 1, W = 4;
    int rc;
    bool plds = false;
    if (s->lane_dma) {
        const int nrec = (int)s->ops.size();
        const int slots = s->lane_stack_slots;
        // ring: 3 slots (measured on C2: 4 are slower than 3 even where they fit)
        if (R == 0) R = 3;
        // two site blocks per wave (RAOTEH_LANE_BLOCKS=2) halve the scalar / P-read
        // work per site, but one wave with two blocks hides latency worse than two
        // waves with one: measured 60 us against 48 us on C2, so one is the default
        B = 1;
        if (const char *v = getenv("RAOTEH_LANE_BLOCKS")) B = atoi(v) == 2 ? 2 : 1;
        if (B == 2 && lane_dma_lds(N, R, 2, 4, nrec, slots) > 160 * 1024) W = 2;
        if (B == 2 && lane_dma_lds(N, R, 2, W, nrec, slots) > 160 * 1024) B = 1, W = 4;
        if (const char *v = getenv("RAOTEH_LANE_WPB")) W = atoi(v) == 2 ? 2 : 4;
        if (B == 1) W = 4;
    }
    if (s->lane_dma) {
#define RT_DMA_CASE(r) \
        rc = B == 2 ? (W == 2 ? launch_lane_dma<N, r, 2, 2>(m, s)               \
                              : launch_lane_dma<N, r, 2, 4>(m, s))              \
                    : launch_lane_dma<N, r, 1, 4>(m, s)
        switch (R) {
        case 2: RT_DMA_CASE(2); break;
        case 4: RT_DMA_CASE(4); break;
        default: R = 3; RT_DMA_CASE(3); break;
        }
#undef RT_DMA_CASE
    } else if (s->rescale) {
        R = 8;
        rc = launch_lane_reg<N, 8, true>(m, s, &plds);
    } else {
        switch (R) {
        case 4: rc = launch_lane_reg<N, 4>(m, s, &plds); break;
        case 6: rc = launch_lane_reg<N, 6>(m, s, &plds); break;
        case 12: rc = launch_lane_reg<N, 12>(m, s, &plds); break;
        default: rc = launch_lane_reg<N, 8>(m, s, &plds); break;
        }
    }
    if (s->lane_dma)
        snprintf(s->kernel_name, sizeof(s->kernel_name), "prune_lane<%d,dma,R%d,B%d,W%d>", N, R, B, W);
    else
        snprintf(s->kernel_name, sizeof(s->kernel_name), "prune_lane<%d,%s,R%d%s>", N,
                 plds ? "reg+ldsP" : "reg", R, s->rescale ? ",rescale" : "");
    *name = s->kernel_name;
    return rc;
}

template <int NT, int KS>
static int launch_mfma_inst(rt_model *m, rt_sites *s)
{
    const int lds_slots = std::max(1, s->lane_stack_slots);
    if (s->mfma_solo) {
        if constexpr (NT <= 2) {
            const int lds = 4 * lds_slots * (NT * 4 * 512);
            const unsigned grid = (unsigned)((s->nblocks + 3) / 4);
            auto kern = prune_mfma_solo_kernel<NT, KS>;
            RT_HIP(hipFuncSetAttribute((const void *)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            RT_LAUNCH_TIMED(m->ctx, kern, dim3(grid), dim3(256), lds, m->d_Pfrag,
                               (const int4_t *)s->d_lane_ops, (int)s->ops.size(), s->d_obs,
                               (int)s->nobs, m->d_root, (int)m->n, lds_slots, s->d_loglik,
                               s->d_status, s->d_partial, (long)s->nsites, (long)s->nblocks,
                               (int)s->rescale);
            return RT_OK;
        }
    }
    constexpr int WAVES = rt_split_waves(NT);
    constexpr int TILES = WAVES / NT;
    const int lds = (TILES * NT * 4 * 64 + TILES * NT * 16) * 8 + WAVES * lds_slots * 2048;
    const unsigned grid = (unsigned)((s->nblocks + TILES - 1) / TILES);
    // the instance with the two stores only when somebody asked for L and M (expect_mfma.hip)
    // observed states / small allowed sets at every leaf: the leaf steps gather (the model's
    // leaf-column table exists once such a batch does); above 32 states only (below, the
    // one-wave kernels run)
    const bool isp = NT > 2 && s->sparse_ok && s->d_leafw && m->d_Pcol && !s->d_Lout && !s->rescale &&
                     !getenv("RAOTEH_INTERP_NO_SPARSE");
    auto kern = s->d_Lout ? prune_mfma_kernel<NT, KS, true>
                : isp ? prune_mfma_kernel<NT, KS, false, (NT > 2)> : prune_mfma_kernel<NT, KS, false>;
    RT_HIP(hipFuncSetAttribute((const void *)kern,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const unsigned *ilw = isp ? s->d_leafw : nullptr;
    const double *ipc = isp ? m->d_Pcol : nullptr;
    const int ism = isp ? (s->sparse_pairs ? 2 : 1) : 0;
    rt_interp_halves hv;
    if (s->interp_halves && !s->d_Lout) {
        // the two root programs as even / odd workgroups, then the combine kernel
        hv.prog1 = (const int4_t *)s->d_lane_ops_b;
        hv.nops0 = s->half_nops[0];
        hv.nops1 = s->half_nops[1];
        hv.rec1 = s->half_rec1;
        hv.kobs1 = s->half_kobs1;
        hv.halfbuf = s->d_half;
        RT_LAUNCH_TIMED(m->ctx, kern, dim3(2 * grid), dim3(WAVES * 64), lds,
                           m->d_Pfrag, (const int4_t *)s->d_lane_ops_a, hv.nops0,
                           s->d_obs, (int)s->nobs, m->d_root, (int)m->n, lds_slots,
                           s->d_loglik, s->d_status, s->d_partial, (long)s->nsites,
                           (long)s->nblocks, s->d_Lout, s->d_Mout, hv, 0, ilw, ipc, ism);
        hipEvent_t ca = nullptr, cb = nullptr;
        rt_time_extra_begin(m->ctx, RT_K_COMBINE, "prune_mfma_combine", &ca, &cb);
        auto ckern = prune_mfma_combine_kernel<NT>;
        const double *chalf = s->d_half;
        if (ca)
            hipExtLaunchKernelGGL(ckern, dim3(grid), dim3(WAVES * 64), 0, m->ctx->stream, ca, cb, 0,
                                  chalf, (const double *)s->d_obs, (int)s->nobs, (KS + 1) / 2,
                                  s->half_kroot, (const double *)m->d_root, (int)m->n, s->d_loglik,
                                  s->d_status, s->d_partial, (long)s->nsites, (long)s->nblocks);
        else
            hipLaunchKernelGGL(ckern, dim3(grid), dim3(WAVES * 64), 0, m->ctx->stream,
                               chalf, (const double *)s->d_obs, (int)s->nobs, (KS + 1) / 2,
                               s->half_kroot, (const double *)m->d_root, (int)m->n, s->d_loglik,
                               s->d_status, s->d_partial, (long)s->nsites, (long)s->nblocks);
        rt_time_extra_end(m->ctx, RT_K_COMBINE, ca, cb);
        return RT_OK;
    }
    RT_LAUNCH_TIMED(m->ctx, kern, dim3(grid), dim3(WAVES * 64), lds,
                       m->d_Pfrag, (const int4_t *)s->d_lane_ops, (int)s->ops.size(),
                       s->d_obs, (int)s->nobs, m->d_root, (int)m->n, lds_slots,
                       s->d_loglik, s->d_status, s->d_partial, (long)s->nsites,
                       (long)s->nblocks, s->d_Lout, s->d_Mout, hv,
                       (int)(s->rescale && !s->d_Lout),      // L and M are stored unscaled
                       ilw, ipc, ism);
    return RT_OK;
}

static int launch_mfma(rt_model *m, rt_sites *s, const char **name)
{
    const int ks = ks_of(m->n);
    snprintf(s->kernel_name, sizeof(s->kernel_name), "prune_mfma%s<%d,%d%s%s>",
             s->mfma_solo ? "_solo" : "", nt_of(m->n), ks,
             s->interp_halves && !s->mfma_solo && !s->d_Lout ? ",halves" : "",
             s->rescale && !s->d_Lout ? ",rescale"
             : (m->n > 32 && s->sparse_ok && !s->mfma_solo && s->d_leafw && m->d_Pcol && !s->d_Lout &&
                !getenv("RAOTEH_INTERP_NO_SPARSE")) ? ",leaf-states" : "");
    *name = s->kernel_name;
    switch (ks) {
    case 2: return launch_mfma_inst<1, 2>(m, s);
    case 3: return launch_mfma_inst<1, 3>(m, s);
    case 4: return launch_mfma_inst<1, 4>(m, s);
    case 5: return launch_mfma_inst<2, 5>(m, s);
    case 6: return launch_mfma_inst<2, 6>(m, s);
    case 7: return launch_mfma_inst<2, 7>(m, s);
    case 8: return launch_mfma_inst<2, 8>(m, s);
    case 9: return launch_mfma_inst<3, 9>(m, s);
    case 10: return launch_mfma_inst<3, 10>(m, s);
    case 11: return launch_mfma_inst<3, 11>(m, s);
    case 12: return launch_mfma_inst<3, 12>(m, s);
    case 13: return launch_mfma_inst<4, 13>(m, s);
    case 14: return launch_mfma_inst<4, 14>(m, s);
    case 15: return launch_mfma_inst<4, 15>(m, s);
    case 16: return launch_mfma_inst<4, 16>(m, s);
    // 64 < n <= 128: one tile of NT = 5..8 waves per workgroup
    case 17: return launch_mfma_inst<5, 17>(m, s);
    case 18: return launch_mfma_inst<5, 18>(m, s);
    case 19: return launch_mfma_inst<5, 19>(m, s);
    case 20: return launch_mfma_inst<5, 20>(m, s);
    case 21: return launch_mfma_inst<6, 21>(m, s);
    case 22: return launch_mfma_inst<6, 22>(m, s);
    case 23: return launch_mfma_inst<6, 23>(m, s);
    case 24: return launch_mfma_inst<6, 24>(m, s);
    case 25: return launch_mfma_inst<7, 25>(m, s);
    case 26: return launch_mfma_inst<7, 26>(m, s);
    case 27: return launch_mfma_inst<7, 27>(m, s);
    case 28: return launch_mfma_inst<7, 28>(m, s);
    case 29: return launch_mfma_inst<8, 29>(m, s);
    case 30: return launch_mfma_inst<8, 30>(m, s);
    case 31: return launch_mfma_inst<8, 31>(m, s);
    case 32: return launch_mfma_inst<8, 32>(m, s);
    default: break;
    }
    rt_set_error("no MFMA pruning kernel for n=%lld", (long long)m->n);
    return RT_ERR_UNSUPPORTED;
}

static int launch_generic(rt_model *m, rt_sites *s, const char **name)
{
    const int n = (int)m->n;
    const int np = (n + 1) & ~1;
    const long nsp = s->nblocks * 64;
    const unsigned grid = (unsigned)s->nblocks;
#define RT_GEN(NMAX)                                                               \
    RT_LAUNCH_TIMED(m->ctx, prune_generic_kernel<NMAX>, dim3(grid), dim3(64), 0,    \
                       m->d_P, s->d_ops, (int)s->ops.size(), s->d_obs,             \
                       (int)s->nobs, n, np, m->d_root, s->d_loglik, s->d_status,   \
                       s->d_partial, s->d_scratch, nsp, (long)s->nsites, (int)s->rescale)
    if (n <= 8) { *name = "prune_generic<8>"; RT_GEN(8); }
    else if (n <= 16) { *name = "prune_generic<16>"; RT_GEN(16); }
    else if (n <= 32) { *name = "prune_generic<32>"; RT_GEN(32); }
    else if (n <= 64) { *name = "prune_generic<64>"; RT_GEN(64); }
    else { *name = "prune_generic<128>"; RT_GEN(128); }
#undef RT_GEN
    return RT_OK;
}

// the batch's previous all-reduce must have finished before its totals are rewritten
static int wait_for_collective(rt_ctx *ctx, rt_sites *s)
{
    if (s->comm_pending) {
        // usually it has, long ago, and a wait packet on the compute stream (a few us
        // of dispatch latency each) is not needed
        if (ctx->comm && hipEventQuery(s->ev_comm_done) != hipSuccess)
            RT_HIP(hipStreamWaitEvent(ctx->stream, s->ev_comm_done, 0));
        (void)hipGetLastError();             // hipErrorNotReady is not an error
        s->comm_pending = false;
    }
    return RT_OK;
}

bool rt_take_pending_reduce(rt_ctx *ctx, rt_reduce_args *out)
{
    rt_sites *s = ctx->pending_reduce;
    if (!s) return false;
    if (wait_for_collective(ctx, s) != RT_OK) return false;    // stays pending: flushed later
    ctx->pending_reduce = nullptr;
    out->partial = s->d_partial;
    out->npartials = (long)s->npartials;
    out->totals = s->d_totals;
    out->nsites = (double)s->nsites;
    return true;
}

int rt_flush_reduce(rt_ctx *ctx)
{
    rt_sites *s = ctx->pending_reduce;
    if (!s) return RT_OK;
    ctx->pending_reduce = nullptr;
    RT_TRY(wait_for_collective(ctx, s));
    hipEvent_t ev = nullptr;
    rt_time_begin(ctx, RT_K_REDUCE, "reduce_partials", &ev);
    RT_LAUNCH_TIMED(ctx, reduce_partials_kernel, dim3(1), dim3(256), 0,
                       s->d_partial, (long)s->npartials, s->d_totals,
                       (double)s->nsites);
    RT_HIP(hipGetLastError());
    rt_time_end(ctx, RT_K_REDUCE, ev);
    return RT_OK;
}

int rt_launch_prune(rt_model *m, rt_sites *s, bool defer_reduce, bool fuse_expm)
{
    rt_ctx *ctx = m->ctx;
    const char *name = "";
    hipEvent_t ev = nullptr;
    rt_fuse_args fuse;
    if (fuse_expm && s->jit_fused && s->jit_fn) {
        // one launch: transitions from the resident rates in the kernel's prologue, the pending
        // reduction (of whichever batch) in an extra workgroup.  That reduction reads the
        // partial sums of ITS batch while this launch writes its own: if it is this batch,
        // the two must be different buffers
        fuse.expm = true;
        if (!rt_take_pending_reduce(ctx, &fuse.red) && ctx->pending_reduce) RT_TRY(rt_flush_reduce(ctx));
        std::swap(s->d_partial, s->d_partial_alt);
        m->have_P = true;
        m->frag_dirty = false;        // workgroup 0 leaves the step-ordered table behind
    } else {
        fuse_expm = false;
        // a reduction still pending reads d_partial of ITS batch: if that is this batch, it
        // must run before the pruning kernel overwrites the partial sums
        if (ctx->pending_reduce) RT_TRY(rt_flush_reduce(ctx));
    }
    // which family this batch was packed for
    const bool generic = s->d_scratch != nullptr;
    if (!generic && !fuse_expm) RT_TRY(rt_launch_pfrag(m));
    rt_time_begin(ctx, RT_K_PRUNE, "", &ev);
    int rc;
    char *jit_name = s->kernel_name;
    if (generic) rc = launch_generic(m, s, &name);
    else if (s->jit_fn) {
        rc = rt_launch_prune_jit(m, s, s->jit_fused ? &fuse : nullptr);
        if (s->layout == RT_LAYOUT_LANE)
            snprintf(jit_name, sizeof(s->kernel_name), "prune_tree_jit<%d,D%d%s%s>", (int)m->n,
                     s->jit_prefetch, s->compact_states == 1 ? ",states"
                                      : s->compact_states == 2 ? ",masks" : "",
                     fuse_expm ? ",expm" : "");
        else
            if (s->jit_fn2)
                snprintf(jit_name, sizeof(s->kernel_name), "prune_tree_jit_mfma%s<%d,T%d+T%d>",
                         s->jit_quad ? "4x4" : "", (int)m->n, s->jit_tiles, s->jit_tiles2);
            else
            snprintf(jit_name, sizeof(s->kernel_name), "prune_tree_jit_mfma%s<%d,T%d%s>",
                     s->jit_quad ? "4x4" : "", (int)m->n, s->jit_tiles,
                     s->jit_sparse && s->jit_halves ? ",halves,leaf-states"
                     : s->jit_sparse ? (s->jit_pipe ? ",pipelined,leaf-states" : ",leaf-states")
                     : s->jit_halves ? ",halves" : "");
        name = jit_name;
    } else if (s->layout == RT_LAYOUT_LANE) {
        switch ((int)m->n) {
        case 1: rc = launch_lane<1>(m, s, &name); break;
        case 2: rc = launch_lane<2>(m, s, &name); break;
        case 3: rc = launch_lane<3>(m, s, &name); break;
        default: rc = launch_lane<4>(m, s, &name); break;
        }
    } else rc = launch_mfma(m, s, &name);
    if (rc != RT_OK) return rc;
    RT_HIP(hipGetLastError());
    snprintf(ctx->slots[RT_K_PRUNE].name, sizeof(ctx->slots[RT_K_PRUNE].name), "%s", name);
    if (name != s->kernel_name) snprintf(s->kernel_name, sizeof(s->kernel_name), "%s", name);
    rt_time_end(ctx, RT_K_PRUNE, ev);

    ctx->pending_reduce = s;
    // deferred (rt_step): the reduction rides on the next expm launch of this context
    if (defer_reduce && !getenv("RAOTEH_NO_DEFER_REDUCE")) return RT_OK;
    return rt_flush_reduce(ctx);
}
