// Internal declarations shared by the HIP translation units of
// libraoteh_hip.so.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/raoteh_hip.h"

// ---- error plumbing ---------------------------------------------------------

void rt_set_error(const char *fmt, ...);

#define RT_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            rt_set_error("%s failed: %s (%s:%d)", #call,                      \
                         hipGetErrorString(e_), __FILE__, __LINE__);          \
            return RT_ERR_HIP;                                                \
        }                                                                     \
    } while (0)

#define RT_REQUIRE(cond, ...)                                                 \
    do {                                                                      \
        if (!(cond)) {                                                        \
            rt_set_error(__VA_ARGS__);                                        \
            return RT_ERR_INVALID;                                            \
        }                                                                     \
    } while (0)

#define RT_TRY(call)                                                          \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != RT_OK) return rc_;                                         \
    } while (0)

struct rt_jit_job;      // a background compile (jit.hip)

// ---- handles ------------------------------------------------------------------

struct rt_timing_slot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
    double total_ms = 0.0;
    int64_t launches = 0;
    int64_t seen = 0;
    char name[64] = "";        // variant that ran last (owned by the slot: no statics)
};

static const int RT_TOTALS_SLOTS = 4096;   // site batches per context sharing the arena

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    bool timing = false;
    int timing_period = 1;
    rt_timing_slot slots[RT_K_COUNT];
    // events of the launch being timed (rt_time_begin .. rt_time_end), else null
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // totals of every site batch of this context live in one arena ([slot][3] doubles)
    // so that the totals of several batches can be all-reduced by ONE collective
    double *d_totals_arena = nullptr;
    std::vector<int> totals_free;          // free slots, highest index last popped first
    hipEvent_t ev_reduced = nullptr;       // compute stream: totals of a group written
    hipEvent_t comm_events[8] = {};        // comm stream: a collective finished (ring)
    int comm_event_next = 0;
    // grow-only device scratch of the reference-format calls (no hipMalloc / hipFree
    // per call: single-site calls are latency-bound)
    unsigned char *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void *comm = nullptr;          // ncclComm_t
    hipStream_t comm_stream = nullptr;   // collectives overlap the next step's kernels
    void *rccl = nullptr;          // dlopen handle
    // options snapshotted by rt_sites_create (rt_ctx_set_option); RT_OPT_UNSET = the
    // process-wide default of rt_set_option applies
    int opt_force_generic = -2;
    int opt_jit = -2;
    int opt_jit_block_sites = -2;
    int opt_jit_async = -2;
    int opt_rescale = -2;
    int opt_leaf_state_kernels = -2;
    // The batch whose per-wave partial sums still await their fixed-order reduction.
    // rt_step defers it: the reduction of step j rides as one extra workgroup of step
    // j + 1's expm launch (two launches per step instead of three: on config 2 the two
    // launch-latency-sized kernels were a quarter of the step); every reader of the
    // totals, rt_prune, rt_ctx_sync and the destructors flush it (rt_flush_reduce).
    struct rt_sites *pending_reduce = nullptr;
    size_t expm_attr_lds = 0;      // dynamic-LDS attribute already granted to expm_kernel
    size_t expm_ts_attr_lds[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ... to the Taylor kernels (per NT)
    size_t expm_split_attr_lds = 0;                           // ... to the two-workgroup form
    size_t expm_wide_attr_lds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double *d_expm_scratch = nullptr;   // matrices of the order > 64 Taylor kernel (grow-only)
    void *expect_cache = nullptr;       // expect_mfma.hip: model + packed batch of the last call
    hipStream_t stream2 = nullptr;      // side stream of two-kernel pruning launches (lazy)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    size_t expm_scratch_bytes = 0;
    // Lifetime (include/raoteh_hip.h, "ownership and destruction order"): a context cannot
    // go while one of its models or chain batches lives, a model not while one of its site
    // batches lives -- the destroy call refuses (RT_ERR_INVALID) instead of leaving the
    // children with a dangling parent.  Counts of live children, own internal objects (the
    // expectation cache, probe batches) included.
    int live_models = 0, live_chains = 0;
};
static const int RT_OPT_UNSET = -2;

// One schedule step of the fast pruning kernels: a node of the tree visited in
// post-order.  For a non-root node v the step forms L_v (pop its accumulator if
// internal, multiply its observation in), t = P_v * L_v and folds t into its
// parent's accumulator slot; the final step (dst = -1) is the root reduction.
static const int RT_FAST_MAX_DEPTH = 16;   // LDS accumulator stack of the fast kernels

struct rt_op {
    int32_t node;      // preorder index of v (also the P slot of edge parent->v)
    int32_t obs;       // position of v in the observation stream, or -1
    int32_t pop;       // accumulator slot of v (internal node), or -1 (leaf)
    int32_t dst;       // parent's accumulator slot | (first child ? 256 : 0); -1 root
};

struct rt_model {
    rt_ctx *ctx = nullptr;
    int64_t nnodes = 0;
    int64_t n = 0;
    std::vector<int64_t> indices, indptr;   // host copy of the CSR
    std::vector<int32_t> parent;            // preorder parent index, -1 root
    std::vector<rt_op> ops;                 // post-order schedule
    std::vector<int32_t> h_qidx;            // what d_qidx / d_t hold (rt_model_set_rates)
    std::vector<double> h_t;
    int max_depth = 0;                      // accumulator slots needed
    // device
    int64_t *d_indices = nullptr, *d_indptr = nullptr;
    rt_op *d_ops = nullptr;
    double *d_P = nullptr;          // [nnodes][n][n] esd_transitions
    double *d_Pfrag = nullptr;      // MFMA A-fragment order (n > 4)
    // 4 < n <= 32: the same matrices as 4 x 4 blocks [step][row quad][col quad][k][i] =
    // P[4 rq + i][4 kk + k], what the v_mfma_f64_4x4x4_4b kernels of jit.hip read (the
    // four blocks of that instruction share A: 4 rows x 16 sites per instruction, so 20
    // states cost 5 x 5 instructions of 16 cycles instead of 2 x 5 of 64)
    double *d_Pquad = nullptr;
    // leaf-column table for the split-M leaf-state kernels (jit.hip, `sparse`): column s of the
    // step's P with the four rows a lane owns adjacent, Pcol[rec][s][m][lane >> 4][r] =
    // P[16 m + 4 r + (lane >> 4)][s]; allocated when the first such batch is created, rewritten
    // whenever the transition matrices change
    double *d_Pcol = nullptr;
    double *d_root = nullptr;       // [n] root weights (ones if unset)
    // rt_model_set_rates_spectral: A [n][n], B [n][n], lam [n], D [n] of the decomposition
    // (spectral.hip); while `spectral` is set the transitions are rebuilt from these
    double *d_spec = nullptr;
    bool spectral = false, spectral_has_D = false;
    double *d_Q = nullptr;          // rate matrices of the last set_rates
    int64_t q_capacity = 0;
    int32_t *d_qidx = nullptr;      // [nnodes]
    double *d_t = nullptr;          // [nnodes]
    int32_t *d_info = nullptr;      // [nnodes][2]
    int32_t *d_step_of_node = nullptr;  // [nnodes] schedule step of each node
    // the same rate-matrix indices / branch lengths in STEP order (the root's step: -1, 0): a
    // lane kernel that computes its own transitions reads them at its thread index, with no
    // step -> node lookup in front of the loads (n <= 4 only)
    int32_t *d_qidx_step = nullptr;
    double *d_t_step = nullptr;
    bool have_P = false;
    bool frag_dirty = true;
    int live_batches = 0;           // site batches created from this model and not yet destroyed
    void *expect_state = nullptr;   // expect_mfma.hip: device buffers of rt_expect_step (lazy)
    void *expect_lane_state = nullptr;  // passes.hip: its n <= 8 form (plan, pattern bits)
};

// Device layouts of a site batch:
//  RT_LAYOUT_LANE  (n <= 4)  [block of 64 sites][obs slot][lane][np] f64,
//                  np = n rounded up to even: one lane owns one site
//  RT_LAYOUT_MFMA  (n  > 4)  [block of 16 sites][obs slot][k-step pair][lane][2]
//                  f64 in v_mfma_f64_16x16x4 B-operand order
enum { RT_LAYOUT_LANE = 0, RT_LAYOUT_MFMA = 1 };

struct rt_sites {
    rt_model *model = nullptr;
    int64_t nsites = 0;
    int64_t nobs = 0;
    int layout = RT_LAYOUT_LANE;
    bool lane_dma = false;          // lane family: LDS-DMA ring instead of VGPR ring
    int lane_ring = 8;              // ring depth of the lane kernel
    bool mfma_solo = false;         // MFMA family, n <= 32: one wave owns a tile outright
    int64_t nblocks = 0;            // site blocks (64 or 16 sites each)
    int64_t obs_bytes = 0;
    std::vector<int32_t> node_obs;  // per node: stream position or -1
    std::vector<rt_op> ops;         // model ops with .obs filled in
    rt_op *d_ops = nullptr;
    int32_t *d_lane_ops = nullptr;  // lane-kernel program (int32[nprog][4])
    int lane_stack_slots = 1;       // LDS accumulator slots the program touches
    int64_t lane_nprog = 0;         // its entries (< steps when cherries are fused)
    double *d_obs = nullptr;
    double *d_loglik = nullptr;     // [nblocks * sites per block]
    int32_t *d_status = nullptr;
    double *d_partial = nullptr;    // [npartials][2] (sum, nzero)
    int64_t npartials = 0;
    double *d_totals = nullptr;     // [3]: a slot of the context's arena (or its own)
    int totals_slot = -1;           // arena slot, -1: d_totals is its own allocation
    hipEvent_t ev_comm_done = nullptr;  // ctx-owned: the collective that last touched the totals
    bool comm_pending = false;
    void *jit_fn = nullptr;         // hipFunction_t of the tree-specialised kernel (jit.hip)
    int jit_prefetch = 0;           // its prefetch distance (stream positions)
    int jit_lookahead = 1;          // ... and P records requested ahead
    int block_sites = 64;           // lane family: sites per block (< 64 only with jit_fn)
    int jit_waves = 1;              // waves per workgroup of the tree-specialised kernel
    int jit_tiles = 1;              // MFMA family: site tiles per wave of that kernel
    // one-wave MFMA family, batches a little over a whole number of tiles per SIMD: the
    // first jit_split_tiles tiles go to jit_fn (one wave per SIMD), the rest to jit_fn2
    // (jit_tiles2 = 1 tile per wave) on the context's second stream, side by side
    void *jit_fn2 = nullptr;
    int jit_tiles2 = 1;
    int64_t jit_split_tiles = 0;
    bool jit_quad = false;          // ... built on v_mfma_f64_4x4x4_4b (reads d_Pquad)
    // split-M family, root halves (jit.hip): the two root programs run as the even / odd
    // workgroups of jit_fn and leave their share of the root's accumulator in d_half
    // ([tile][half][k-step][lane]); jit_combine (same module) finishes the sites
    // lane family: the kernel can compute the transition matrices of a step in its own
    // prologue (jit.hip, `fuse`); rt_step then launches ONE kernel.  The batch sum of step j
    // rides as an extra workgroup of step j + 1's launch while that launch writes its own
    // partial sums: two buffers, swapped at every such launch
    bool jit_fused = false;
    double *d_partial_alt = nullptr;
    bool jit_halves = false;
    void *jit_combine = nullptr;
    double *d_half = nullptr;
    // folded form: the second workgroup of a pair to arrive finishes the pair's tiles (no
    // combine launch); d_half_count: one arrival counter per pair, zero between launches
    bool jit_fold = false;
    int *d_half_count = nullptr;
    // the same cut for the split-M INTERPRETER kernel (prune.hip rt_interp_halves), used while
    // the batch has no tree-specialised kernel: the two root programs, the P record and the
    // stream position the second one starts at, the root's own stream position (-1: none)
    bool interp_halves = false;
    // "rescale" option at creation: messages rescaled by powers of two, exponent per site
    // (interpreter kernels only; such a batch never gets a tree-specialised kernel)
    bool rescale = false;
    int32_t *d_lane_ops_a = nullptr, *d_lane_ops_b = nullptr;
    // expectation path (expect_mfma.hip): per step {slot of the parent's D, own slot or -1}
    int32_t *d_down_meta = nullptr;
    int down_slots = 0;
    int half_nops[2] = {0, 0};
    int half_rec1 = 0, half_kobs1 = 0, half_kroot = -1;
    // not owned: where the split-M interpreter kernel leaves L_v and M_v of every step
    // (expect_mfma.hip sets them around its own launch)
    double *d_Lout = nullptr, *d_Mout = nullptr;
    int compact_states = 0;         // lane family + specialised kernel: the batch stays resident
                                    // as one byte per leaf: 1 = uint8 states, 2 = allowed-set masks
    double *d_scratch = nullptr;    // generic kernel message stack
    int64_t scratch_bytes = 0;
    // background compile of the tree-specialised kernel (MFMA family): the batch runs the
    // interpreter kernel until the job is done and rt_sites_jit_poll swaps the kernel in
    std::shared_ptr<rt_jit_job> jit_job;
    struct jit_cand { int T; bool halves; bool quad; bool sparse = false; bool pipe = false; };
    std::vector<jit_cand> jit_cands;        // what each candidate source of the job was built with
    std::vector<std::string> jit_srcs;
    int jit_kind = 0;                       // observation kind of the batch (probe batches)
    // lane family (n <= 4): the kernel's resident layout differs from the interpreter's, so
    // while the job runs the batch keeps the caller's observations on the device (d_raw, in
    // the caller's format) and packs them again at the switch; what the kernel was built for:
    bool keep_raw = false;
    void *d_raw = nullptr;
    int *d_raw_src = nullptr;
    struct { int S = 64, WG = 1, D = 1, LA = 1, compact = 0; bool fuse = false; } jit_lane;
    // n > 32, observed STATES at leaves only, none unobserved: next to the dense image the
    // batch keeps the state bytes (leafw[tile][ceil(K/4)][16 sites], four stream positions per
    // word) for the tree-specialised kernel whose leaf steps gather columns of P (jit.hip)
    bool sparse_ok = false;
    bool sparse_pairs = false;       // ... as allowed sets of one or two states (RT_OBS_MASK): 16 bits per leaf
    bool jit_sparse = false;
    bool jit_pipe = false;                  // ... from the pipelined generator (leaves as factors)
    unsigned *d_leafw = nullptr;
    // rt_expect_step: per-site multiplicities on the device (null: ones), and the batch's
    // split-M interpreter twin -- its own program, partial sums and per-site outputs over the
    // SAME resident observations (obs_borrowed: d_obs belongs to the batch it was made from)
    double *d_weights = nullptr;
    unsigned char *d_sets = nullptr;        // n <= 8: allowed sets [node][site] (built once)
    rt_sites *expect_twin = nullptr;
    bool obs_borrowed = false;
    bool counted = false;           // this batch is in its model's live_batches
    char kernel_name[64] = "";      // the pruning kernel variant of this batch
    double jit_compile_s = 0.0;     // hiprtc time spent for this batch (0: cache hit / none)
};

// ---- internal entry points ---------------------------------------------------------

void rt_time_begin(rt_ctx *ctx, int kernel, const char *name, hipEvent_t *start);
void rt_time_end(rt_ctx *ctx, int kernel, hipEvent_t start);
// a second kernel inside a launch that is being sampled (ctx->ev_start set): its own pair of
// events in its own slot; false = not timed (a, b stay null)
bool rt_time_extra_begin(rt_ctx *ctx, int kernel, const char *name, hipEvent_t *a, hipEvent_t *b);
void rt_time_extra_end(rt_ctx *ctx, int kernel, hipEvent_t a, hipEvent_t b);

// the reduction a launch may carry in one extra workgroup (partial == nullptr: none)
struct rt_reduce_args {
    const double *partial = nullptr;   // [npartials][2]
    long npartials = 0;
    double *totals = nullptr;          // [3]
    double nsites = 0.0;
};
// expm_wide.hip: the LDS-resident Taylor kernel for 64 < n <= 128 (scratch: ctx->d_expm_scratch)
int rt_expm_wide_launch(rt_ctx *ctx, int nt, bool split2, size_t grid, int64_t n, const double *d_Q,
                        const int *d_qidx, const double *d_t, double *d_P, int *d_info,
                        const int *d_step_of_node, int frag_kind, double *d_Pfrag,
                        const rt_reduce_args &red);
int rt_launch_expm(rt_ctx *ctx, int64_t n, int64_t count, const double *d_Q,
                   const int32_t *d_qidx, const double *d_t, double *d_P,
                   int32_t *d_info, const int32_t *d_step_of_node, int frag_kind,
                   double *d_Pfrag, const rt_reduce_args *fused_reduce = nullptr,
                   double *d_Pquad = nullptr);
// spectral.hip: P_e = A diag(exp(lam t_e)) B per edge, outputs as rt_launch_expm's
int rt_launch_spectral(rt_ctx *ctx, int64_t n, int64_t count, const double *d_A,
                       const double *d_lam, const double *d_B, const double *d_D,
                       const int32_t *d_qidx, const double *d_t, double *d_P, int32_t *d_info,
                       const int32_t *d_step_of_node, int frag_kind, double *d_Pfrag,
                       const rt_reduce_args *fused_reduce = nullptr, double *d_Pquad = nullptr);
// the pending reduction, handed to the next expm launch (-> true) ...
bool rt_take_pending_reduce(rt_ctx *ctx, rt_reduce_args *out);
// ... or launched on its own now (no-op when nothing is pending)
int rt_flush_reduce(rt_ctx *ctx);
// A timed launch: the runtime stamps the two events with the kernel's own begin and
// end (what rocprofv3's kernel trace reports), not with the stream position of an
// event record, which adds 2-3 us per pair.  Null events: a plain launch.
#define RT_LAUNCH_TIMED(ctx, kern, grid, block, lds, ...)                               \
    do {                                                                                \
        if ((ctx)->ev_start)                                                            \
            hipExtLaunchKernelGGL(kern, grid, block, lds, (ctx)->stream,                \
                                  (ctx)->ev_start, (ctx)->ev_stop, 0, __VA_ARGS__);     \
        else                                                                            \
            hipLaunchKernelGGL(kern, grid, block, lds, (ctx)->stream, __VA_ARGS__);     \
    } while (0)
int rt_launch_pfrag(rt_model *m);
int rt_model_pack_pcol(rt_model *m);            // (no-op without d_Pcol)
int rt_model_need_pcol(rt_model *m);            // allocate + fill from the current d_P
// fuse_expm: the pruning launch computes the transitions from the resident rates itself
// (batches with jit_fused only) and carries the pending reduction
int rt_launch_prune(rt_model *m, rt_sites *s, bool defer_reduce = false, bool fuse_expm = false);
// what such a launch is handed besides the pruning arguments
struct rt_fuse_args {
    rt_reduce_args red;     // partial == nullptr: nothing carried
    bool expm = false;      // compute the transitions (else copy the resident table)
};
// jit.hip
std::string rt_jit_lane_source(const std::vector<rt_op> &ops, int n, int K, int D, int LA,
                               int S, int WG, int compact = 0, bool fuse = false);
std::string rt_jit_mfma_source(const std::vector<rt_op> &ops, int n, int K, int T, int D, int LA,
                               bool quad = false, int sparse = 0);
// (sparse: 0 dense leaf vectors, 1 one observed state per leaf, 2 one or two allowed states)
std::string rt_jit_mfma_split_source(const std::vector<rt_op> &ops, int n, int K, int T, int D,
                                     int LA, int sparse = 0);
bool rt_jit_fold_enabled();
std::string rt_jit_mfma_split_pipelined_source(const std::vector<rt_op> &ops, int n, int K, int T,
                                               int D, int LA, bool halves = false,
                                               int sparse = 0);
// steps of the two root programs the halves form would run (0, 0: the root has < 2 children)
void rt_jit_root_halves(const std::vector<rt_op> &ops, int *stepsA, int *stepsB);
// the cut itself: A = the subtrees of all children of the root but the last, B = the last
// child's subtree, each followed by the root's step (both contiguous runs of `ops`)
bool rt_split_at_root(const std::vector<rt_op> &ops, std::vector<rt_op> *A, std::vector<rt_op> *B);
// another kernel of the module `fn` came from (the halves form's rt_jit_combine)
int rt_jit_companion(const rt_ctx *ctx, void *fn, const char *name, void **out);
int rt_jit_get(rt_ctx *ctx, const std::string &src, void **fn, bool mfma = false,
               double *compile_s = nullptr);
// background compilation (jit.hip): a host thread of this process works through candidate
// sources (the first that compiles without scratch wins) and leaves the kernel in the
// per-context cache; rt_jit_job_done polls (or joins), rt_jit_join_all before the context goes
std::shared_ptr<rt_jit_job> rt_jit_start(rt_ctx *ctx, std::vector<std::string> sources, bool mfma);
bool rt_jit_job_done(rt_jit_job *job, bool wait);
void rt_jit_job_result(rt_jit_job *job, int *rc, int *chosen, double *seconds, std::string *error);
void rt_jit_join_all(const rt_ctx *ctx);
int rt_jit_jobs_pending();
// (ctx, src): 1 compiled and usable, -1 compiled and rejected, 0 unknown (no reference taken)
int rt_jit_cached(const rt_ctx *ctx, const std::string &src);
// api.hip: swap the finished background kernel of a batch in (wait: join the job first)
int rt_sites_jit_poll(rt_sites *s, bool wait);
void rt_jit_ref(const rt_ctx *ctx, void *fn, int delta);
// every freshly compiled kernel is run once against the interpreter kernel on a probe
// batch before a user batch may launch it (api.hip verify_jit_kernel): 0 = not yet,
// 1 = verified; rt_jit_set_verified(false) rejects the kernel for good
int rt_jit_verified(const rt_ctx *ctx, void *fn);
void rt_jit_set_verified(const rt_ctx *ctx, void *fn, bool ok);
void rt_jit_release(const rt_ctx *ctx);
int rt_jit_read_global(const rt_ctx *ctx, void *fn, const char *name, void *dst, size_t bytes);
int rt_launch_prune_jit(rt_model *m, rt_sites *s, const rt_fuse_args *fuse = nullptr);
// a batch that runs the interpreter kernels only (no tree-specialised kernel is compiled)
int rt_sites_create_interpreter(rt_model *m, int64_t nsites, int kind, int64_t nobs,
                                const int64_t *obs_nodes, const void *data, rt_sites **out);
// drops what rt_expectation_weights_mfma keeps between calls (rt_ctx_destroy)
void rt_expect_cache_release(rt_ctx *ctx);
// expect.hip: assemble / exponentiate / contract the Frechet blocks on device-resident operands
int rt_frechet_statistics_device(rt_ctx *ctx, int64_t n, int64_t nedges, const double *dQ,
                                 const int32_t *dqidx, const double *dt, const double *dW,
                                 double *dB, double *dE, double *dscale, const double *dones,
                                 const int32_t *dident, double *ddwell, double *dtrans);
// ... and what rt_expect_step keeps with a model (rt_model_destroy)
void rt_expect_state_release(rt_model *m);
// passes.hip: the n <= 8 form of rt_expect_step's passes (W and status on the device)
int rt_expect_lane_resident(rt_model *m, rt_sites *s, double *d_W, int *d_status);
void rt_expect_lane_release(rt_model *m);
// a split-M interpreter batch over the resident observations of `src` (api.hip)
int rt_sites_twin_interpreter(rt_sites *src, rt_sites **out);
// the context's grow-only device scratch (ctx->d_scratch) holds at least `bytes` afterwards
int rt_scratch_reserve(rt_ctx *ctx, size_t bytes);
// expectation path on the matrix pipe (expect_mfma.hip); RT_ERR_UNSUPPORTED = not this case
int rt_expectation_weights_mfma(rt_ctx *ctx, int64_t nnodes, int64_t n, int64_t nsites,
                                const int64_t *idx, const int64_t *ptr, const double *esd,
                                const double *root_distn, int64_t nobs, const int64_t *obs_nodes,
                                int kind, const void *data, const double *site_weights,
                                double *edge_weights, int32_t *status);
int rt_sites_pack(rt_sites *s, int kind, const int64_t *obs_order,
                  const void *data);
int rt_sites_pack_device(rt_sites *s, int kind, const void *d_in, const int *d_src);

static inline int64_t rt_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// doubles per step of the quad-block table (rt_model::d_Pquad): ceil(n/4)^2 blocks of 16,
// rounded up to whole 1 KiB wave loads (64 lanes x 16 bytes)
__host__ __device__ static inline int rt_quad_stride(int n)
{
    const int ks = (n + 3) / 4;
    return (ks * ks * 16 + 127) / 128 * 128;
}
