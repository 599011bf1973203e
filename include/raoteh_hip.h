/*
 * raoteh_hip.h -- C ABI of libraoteh_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the tree-CTMC likelihood hot path of
 * argriffing/raoteh.  Plain pointers and sizes only; the caller owns every
 * host buffer; the library never keeps a host pointer after a call returns;
 * device memory is owned by the opaque handles.  No exception crosses the
 * ABI: every function returns 0 (RT_OK) or a negative RT_ERR_* code and
 * rt_last_error() returns a human-readable message for the calling thread.
 *
 * Ownership and destruction order.  A site batch (rt_sites) refers to its model, a model and
 * a chain batch (rt_chains) refer to their context, until they are destroyed.  Children go
 * first: rt_model_destroy returns RT_ERR_INVALID (and destroys nothing) while a batch created
 * from the model is alive, rt_ctx_destroy while a model or a chain batch of the context is
 * alive; rt_last_error() names what is left.  Destroying NULL is a no-op.  Using a handle
 * after its destroy call returned RT_OK is undefined.
 *
 * "Reference" citations are file:line under the reference repository root
 * (argriffing/raoteh).  The reference's only native boundary on this path is
 * the third-party Cython module `pyfelscore` (absent from the reference
 * tree, version unpinned, README.md:8-9); section 1 below replaces exactly
 * the pyfelscore entry points the path calls plus the scipy expm call.
 * Section 2 is the batched, device-resident form of the same path.
 *
 * All floating point data is IEEE binary64; all index data int64 (as the
 * reference passes, _density.py:138-139, _mcy_dense.py:49); row-major,
 * C-contiguous.
 */
#ifndef RAOTEH_HIP_H
#define RAOTEH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK                0
#define RT_ERR_INVALID      -1   /* bad argument (shape, null, range)        */
#define RT_ERR_HIP          -2   /* a HIP runtime call failed                */
#define RT_ERR_UNSUPPORTED  -3   /* valid input outside implemented limits   */
#define RT_ERR_NOMEM        -4   /* host or device allocation failed         */
#define RT_ERR_RCCL         -5   /* RCCL missing or a collective failed      */
#define RT_ERR_SINGULAR     -6   /* expm: non-finite Q t / singular Pade den. */
#define RT_ERR_ZERO_PROB    -7   /* a chain's observations have likelihood 0 */

#define RT_MAX_STATES      128   /* pruning kernels and the reference-format passes
                                    (pset / set / pmap / distn / joint): n <= 128 --
                                    examples/p53/liwen.py:599-621 runs _mcy_dense on
                                    2 x 61 = 122 compound states                */
#define RT_MAX_EXPECT_STATES 64  /* expectation path (rt_mjp_*), Rao-Teh forest /
                                    chains, spectral reconstruction: n <= 64    */
#define RT_MAX_EXPM_STATES 128   /* expm: n <= 64 LDS-resident, <= 128 through
                                    L2-resident scratch (the Frechet blocks of the
                                    61-state codon model have order 122)        */

/* per-site status written by rt_prune (reference: StructuralZeroProb raised
 * by _mc0_dense.py:190-203 when the root likelihood is zero)                */
#define RT_SITE_OK           0
#define RT_SITE_ZERO_PROB    1
#define RT_SITE_NEGATIVE     4   /* bit: a negative root pmap entry was
                                    clamped (reference warns + clamps,
                                    _mc0_dense.py:184-189)                   */

/* observation encodings for rt_sites_create                                 */
#define RT_OBS_DENSE         0   /* f64 [nsites][nobs][n] likelihood vectors
                                    (type z, _mcz.py:159-160; 0/1 masks and
                                    one-hot vectors are special cases)       */
#define RT_OBS_STATE         1   /* uint8 [nsites][nobs] observed state,
                                    255 = unobserved (type x, _mcx.py:12-23) */
#define RT_OBS_MASK          2   /* uint64 [nsites][nobs][ceil(n/64)]: bit s % 64
                                    of word s / 64 = state s allowed (type y,
                                    _mcy.py:12-16); one word per node for n <= 64 */

/* kernel ids for rt_ctx_kernel_time                                         */
#define RT_K_EXPM            0
#define RT_K_PRUNE           1
#define RT_K_REDUCE          2
#define RT_K_COMBINE         3   /* second kernel of a root-halves pruning launch */
#define RT_K_COUNT           4

typedef struct rt_ctx   rt_ctx;     /* one per GPU / process                 */
typedef struct rt_model rt_model;   /* tree + transition matrices on device  */
typedef struct rt_sites rt_sites;   /* one resident batch of site data       */

/* ---- 0. library / context ------------------------------------------------ */

int         rt_version(void);                 /* 100*major + minor            */
const char *rt_last_error(void);              /* thread-local, never NULL     */
int         rt_device_count(int *count);
int         rt_ctx_create(int device, rt_ctx **ctx);
int         rt_ctx_destroy(rt_ctx *ctx);
int         rt_ctx_sync(rt_ctx *ctx);         /* wait for the ctx stream      */
/* Optional per-kernel HIP-event timing (events are recorded on the stream
 * the kernels are launched on).  rt_ctx_kernel_time drains finished events
 * and returns the accumulated device time and launch count since the last
 * reset; `name` receives a static string with the kernel variant used.      */
int         rt_ctx_set_timing(rt_ctx *ctx, int enabled); /* N > 1: every N-th launch */
int         rt_ctx_reset_timing(rt_ctx *ctx);
int         rt_ctx_kernel_time(rt_ctx *ctx, int kernel, double *total_ms,
                               int64_t *launches, const char **name);

/* ---- 1. reference-shaped entry points (host pointers in and out) --------- */

/* P[b] = expm(Q[q_index[b]] * t[b]) for b in [0,count).  Replaces
 * scipy.linalg.expm(Q * weight) at _mjp_dense.py:24-25 (called once per edge
 * at _mjp_dense.py:352-358) and pyfelscore.get_tolerance_rate_matrix(t,Q,P)
 * (_tmjp_dense.py:239, tests/test_expm.py:38).  q_index may be NULL
 * (then matrix b uses Q[b] if nq == count, or Q[0] if nq == 1).
 * info (optional, int32[count][2]) receives the Pade degree and the number
 * of squarings used.  n <= RT_MAX_EXPM_STATES.                              */
int rt_expm(rt_ctx *ctx, int64_t n, int64_t count,
            const double *Q, int64_t nq, const int64_t *q_index,
            const double *t, double *P, int32_t *info);
/* getp_spectral_v2 (examples/p53/qtop.py:76-88) at `count` branch lengths in one launch:
 * P[b] = A diag(exp(lam t[b])) B, diagonal 1 where D == 0 (D may be NULL).  n <= 64.    */
int rt_expm_spectral(rt_ctx *ctx, int64_t n, int64_t count, const double *A,
            const double *lam, const double *B, const double *D, const double *t, double *P);

/* pyfelscore.get_lb_transition_matrix(t, Q, P) (examples/p53/liwen.py:45; pure-Python twin
 * getp_lb, liwen.py:47-82) at `count` interval lengths in one launch: the lower bound of
 * expm(Q t) that keeps the histories with no change (diagonal: exp(t Q[a][a])) or exactly one
 * change a -> b (rab (exp(-ra t) - exp(-rb t)) / (rb - ra), or rab t exp(-rb t) when ra == rb;
 * ra = -Q[a][a]).  Q f64[n][n] with diagonal = minus the row sum, t f64[count],
 * P f64[count][n][n] out.  Any n.                                                       */
int rt_lb_transition_matrix(rt_ctx *ctx, int64_t n, int64_t count, const double *Q,
            const double *t, double *P);
/* pyfelscore.tmjp_get_inhomogeneous_mjp (_tmjp_dense.py:1039-1054; sparse twin
 * _tmjp.get_inhomogeneous_mjp, _tmjp.py:863-900), same argument order: the tree CSR, the
 * primary state of the edge above every node (int64[nnodes], entry 0 ignored),
 * primary_to_part int64[nprimary], Q_primary f64[nprimary][nprimary], the two tolerance
 * rates and the class under consideration; node_to_allowed_tolerances int64[nnodes][2]
 * (ones on entry; column 0 cleared where the class of an adjacent edge's state is the class
 * under consideration) and tol_rate_matrices f64[nnodes][3][3] (keyed by the child index,
 * diagonal = minus the row sum; the root's slot zero) are filled.  Host only: no context. */
int rt_tmjp_get_inhomogeneous_mjp(int64_t nnodes, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const int64_t *edge_to_primary_state,
            int64_t nprimary, const int64_t *primary_to_part, const double *Q_primary,
            double rate_on, double rate_off, int64_t tolerance_class,
            int64_t *node_to_allowed_tolerances, double *tol_rate_matrices);

/* The three pyfelscore passes of _mcy_dense.py:261-291, batched over
 * `nsites` independent sites that share the tree and the transitions:
 *   tree_csr_indices int64[nnodes-1], tree_csr_indptr int64[nnodes+1]:
 *       children CSR in DFS-preorder index space (_density.py:104-140)
 *   esd_transitions  f64[nnodes][n][n], slot = CHILD preorder index, root
 *       slot ignored (_density.py:143-180)
 *   state_mask       int64[nsites][nnodes][n], 0/1
 *   subtree_probability f64[nsites][nnodes][n], every entry written.
 * nsites == 1 is the exact single-call shape of the reference.              */

/* pyfelscore.mcy_esd_get_node_to_pset (call sites _mcy_dense.py:168,270,
 * _mcx_dense.py:145, _mcy.py:219,309,508): backward boolean pass, in place. */
int rt_mcy_esd_get_node_to_pset(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            int64_t *state_mask);

/* pyfelscore.esd_get_node_to_set (_mcy_dense.py:175,277, _mcx_dense.py:152,
 * _mcy.py:226,515): forward boolean pass, in place.                         */
int rt_esd_get_node_to_set(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            int64_t *state_mask);

/* pyfelscore.mcy_esd_get_node_to_pmap (_mcy_dense.py:184,286,
 * _mcx_dense.py:161, _mcy.py:533): Felsenstein upward pass
 *   L[v,s] = mask[v,s] * obs_lik[v,s] * prod_c sum_s' P_c[s,s'] L[c,s'].
 * obs_likelihood (optional, f64[nsites][nnodes][n]) is the type-z factor of
 * _mcz.py:159-160; pass NULL for the plain type-x/y pass.                   */
int rt_mcy_esd_get_node_to_pmap(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            const int64_t *state_mask, const double *obs_likelihood,
            double *subtree_probability);
/* The three passes above in one call (the sequence _mcy_dense.py:261-291 runs:
 * pset, set, pmap; with obs_likelihood the type-z upward pass, _mcz.py:128-163):
 * state_mask is updated in place by the two boolean passes, subtree_probability
 * is filled; one upload of the tree and the matrices instead of three.        */
int rt_mcy_esd_passes(rt_ctx *ctx, int64_t nnodes, int64_t n, int64_t nsites,
            const int64_t *tree_csr_indices, const int64_t *tree_csr_indptr,
            const double *esd_transitions, int64_t *state_mask,
            const double *obs_likelihood, double *subtree_probability);

/* pyfelscore.mc0_esd_get_node_to_distn (_mc0_dense.py:381, _mcy_dense.py:195;
 * pure-Python twin _mc0_dense.py:446-486): downward pass, posterior marginal
 * state distribution of every node given the upward messages.  root_distn
 * f64[n] or NULL (= ones).  node_to_distn_array f64[nsites][nnodes][n] out.
 * status (optional) int32[nsites]: 2 where a normalising denominator is zero
 * (the reference raises NumericalZeroProb, _util.py:164-165).                */
int rt_mc0_esd_get_node_to_distn(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            const double *root_distn, const double *subtree_probability,
            double *node_to_distn_array, int32_t *status);

/* pyfelscore.mc0_esd_get_joint_endpoint_distn (_mcy_dense.py:205; twin
 * _mc0_dense.py:246-267): joint (parent state, child state) posterior of every
 * edge, f64[nsites][nnodes][n][n] keyed by the child index (root slot zero). */
int rt_mc0_esd_get_joint_endpoint_distn(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            const double *subtree_probability, const double *node_to_distn_array,
            double *joint_distns);

/* Site sums for _mjp_dense.get_expected_history_statistics (_mjp_dense.py:458-475,
 * 497-533): the upward passes (:261-291 of _mcy_dense.py), the downward pass
 * (mc0_esd_get_node_to_distn) and, per edge, the site sum of J / P over the nonzero
 * entries of the joint endpoint posterior J (what the reference contracts with its
 * Frechet derivatives, one site at a time) in one call; only n*n numbers per edge
 * leave the device.  state_mask int64[nsites][nnodes][n] as for the passes (input
 * only), site_weights f64[nsites] or NULL (= ones), root_distn f64[n] or NULL.
 * edge_weights f64[nnodes][n][n] out, keyed by the child index; slot 0 carries the
 * weighted sum of the root posteriors in its column 0.  status as above.        */
int rt_mjp_esd_expectation_weights(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            const double *root_distn, const int64_t *state_mask,
            const double *site_weights, double *edge_weights, int32_t *status);

/* The same with the observations in the compact encodings of rt_sites_create instead
 * of the reference's int64 mask array (8 n bytes per site and NODE): kind RT_OBS_STATE,
 * data uint8[nsites][nobs] (a value >= n = unobserved), or RT_OBS_MASK, data
 * uint64[nsites][nobs]; obs_nodes int64[nobs] are preorder indices, every other node
 * is unrestricted.  The mask array is built on the device.                      */
int rt_mjp_esd_expectation_weights_obs(rt_ctx *ctx, int64_t nnodes, int64_t n,
            int64_t nsites, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *esd_transitions,
            const double *root_distn, int64_t nobs, const int64_t *obs_nodes, int kind,
            const void *data, const double *site_weights, double *edge_weights,
            int32_t *status);

/* The rest of _mjp_dense.get_expected_history_statistics (:476-533) on the device: from
 * the per-edge weights W_e (edge_weights above, one slice per edge: f64[nedges][n][n]),
 * the rate matrix Q[q_index[e]] and branch length t[e] of every edge,
 *   M_e = L(t_e Q_e^T, W_e)  (Frechet derivative of expm: the corner of the exponential
 *   of the 2n x 2n block [[t Q^T, W], [0, t Q^T]], all edges in one expm launch),
 *   dwell[c] = sum_e t_e M_e[c][c],   trans[c][d] = sum_e t_e Q_e[c][d] M_e[c][d]
 * -- what the reference gets from n + nnz(Q) scipy.linalg.expm_frechet calls per edge
 * and site.  2n <= RT_MAX_EXPM_STATES (n <= 64: the 61-state codon model included).   */
int rt_mjp_frechet_statistics(rt_ctx *ctx, int64_t n, int64_t nedges, const double *Q,
            int64_t nq, const int64_t *q_index, const double *t, const double *W,
            double *dwell, double *trans);

/* ---- 2. batched, device-resident hot path --------------------------------
 * _mjp_dense.get_likelihood (_mjp_dense.py:362-407) for many sites:
 *   rt_model_create        tree (same CSR as above) -> device, schedule built
 *   rt_model_set_rates     per-edge expm(Q*t) on the device, P stays resident
 *                          (get_expm_augmented_tree, _mjp_dense.py:328-359)
 *   rt_sites_create        per-site observations -> kernel-native HBM layout
 *   rt_prune               upward pass + root reduce + log + batch sum
 *                          (_mcy_dense.py:286 + _mc0_dense.py:147-212)
 * rt_model_set_rates, rt_prune and rt_allreduce_totals are asynchronous on the
 * context's stream; the rt_*_get_* functions synchronise.                    */

int rt_model_create(rt_ctx *ctx, int64_t nnodes, int64_t n,
            const int64_t *tree_csr_indices, const int64_t *tree_csr_indptr,
            rt_model **model);
int rt_model_destroy(rt_model *model);

/* Q f64[nq][n][n]; node_q int64[nnodes] = which Q the edge above node v uses
 * (entry 0, the root, is ignored; NULL = all edges use Q[0]); t f64[nnodes]
 * branch length of the edge above node v (t[0] ignored).                    */
int rt_model_set_rates(rt_model *model, const double *Q, int64_t nq,
            const int64_t *node_q, const double *t);
/* ONE time-reversible rate matrix given by its spectral decomposition, the reference's
 * optional fast path (examples/p53/qtop.py:128-152 decompose_spectral_v2, :76-88
 * getp_spectral_v2): Q = S diag(D), eigh(diag(sqrt D) S diag(sqrt D)) = U diag(lam) U^T,
 * A = diag(1/sqrt D) U (rows of states with D == 0 zero), B = U^T diag(sqrt D).  Every edge
 * gets P_e = A diag(exp(lam t_e)) B, and P_e[i][i] = 1 where D[i] == 0 (D may be NULL: no
 * such state).  The decomposition is the caller's (once per Q); this call and every later
 * rt_model_recompute_transitions / rt_step rebuild all edges from it in one launch, until
 * rt_model_set_rates is called again.  A, B f64[n][n]; lam, D f64[n]; t as above.  n <= 64. */
int rt_model_set_rates_spectral(rt_model *model, const double *A, const double *lam,
            const double *B, const double *D, const double *t);
/* Re-run the per-edge expm from the Q / node_q / t already resident on the
 * device (what an optimiser or MCMC loop does after changing rates in place;
 * also one benchmark step).                                                 */
int rt_model_recompute_transitions(rt_model *model);
/* Set / read back esd_transitions f64[nnodes][n][n] directly.               */
int rt_model_set_transitions(rt_model *model, const double *esd_transitions);
int rt_model_get_transitions(rt_model *model, double *esd_transitions);
/* Pade degree / squarings per node from the last rt_model_set_rates.        */
int rt_model_get_expm_info(rt_model *model, int32_t *info /*[nnodes][2]*/);
/* root_distn f64[n] or NULL (= weights of one, NOT uniform:
 * _mjp_dense.py:389-393).                                                   */
int rt_model_set_root_distn(rt_model *model, const double *root_distn);

/* Number of accumulator slots the post-order schedule needs (the fast
 * kernels hold 8 in registers; deeper trees use the generic kernel).        */
int rt_model_schedule_depth(const rt_model *model);
/* Copy the schedule out: int32[nops][4] = {node, obs, pop, dst} (see
 * csrc/common.h rt_op).  ops may be NULL to query nops only.                */
int rt_model_get_schedule(const rt_model *model, int32_t *ops, int64_t capacity,
            int64_t *nops);
/* The same schedule computed on the host only (no device needed; tests).    */
int rt_build_schedule(int64_t nnodes, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, int32_t *ops /*[nnodes][4]*/,
            int32_t *depth);
/* Process-wide options: "force_generic" (0/1) routes every later
 * rt_sites_create to the generic fallback kernel.  "jit" (-1 automatic, 0
 * never, 1 always): rt_sites_create compiles (hiprtc, once per distinct tree
 * and set of observed nodes) a pruning kernel specialised for the tree (n <=
 * 64); automatic = batches of at least 65 536 / n sites.  Results are
 * bit-identical with and without it.  "jit_block_sites" (0 automatic, 1..64):
 * sites per wave of those kernels (automatic balances the waves over the CUs). */
/* "jit_async" (1 default / 0): rt_sites_create does not wait for hiprtc -- a host
 * thread of this process compiles the kernel (or loads its code object from the persistent
 * cache directory: RAOTEH_JIT_CACHE_DIR, default ~/.cache/raoteh_amd/jit; RAOTEH_JIT_CACHE=0
 * disables it) while the batch runs the interpreter kernel; a later rt_prune / rt_step swaps
 * the kernel in once it is there and has passed the probe verification.  rt_sites_jit_wait
 * blocks until that has happened (or failed: the batch then stays on the interpreter).
 * At most RAOTEH_JIT_MAX_JOBS (default 2) such threads run at a time in a process: a batch
 * created while they are busy keeps the interpreter kernel (a search over topologies does
 * not pile up compiles; set "jit" to 0 there to skip the source generation as well).     */
/* "rescale" (0 default / 1): batches created while it is set rescale their messages on the
 * way up -- whenever the largest entry of a site's message falls below 2^-256 the message is
 * multiplied by the exact power of two that brings it back to [1, 2) and the exponent is kept
 * per site; log-likelihood = log(scaled likelihood) + exponent ln 2.  The reference never
 * rescales (_mc0_dense.py:184-209 works on plain f64): a tree of a thousand leaves is "zero
 * probability" there and, by default, here.  With the option such a batch keeps the
 * interpreter kernels (no tree-specialised kernel); a batch that never comes near the
 * threshold gets the default kernels' numbers bit for bit (powers of two are exact).
 * Likelihood evaluation only (rt_prune / rt_step): expectations stay unscaled.             */
/* "leaf_state_kernels" (1 default / 0): a batch uploaded as observed states at the leaves
 * (RT_OBS_STATE, every leaf observed, 5..128 states) may run a tree-specialised kernel whose
 * leaf steps gather a column of P instead of multiplying P by the one-hot vector -- the same
 * numbers bit for bit, about half the products; likewise RT_OBS_MASK batches whose leaves all
 * have one or two allowed states (two columns added); 0 keeps the products (what a dense upload
 * runs).                                                                                      */
int rt_set_option(const char *key, int64_t value);
/* The same options for ONE context (they take precedence over the process-wide
 * defaults above; value -2 = back to the default): two contexts on two threads can
 * then run with different settings without sharing mutable state.              */
int rt_ctx_set_option(rt_ctx *ctx, const char *key, int64_t value);
/* Diagnostics, host only: the HIP source rt_sites_create would compile for this
 * tree (n <= 32; MFMA family for n > 4), observation stream and prefetch distance. */
int rt_jit_source(int64_t nnodes, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, int64_t n, int64_t nobs,
            const int64_t *obs_nodes, int64_t prefetch, char *buf, int64_t capacity);

/* obs_nodes int64[nobs]: preorder indices of the nodes that carry per-site
 * data (all other nodes are unrestricted).  data layout per `kind` above.   */
int rt_sites_create(rt_model *model, int64_t nsites, int kind, int64_t nobs,
            const int64_t *obs_nodes, const void *data, rt_sites **sites);
/* A second batch with the same values at different HBM addresses (used by
 * the benchmark to rotate batches so the 256 MiB Infinity Cache cannot hold
 * the working set).                                                         */
int rt_sites_clone(rt_sites *sites, rt_sites **clone);
/* Wait for the background compile of this batch's tree-specialised kernel, if one is
 * pending, and switch the batch to it (see "jit_async").                              */
int rt_sites_jit_wait(rt_sites *sites);
/* Join every background compile of the process.  Call it before the process exits if batches
 * may still be compiling (rt_ctx_destroy does it for its own context): a compile thread must
 * not be running while exit handlers tear the compiler down.                            */
int rt_jit_wait_all(void);
int rt_sites_destroy(rt_sites *sites);
int64_t rt_sites_device_bytes(const rt_sites *sites);
/* Seconds rt_sites_create spent in hiprtc for this batch's tree-specialised kernel
 * (0: none, or the kernel came from the cache), and the name of the pruning kernel
 * variant the batch last ran (owned by the batch; "" before the first rt_prune). */
double rt_sites_jit_compile_seconds(const rt_sites *sites);
/* Diagnostics: copy a __device__ variable of the batch's compiled kernel to the host
 * (the per-step clock stamps of RAOTEH_JIT_TRACE, tools/trace_c3.py).              */
int rt_debug_jit_global(rt_sites *sites, const char *name, void *dst, int64_t bytes);
const char *rt_sites_kernel_name(const rt_sites *sites);

int rt_prune(rt_model *model, rt_sites *sites);
/* One evaluation of the repeated-evaluation loop (optimiser / MCMC iteration) in
 * one call: rt_model_recompute_transitions (if recompute_transitions != 0) +
 * rt_prune.                                                                   */
int rt_step(rt_model *model, rt_sites *sites, int recompute_transitions);
/* _mjp_dense.get_expected_history_statistics (_mjp_dense.py:410-539) summed over a RESIDENT
 * batch: (optionally) the per-edge expm from the resident rates, the upward pass, the downward
 * pass (mc0_esd_get_node_to_distn), the per-edge site sums of J / P and one Frechet block
 * exponential per edge, all on the device; 2 n + n^2 numbers come back:
 *   dwell[c]       expected time spent in state c, summed over edges and sites
 *   root_posterior[c]  sum over sites of the posterior probability of state c at the root
 *   trans[c][d]    expected number of c -> d transitions (0 where no rate matrix has c -> d)
 * each site weighted by rt_sites_set_weights (multiplicities of site patterns; default 1).
 * status (optional, int32[nsites]): 2 where a normalising denominator is zero (the reference
 * raises NumericalZeroProb).  n <= RT_MAX_EXPECT_STATES, batches created by rt_sites_create
 * (n > 4: any observation kind; n <= 4: the fused lane kernel works on allowed sets, so a
 * dense batch counts a state as allowed where its likelihood is not zero), rates set by
 * rt_model_set_rates.  Synchronous.                                                     */
int rt_expect_step(rt_model *model, rt_sites *sites, int recompute_transitions,
            double *dwell, double *root_posterior, double *trans, int32_t *status);
/* weights f64[nsites] (copied to the device) or NULL = every site counts once           */
int rt_sites_set_weights(rt_sites *sites, const double *weights);
/* loglik f64[nsites] (-inf where status has RT_SITE_ZERO_PROB),
 * status int32[nsites]; either may be NULL.                                 */
int rt_sites_get_logliks(rt_sites *sites, double *loglik, int32_t *status);
/* totals[0] = sum of log-likelihoods over sites with non-zero likelihood,
 * totals[1] = number of zero-probability sites, totals[2] = number of sites */
int rt_sites_get_totals(rt_sites *sites, double totals[3]);

/* ---- 3. multi-GPU: one process per GPU, RCCL over xGMI ------------------- */

/* rank 0 calls rt_comm_unique_id and ships the 128 bytes to the other ranks
 * by any host channel; then every rank calls rt_comm_init.                  */
/* rt_comm_available: RT_OK iff librccl loads (dlopen + symbols only, no GPU, no
 * network) -- the ranks agree on it over their host channel BEFORE any rank enters
 * rt_comm_init, where a missing peer would be a hang.                        */
int rt_comm_available(void);
int rt_comm_unique_id(unsigned char id[128]);
int rt_comm_init(rt_ctx *ctx, int nranks, int rank, const unsigned char id[128]);
int rt_comm_destroy(rt_ctx *ctx);
/* ncclAllReduce(sum, f64, 3) of the totals of `sites`, in place on the
 * device, asynchronously on the context's communication stream (the next
 * rt_prune of the batch and rt_sites_get_totals wait for it).               */
int rt_allreduce_totals(rt_ctx *ctx, rt_sites *sites);
/* The totals of `count` batches in one collective (3 * count doubles) when the
 * batches were created one after the other (their totals are then neighbours in
 * device memory), else one collective each.                                  */
int rt_allreduce_totals_group(rt_ctx *ctx, rt_sites **sites, int64_t count);

/* ---- 4. Rao-Teh sweep core: ragged batches of trees, one shared matrix --------
 * One sweep of the Rao-Teh sampler (_sampler.py:366-390) re-samples the states of a
 * chunk tree (_graph_transform.py:298-375) per chain, all with the SAME uniformized
 * transition matrix P = I + Q / omega (_sample_mjp_dense.py:72-114); the topologies
 * differ from chain to chain and from sweep to sweep.  A forest is the concatenation
 * of `ntrees` such trees:
 *   tree_node_offset int64[ntrees + 1]   tree k owns nodes [off[k], off[k+1]), numbered
 *       in ITS OWN DFS preorder (local index 0 = its root)
 *   tree_csr_indptr  int64[total + ntrees]  tree k's indptr (nnodes_k + 1 entries, as
 *       _density.digraph_to_bool_csr returns it) starts at off[k] + k
 *   tree_csr_indices int64[total - ntrees]  tree k's child indices (nnodes_k - 1 local
 *       preorder indices) start at off[k] - k
 *   P f64[n][n]: an entry that is exactly zero is a structural zero; n <= 64
 *   allowed_sets uint64[total]: bit s = state s allowed at the node (in / out).
 *
 * rt_forest_passes: pyfelscore.mcy_get_node_to_pset then pyfelscore.get_node_to_set
 * with a boolean CSR of P (_mcy.py:139-181; un-accelerated twins _mcy.py:396-470,
 * _mc0.py:89-138), in place on allowed_sets, then (subtree_probability != NULL) the
 * upward pass with that matrix on every edge (_mcy.py:611-682): f64[total][n], every
 * entry written.                                                                  */
int rt_forest_passes(rt_ctx *ctx, int64_t n, int64_t ntrees,
            const int64_t *tree_node_offset, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *P,
            uint64_t *allowed_sets, double *subtree_probability);
/* pyfelscore.mcy_get_node_to_pset (_mcy.py:158,259; un-accelerated twin _mcy.py:396-470)
 * and pyfelscore.get_node_to_set (_mcy.py:168; twin _mc0.py:89-138) with the reference's own
 * arguments: ONE tree (children CSR in preorder index space), the transition matrix as a
 * boolean CSR shared by every edge (trans_csr_indptr int64[n + 1], trans_csr_indices the
 * column indices of the nonzero entries of each row, _mcy.py:148-149), state_mask
 * int64[nnodes][n] 0/1 updated in place.  tmp_state_mask int64[n] is the scratch row
 * pyfelscore asks for (may be NULL).  n <= 64.                                        */
int rt_mcy_get_node_to_pset(rt_ctx *ctx, int64_t nnodes, int64_t n,
            const int64_t *tree_csr_indices, const int64_t *tree_csr_indptr,
            const int64_t *trans_csr_indices, const int64_t *trans_csr_indptr,
            int64_t *state_mask);
int rt_get_node_to_set(rt_ctx *ctx, int64_t nnodes, int64_t n,
            const int64_t *tree_csr_indices, const int64_t *tree_csr_indptr,
            const int64_t *trans_csr_indices, const int64_t *trans_csr_indptr,
            int64_t *state_mask, int64_t *tmp_state_mask);
/* The same passes followed by _sample_mc0_dense.resample_states (:53-98) for every
 * tree: root ~ root_distn * L[root] (root_distn NULL = weights of one), child ~
 * P[parent's state] * L[child].  Draws come from Philox-4x32-10 keyed by `seed` with
 * counter (sweep, global node index): the same (seed, sweep, forest) gives the same
 * states whatever the launch shape.  states int32[total] out (-1 where status != 0),
 * status int32[ntrees] out: 1 = the tree has zero likelihood (the reference raises
 * StructuralZeroProb / NumericalZeroProb, _sample_mc0_dense.py:57-62), 2 = a
 * non-root node had no state of positive weight.  subtree_probability (optional,
 * f64[total][n]) receives the upward messages the draws were made from.           */
int rt_forest_resample_states(rt_ctx *ctx, int64_t n, int64_t ntrees,
            const int64_t *tree_node_offset, const int64_t *tree_csr_indices,
            const int64_t *tree_csr_indptr, const double *P, const double *root_distn,
            uint64_t *allowed_sets, uint64_t seed, uint64_t sweep, int32_t *states,
            int32_t *status, double *subtree_probability);
/* The same with the topologies as ONE local parent index per node instead of the CSR
 * (tree_parent int32[total]: -1 at each tree's node 0, else an index below the node's
 * own) -- what a batched sweep has at hand after cutting its histories into chunks
 * (raoteh_amd/_sampler.py; the chunk trees of _graph_transform.py:298-375 come out in
 * that order), without a pass over the forest to build child lists.              */
int rt_forest_resample_states_parents(rt_ctx *ctx, int64_t n, int64_t ntrees,
            const int64_t *tree_node_offset, const int32_t *tree_parent, const double *P,
            const double *root_distn, uint64_t *allowed_sets, uint64_t seed, uint64_t sweep,
            int32_t *states, int32_t *status);

/* ---- 5. Rao-Teh sweeps with the histories resident on the device ---------------
 * A batch of `nchains` independent chains on ONE base tree (nnodes <= 1024 nodes in an
 * order with parent[v] < v, parent[0] = -1; branch_lengths[v] = length of the edge above
 * v) under ONE rate matrix, given as the caller computes it for the reference's sampler
 * (_sampler.py:344-357): P = I + Q / omega (f64[n][n]), poisson_rates[s] = omega - q_s,
 * root_distn (f64[n] or NULL = weights of one), node_masks uint64[nchains][nnodes]
 * allowed-state sets of the base nodes.  A history is a run of rows (edge = index of the
 * edge's lower node, length, state) sorted by (edge, position); rt_chains_create finds a
 * first feasible one by bisecting the edges (_sampler.py:563-648; RT_ERR_ZERO_PROB when
 * some chain has none), rt_chains_sweep runs whole sweeps (_sampler.py:366-390: Poisson
 * events, chunk trees, posterior draw of the chunk states, removal of self transitions)
 * without the rows leaving the device.  Draws are counter-based: (seed, batch) fixes
 * every history.  rt_chains_get_statistics: per chain the time spent in each state
 * (f64[nchains][n]), the transition counts (int64[nchains][n][n]) and the states of the
 * base nodes (int32[nchains][nnodes]) of the CURRENT histories; any pointer may be NULL.
 * rt_chains_get_rows: the histories themselves (chain_offset int64[nchains + 1], then
 * `capacity` >= total rows entries of edge / length / state; rt_chains_get_sizes tells
 * the total).                                                                        */
typedef struct rt_chains rt_chains;
int rt_chains_create(rt_ctx *ctx, int64_t nnodes, const int32_t *parent,
            const double *branch_lengths, int64_t n, const double *P,
            const double *poisson_rates, const double *root_distn, int64_t nchains,
            const uint64_t *node_masks, uint64_t seed, rt_chains **out);
int rt_chains_sweep(rt_chains *chains, int64_t nsweeps);
int rt_chains_get_sizes(const rt_chains *chains, int64_t *rows, int64_t *chunks,
            int64_t *sweeps);
int rt_chains_get_statistics(rt_chains *chains, double *dwell, int64_t *transitions,
            int32_t *node_states);
int rt_chains_get_rows(rt_chains *chains, int64_t capacity, int64_t *chain_offset,
            int32_t *edge, double *length, int32_t *state);
/* Metropolis-Hastings on top of the sweeps (_sampler.py:393-551): rt_chains_snapshot keeps
 * a copy of the current histories; after further sweeps rt_chains_restore gives the chains
 * with reject[c] != 0 (uint8[nchains]) their snapshot back, the others keep what they have. */
int rt_chains_snapshot(rt_chains *chains);
int rt_chains_restore(rt_chains *chains, const uint8_t *reject);
int rt_chains_destroy(rt_chains *chains);

#ifdef __cplusplus
}
#endif
#endif /* RAOTEH_HIP_H */
