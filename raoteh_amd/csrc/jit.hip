// Tree-specialised pruning kernels, generated and compiled at run time: three
// generators (n <= 4: lane per site; 4 < n <= 32: one wave = T site tiles on the
// f64 matrix pipe; 32 < n <= 64: split-M, NT waves share T tiles), one cache, one
// launch path.  The description below is the lane family's; the MFMA families'
// are at their generators.
//
// The interpreter kernels of prune.hip walk a host-built program: per step a
// scalar fetch of the entry, wave-uniform branches on its flags, LDS offsets
// for the pending accumulators.  On config 2 (4 states, 64 leaves) that
// bookkeeping is 38 % of the instructions and every step exposes an LDS or
// scalar-cache round trip that two waves per SIMD cannot hide.  For a batch that
// is evaluated many times (every optimiser / MCMC iteration re-uses tree and
// sites) the tree is a constant, so this file emits the walk as straight-line
// HIP source for THAT tree and compiles it with hiprtc for gfx950:
//   * every accumulator is a named register variable (no LDS stack),
//   * every P element is an LDS read at an immediate offset (one shared copy of
//     the step-ordered table per workgroup),
//   * the leaf vectors stream HBM -> VGPR with a compile-time prefetch distance
//     and the compiler's own counted vmcnt waits (no loop-carried loads),
//   * no scalar program, no branches.
// The arithmetic (order of every multiply / fma) is the interpreter's, so both
// produce bit-identical log-likelihoods (tests/test_gpu_parity.py checks it).
//
// Reference path replaced: the per-site loop around _mjp_dense.get_likelihood
// (_mjp_dense.py:362-407), as for prune.hip.
#include "common.h"

#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>

namespace {

struct jit_entry {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;     // null: compiled and rejected (register spills)
    int refs = 0;                   // site batches that may still launch it
    bool verified = false;          // bit-identical with the interpreter on the probe batch
    bool rejected = false;          // compiled, loaded, and found WRONG on the probe batch
};

// source text shared with the library's own kernels (expm_small.inc, reduce_body.inc): the
// n <= 4 exponential and the fixed-order batch sum, for lane kernels that compute the
// transition matrices of a step in their prologue (`fuse`)
#define RT_SHARED_SOURCE(...) #__VA_ARGS__
const char *const k_expm_small_src =
#include "expm_small.inc"
    ;
const char *const k_reduce_src =
#include "reduce_body.inc"
    ;
#undef RT_SHARED_SOURCE

std::mutex g_jit_mutex;
std::map<std::pair<const rt_ctx *, std::string>, jit_entry> g_jit_cache;   // (context, source)

// registers of P record rec: p<rec>_<k>, loaded LA steps before their use
void emit_p_load(std::ostringstream &o, int n, int rec)
{
    const int total = n * n;
    for (int k = 0; k < total;) {
        const int e = rec * total + k;
        if ((e & 1) == 0 && k + 1 < total) {
            o << "    const rt_d2 q" << rec << "_" << k << " = plv[" << e / 2 << "];\n";
            o << "    const double p" << rec << "_" << k << " = q" << rec << "_" << k << ".x, p"
              << rec << "_" << (k + 1) << " = q" << rec << "_" << k << ".y;\n";
            k += 2;
        } else {        // odd n: an element that does not start a 16-byte pair
            o << "    const double p" << rec << "_" << k << " = pls[" << e << "];\n";
            k += 1;
        }
    }
}

void emit_matvec(std::ostringstream &o, int n, int rec, const std::string &x,
                 const std::string &t)
{
    // t_r = P[rec][r][0] * x_0, then fma over the remaining columns: exactly
    // LaneCtx::matvec (prune.hip)
    for (int r = 0; r < n; ++r) {
        o << "    double " << t << r << " = p" << rec << "_" << (r * n) << " * " << x << "0;\n";
        for (int j = 1; j < n; ++j)
            o << "    " << t << r << " = fma(p" << rec << "_" << (r * n + j) << ", " << x << j
              << ", " << t << r << ");\n";
    }
}

}  // namespace

// A pair load (rt_d2) of which only .x is used -- the last k-pair when the number of
// k-steps is odd -- is emitted as an 8-byte load.  Loading the dead half too is not
// harmless: the register allocator hands the dead registers to another value while the
// load is still in flight, and the write-after-write hazard costs an s_waitcnt vmcnt(0)
// behind the prefetches just issued -- one full memory round trip per leaf step of the
// 20-state kernel (found with the step stamps of tools/trace_c5.py).
static std::string half_pair_load(const std::string &addr, bool nontemporal)
{
    const std::string p = "(const __attribute__((address_space(1))) double *)&" + addr;
    return "{" + (nontemporal ? "__builtin_nontemporal_load(" + p + ")" : "*(" + p + ")") + ", 0.0}";
}

// Source of the kernel for one schedule.  ops: post-order schedule with .obs
// filled in (rt_sites::ops); D: prefetch distance in stream positions.
// compact = 1 / 2: the batch is resident as one byte per observed node and site --
// its state (255 = unobserved) or its allowed-state set as a bit mask (n <= 4: what
// the IUPAC codes of a DNA alignment are) -- four stream positions per 32-bit word
// and lane ([block][word][lane]).  64 bytes per site instead of 2 KB: the kernel is
// no longer bound by HBM; the arithmetic is unchanged, so the results are those of
// the dense encoding bit for bit.
// fuse: the kernel can compute the transition matrices itself.  With `fq` non-null every
// workgroup runs the n <= 4 exponential of every edge into its LDS copy of the P table
// (thread = step; the text of the library's own kernel, expm_small.inc) instead of copying the
// table the expm launch left in global memory; workgroup 0 also leaves the matrices where
// that launch would have left them (esd order, step order, order / squarings), and one extra
// workgroup at the end of the grid may carry the pending batch-sum reduction, as an expm
// launch does.  One launch per step instead of two: on config 2 the boundary between two
// dependent launches costs more than the small kernel behind it.
std::string rt_jit_lane_source(const std::vector<rt_op> &ops, int n, int K, int D, int LA,
                               int S, int WG, int compact, bool fuse)
{
    const bool states = compact != 0;       // 1: uint8 states, 2: allowed-set masks (bytes)
    const bool masks = compact == 2;
    const int W = masks ? (1 << n) : n + 1;    // columns of the leaf table
    const int np = (n + 1) & ~1;
    const int hp = np / 2;
    const int nrec = (int)ops.size();
    int nslots = 1;
    for (const rt_op &op : ops) {
        if (op.pop >= 0) nslots = std::max(nslots, op.pop + 1);
        if (op.dst >= 0) nslots = std::max(nslots, (op.dst & 255) + 1);
    }
    std::ostringstream o;
    o << "// generated by raoteh_amd/csrc/jit.hip: " << nrec << " steps, " << n << " states, "
      << K << " observed nodes, prefetch distance " << D << " leaves / " << LA
      << " P records, " << S << " sites per wave, " << WG << " waves per workgroup\n";
    o << "typedef double rt_d2 __attribute__((ext_vector_type(2)));\n";
    if (fuse) {
        o << k_expm_small_src << "\n" << k_reduce_src << "\n";
        o << "__constant__ short rt_step_node[" << nrec << "] = {";
        for (int i = 0; i < nrec; ++i) o << (i ? ", " : "") << (int)ops[(size_t)i].node;
        o << "};\n";
    }
    // WG waves per workgroup share one copy of the step-ordered P table in LDS
    // (16 KB for 64 leaves x 4 states).  The host picks S (and WG = 1 by default)
    // so that a batch that fits the chip in one round puts the same number of
    // waves on every CU (C2: 1 792 waves of 56 sites, 7 per CU).  A pure streaming
    // kernel with this access pattern takes 37 us when its waves are spread evenly
    // and 40 us as 391 four-wave workgroups (tools/micro/membench.hip).  At most
    // 256 VGPRs: two waves per SIMD.
    const int total = nrec * n * n;
    const int NT = 64 * WG;
    // states: an observed leaf whose state is known needs one column of P, not the
    // product with a 0/1 vector: t_r = P[r][state] exactly (the other terms are +0),
    // and an unobserved one (255) the row sums in the order of the fma chain.  Such
    // steps fetch no P record: a second LDS table holds, for every leaf record, the
    // rows extended by their sum ([leaf][r][n + 1]); each lane gathers the n entries
    // of its own column (ds_read_b64 at a per-lane address: the n + 1 words of a row
    // sit on different banks, equal addresses broadcast), decoded and requested LA
    // steps ahead like a P record.
    std::vector<int> gather_of((size_t)nrec, -1);
    std::vector<int> gather_steps;
    if (states)
        for (int i = 0; i < nrec; ++i)
            if (ops[(size_t)i].pop < 0 && ops[(size_t)i].obs >= 0 && ops[(size_t)i].dst >= 0) {
                gather_of[(size_t)i] = (int)gather_steps.size();
                gather_steps.push_back(i);
            }
    const int G = (int)gather_steps.size();
    if (G > 0) {
        o << "__constant__ short rt_gmap[" << G << "] = {";
        for (int k = 0; k < G; ++k) o << (k ? ", " : "") << gather_steps[(size_t)k];
        o << "};\n";
    }
    // two waves per SIMD: a 256-VGPR budget
    o << "extern \"C\" __global__ void __launch_bounds__(" << NT << ")"
      << (WG == 1 ? " __attribute__((amdgpu_waves_per_eu(2, 2)))" : "") << "\n"
         "rt_jit_prune(const double *__restrict__ Pord, const rt_d2 *__restrict__ obs,\n"
         "             const double *__restrict__ root_w, double *__restrict__ loglik,\n"
         "             int *__restrict__ status, double *__restrict__ partial,\n"
         "             long nsites, long nblocks";
    if (fuse)
        o << ",\n             const double *__restrict__ fq, const int *__restrict__ fqidx,\n"
             "             const double *__restrict__ ftt, double *__restrict__ fP,\n"
             "             double *__restrict__ fPord, int *__restrict__ finfo,\n"
             "             const double *__restrict__ red_partial, long red_npartials,\n"
             "             double *__restrict__ red_totals, double red_nsites";
    o << ")\n{\n";
    if (fuse)
        o << "    if (red_partial && blockIdx.x == gridDim.x - 1) {      // the carried reduction\n"
             "        rt_reduce_partials_body(red_partial, red_npartials, red_totals, red_nsites);\n"
             "        return;\n"
             "    }\n";
    o << "    __shared__ __attribute__((aligned(16))) double pl[" << total + (total & 1) << "];\n";
    if (G > 0) o << "    __shared__ double pg[" << G * n * W << "];\n";
    o << "    const int lane = threadIdx.x & 63;\n";
    o << "    const long gw = (long)blockIdx.x * " << WG << " + (threadIdx.x >> 6);\n";
    // explicit address spaces: the pointers pass through (empty) inline asm below
    // and must stay global_load / ds_read, not flat
    o << "    typedef const __attribute__((address_space(1))) rt_d2 *rt_glb2;\n";
    // S sites per wave (block): lanes >= S re-read the last site's vector and their
    // results are dropped
    const int KQ = (K + 3) / 4;               // state words per site (states = true)
    const std::string lane_c =
        S < 64 ? "(lane < " + std::to_string(S) + " ? lane : " + std::to_string(S - 1) + ")"
               : std::string("lane");
    if (states) {
        o << "    typedef const __attribute__((address_space(1))) unsigned *rt_glbw;\n";
        o << "    rt_glbw g = (rt_glbw)obs + (size_t)(gw < nblocks ? gw : nblocks - 1) * "
          << (long)KQ * S << " + " << lane_c << ";\n";
    } else {
        o << "    rt_glb2 g = (rt_glb2)obs + (size_t)(gw < nblocks ? gw : nblocks - 1) * "
          << (long)K * hp * S << " + " << lane_c << ";\n";
    }
    o << "    typedef const __attribute__((address_space(3))) rt_d2 *rt_lds2;\n"
         "    typedef const __attribute__((address_space(3))) double *rt_lds1;\n"
         "    rt_lds2 plv = (rt_lds2)pl;\n"
         "    rt_lds1 pls = (rt_lds1)pl;\n";
    if (G > 0) o << "    rt_lds1 pgs = (rt_lds1)pg;\n";
    for (int j = 0; j < n; ++j) o << "    const double w" << j << " = root_w[" << j << "];\n";
    o << "    double lik = 0.0;\n    bool negative = false;\n";
    for (int s = 0; s < nslots; ++s)
        for (int j = 0; j < n; ++j) o << "    double a" << s << "_" << j << " = 1.0;\n";

    // timing experiment only (RAOTEH_JIT_NOLOAD): every leaf re-reads slot 0
    const bool noload = getenv("RAOTEH_JIT_NOLOAD") != nullptr;
    // the leaf vectors are read exactly once: non-temporal loads keep them from
    // displacing L2 / Infinity-Cache lines (C2: 36.5 -> 33 us); RAOTEH_JIT_NT=0 for A/B runs
    const bool nt = !(getenv("RAOTEH_JIT_NT") && atoi(getenv("RAOTEH_JIT_NT")) == 0);
    auto emit_load = [&](int k) {
        if (states) {
            // one word = four stream positions; position k is decoded where it is used
            if (k % 4 == 0)
                o << "    const unsigned sw" << k / 4 << " = __builtin_nontemporal_load(&g["
                  << (long)(k / 4) * S << "]);\n";
            return;
        }
        for (int h = 0; h < hp; ++h) {
            const long at = ((long)(noload ? 0 : k) * hp + h) * S;
            if ((n & 1) && h == hp - 1)
                o << "    const rt_d2 o" << k << "_" << h << " = "
                  << half_pair_load("g[" + std::to_string(at) + "]", nt) << ";\n";
            else
                o << "    const rt_d2 o" << k << "_" << h << " = "
                  << (nt ? "__builtin_nontemporal_load(&g[" : "g[") << at << (nt ? "]);\n" : "];\n");
        }
    };
    // Between two steps stands a scheduling barrier (no instruction): the
    // compiler's scheduler, left alone, sinks every LDS read to just before its
    // use (two reads in flight: the kernel is then bound by LDS latency), so the
    // P record of step i + LA and the leaf vectors D stream positions ahead are
    // pinned to the region of step i; the waits the compiler inserts are counted
    // (LDS and VMEM each return in order).
    // the HBM stream starts first; the P table (L2) is staged behind it
    for (int k = 0; k < std::min(D, K); ++k) emit_load(k);
    if (fuse) {
        const int NN = n * n;
        o << "    if (fq) {\n"
             "        for (int k = threadIdx.x; k < " << nrec << "; k += " << NT << ") {\n"
             "            const int qi = fqidx[k];      // in step order: no lookup before the loads\n"
             "            const double t = ftt[k];\n"
             "            SmallMat<" << n << "> A, X;\n"
             "            _Pragma(\"unroll\")\n"
             "            for (int i = 0; i < " << n << "; ++i)\n"
             "                _Pragma(\"unroll\")\n"
             "                for (int j = 0; j < " << n << "; ++j) A.a[i][j] = fq[i * " << n << " + j];\n"
             "            int em = 0, es = 0;\n"
             "            if (qi < 0) {             // the root's step: zeros (_density.py:171)\n"
             "                _Pragma(\"unroll\")\n"
             "                for (int i = 0; i < " << n << "; ++i)\n"
             "                    _Pragma(\"unroll\")\n"
             "                    for (int j = 0; j < " << n << "; ++j) X.a[i][j] = 0.0;\n"
             "            } else {\n"
             "                if (qi > 0) {\n"
             "                    const double *Qb = fq + (long)qi * " << NN << ";\n"
             "                    _Pragma(\"unroll\")\n"
             "                    for (int i = 0; i < " << n << "; ++i)\n"
             "                        _Pragma(\"unroll\")\n"
             "                        for (int j = 0; j < " << n << "; ++j) A.a[i][j] = Qb[i * " << n << " + j];\n"
             "                }\n"
             "                _Pragma(\"unroll\")\n"
             "                for (int i = 0; i < " << n << "; ++i)\n"
             "                    _Pragma(\"unroll\")\n"
             "                    for (int j = 0; j < " << n << "; ++j) A.a[i][j] *= t;\n"
             "                rt_expm_small_taylor<" << n << ">(A, X, em, es);\n"
             "            }\n"
             "            _Pragma(\"unroll\")\n"
             "            for (int i = 0; i < " << n << "; ++i)\n"
             "                _Pragma(\"unroll\")\n"
             "                for (int j = 0; j < " << n << "; ++j) {\n"
             "                    pl[k * " << NN << " + i * " << n << " + j] = X.a[i][j];\n"
             "                }\n"
             "            if (blockIdx.x == 0) {\n"
             "                const int node = rt_step_node[k];\n"
             "                _Pragma(\"unroll\")\n"
             "                for (int e = 0; e < " << NN << "; ++e) {\n"
             "                    fPord[(long)k * " << NN << " + e] = X.a[e / " << n << "][e % " << n << "];\n"
             "                    fP[(long)node * " << NN << " + e] = X.a[e / " << n << "][e % " << n << "];\n"
             "                }\n"
             "                finfo[2 * node] = em;\n"
             "                finfo[2 * node + 1] = es;\n"
             "            }\n"
             "        }\n"
             "    } else\n";
    }
    o << "    {\n"
         "        const rt_d2 *src = (const rt_d2 *)Pord;\n"
         "        rt_d2 *dst = (rt_d2 *)pl;\n"
         "        rt_d2 z = {0.0, 0.0};\n";
    // all loads, then all stores: one L2 round trip, not one per KiB
    const int pairs = total / 2;
    const int rows = (pairs + NT - 1) / NT;
    for (int r = 0; r < rows; ++r) {
        const int left = pairs - r * NT;
        o << "        const rt_d2 s" << r << " = ";
        if (left >= NT) o << "src[threadIdx.x + " << r * NT << "];\n";
        else o << "threadIdx.x < " << left << " ? src[threadIdx.x + " << r * NT << "] : z;\n";
    }
    for (int r = 0; r < rows; ++r) {
        const int left = pairs - r * NT;
        if (left >= NT) o << "        dst[threadIdx.x + " << r * NT << "] = s" << r << ";\n";
        else o << "        if (threadIdx.x < " << left << ") dst[threadIdx.x + " << r * NT << "] = s" << r << ";\n";
    }
    if (total & 1) o << "        if (threadIdx.x == 0) pl[" << total - 1 << "] = Pord[" << total - 1 << "];\n";
    o << "    }\n"
         "    __syncthreads();\n";
    if (G > 0) {
        // column c of a leaf row = the fma chain of the general path with the 0/1 vector
        // x(c): states: x_j = (c == j || c == n, the unobserved leaf); masks: bit j of c
        o << "    for (int row = threadIdx.x; row < " << G * n << "; row += " << NT << ") {\n"
             "        const int base = rt_gmap[row / " << n << "] * " << n * n << " + (row % " << n
          << ") * " << n << ";\n";
        for (int j = 0; j < n; ++j) o << "        const double c" << j << " = pl[base + " << j << "];\n";
        o << "        for (int c = 0; c < " << W << "; ++c) {\n";
        for (int j = 0; j < n; ++j) {
            if (masks) o << "            const double x" << j << " = (c >> " << j << ") & 1 ? 1.0 : 0.0;\n";
            else o << "            const double x" << j << " = (c == " << j << " || c == " << n << ") ? 1.0 : 0.0;\n";
        }
        o << "            double t = c0 * x0;\n";
        for (int j = 1; j < n; ++j) o << "            t = fma(c" << j << ", x" << j << ", t);\n";
        o << "            pg[row * " << W << " + c] = t;\n"
             "        }\n"
             "    }\n"
             "    __syncthreads();\n";
    }
    o << "    if (gw >= nblocks) return;            // wave-uniform, after the last barrier\n"
         "    __builtin_amdgcn_sched_barrier(0);\n";
    auto emit_gather = [&](int i) {
        const rt_op &q = ops[(size_t)i];
        o << "    const unsigned st" << i << " = (sw" << q.obs / 4 << " >> " << 8 * (q.obs % 4)
          << ") & 255u;\n";
        if (masks)
            o << "    const unsigned col" << i << " = st" << i << " & " << (W - 1) << "u;\n";
        else
            o << "    const unsigned col" << i << " = st" << i << " < " << n << "u ? st" << i << " : " << n
              << "u;\n";
        for (int r = 0; r < n; ++r)
            o << "    const double q" << i << "_" << r << " = pgs["
              << (gather_of[(size_t)i] * n + r) * W << " + col" << i << "];\n";
    };
    for (int i = 0; i < std::min(LA, nrec); ++i) {
        if (gather_of[(size_t)i] >= 0) emit_gather(i);
        else if (ops[(size_t)i].dst >= 0) emit_p_load(o, n, i);
    }

    std::string dep = "w0";
    for (int i = 0; i < nrec; ++i) {
        const rt_op &op = ops[(size_t)i];
        // no instruction: ties the two base pointers to a result of the previous
        // step, so the loads below cannot be hoisted above it ...
        o << "    asm volatile(\"\" : \"+v\"(plv), \"+v\"(pls), \"+v\"(g)" << (G > 0 ? ", \"+v\"(pgs)" : "")
          << " : \"v\"(" << dep << "));\n";
        // ... and nothing sinks below the step it was written in
        o << "    __builtin_amdgcn_sched_barrier(0);\n";
        if (i + LA < nrec) {
            if (gather_of[(size_t)(i + LA)] >= 0) emit_gather(i + LA);
            else if (ops[(size_t)(i + LA)].dst >= 0) emit_p_load(o, n, i + LA);
        }
        if (op.obs >= 0 && op.obs + D < K) emit_load(op.obs + D);
        if (op.dst >= 0) dep = "a" + std::to_string(op.dst & 255) + "_0";
        o << "    {   // step " << i << ": node " << op.node << "\n";
        if (gather_of[(size_t)i] >= 0) {
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int r = 0; r < n; ++r)
                o << "    a" << d << "_" << r << (first ? " = q" : " *= q") << i << "_" << r << ";\n";
            o << "    }\n";
            continue;
        }
        // x = (internal ? accumulator : 1) * (observed ? leaf vector : 1)
        if (states && op.obs >= 0)
            o << "    const unsigned st = (sw" << op.obs / 4 << " >> " << 8 * (op.obs % 4)
              << ") & 255u;\n";
        for (int j = 0; j < n; ++j) {
            std::ostringstream obs_j;
            if (op.obs >= 0) {
                if (masks) obs_j << "(((st >> " << j << ") & 1u) ? 1.0 : 0.0)";
                else if (states) obs_j << "((st == " << j << "u || st == 255u) ? 1.0 : 0.0)";
                else obs_j << "o" << op.obs << "_" << (j >> 1) << ((j & 1) ? ".y" : ".x");
            }
            o << "    const double x" << j << " = ";
            if (op.pop >= 0) {
                o << "a" << op.pop << "_" << j;
                if (op.obs >= 0) o << " * " << obs_j.str();
            } else if (op.obs >= 0) {
                o << obs_j.str();
            } else {
                o << "1.0";
            }
            o << ";\n";
        }
        if (op.dst < 0) {
            // root reduction (_mc0_dense.py:184-209), as LaneCtx::compute
            o << "    double sacc = 0.0;\n";
            for (int j = 0; j < n; ++j) {
                o << "    negative |= x" << j << " < 0.0;\n";
                o << "    sacc += w" << j << " * fmax(x" << j << ", 0.0);\n";
            }
            o << "    lik = sacc;\n";
        } else {
            emit_matvec(o, n, i, "x", "t");
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int r = 0; r < n; ++r) {
                if (first) o << "    a" << d << "_" << r << " = t" << r << ";\n";
                else o << "    a" << d << "_" << r << " *= t" << r << ";\n";
            }
        }
        o << "    }\n";
    }

    // per-site log, status, wave partial sums: finish_site / wave_sum of prune.hip
    o << "    const long site = gw * " << S << " + lane;\n"
         "    const bool ok = lik > 0.0;\n"
         "    double sum = 0.0, nzero = 0.0;\n"
         "    if (lane < " << S << " && site < nsites) {\n"
         "        loglik[site] = ok ? log(lik) : -__builtin_inf();\n"
         "        status[site] = (ok ? " << RT_SITE_OK << " : " << RT_SITE_ZERO_PROB
      << ") | (negative ? " << RT_SITE_NEGATIVE << " : 0);\n"
         "        sum = ok ? log(lik) : 0.0;\n"
         "        nzero = ok ? 0.0 : 1.0;\n"
         "    }\n"
         "    for (int off = 32; off > 0; off >>= 1) {\n"
         "        sum += __shfl_xor(sum, off, 64);\n"
         "        nzero += __shfl_xor(nzero, off, 64);\n"
         "    }\n"
         "    if (lane == 0) {\n"
         "        partial[gw * 2] = sum;\n"
         "        partial[gw * 2 + 1] = nzero;\n"
         "    }\n"
         "}\n";
    return o.str();
}

// ---------------------------------------------------------------------------
// 4 < n <= 32: tree-specialised f64 MFMA kernel, one wave = T site tiles
// ---------------------------------------------------------------------------
//
// Same arithmetic as prune_mfma_solo_kernel (prune.hip): per step and 16-site
// tile, t = P_e * x as NT * KS v_mfma_f64_16x16x4_f64 (D register r of row tile
// m is the B operand of k-step 4m + r of the next edge, so messages never leave
// the lane).  What the specialisation changes:
//   * one wave owns T tiles: the A fragments of P_e (6 KB per step at n = 20) are
//     fetched once per T tiles instead of once per tile -- the interpreter kernel
//     requests 1.2 GB of them per launch on config 5;
//   * pending accumulators are named registers holding only the KS valid
//     doubles per lane (the interpreter keeps them in LDS, padded to 4 * NT);
//   * no program fetch, no branches, no copies between register arrays (620 of
//     the interpreter's 6 500 VALU instructions per wave are MFMAs).
//
// quad = true: the products run on v_mfma_f64_4x4x4_4b_f64 instead.  Its four blocks
// compute D_g = A_g B_g with (measured, tools/micro/mfma4_layout.hip) A lane 16k + 4g + i
// = A_g[i][k], B lane 16k + 4g + j = B_g[k][j], D lane 16i + 4g + j = D_g[i][j]: with the
// same A in all four blocks that is D[4 x 16] = A[4 x 4] B[4 x 16] with B and D in exactly
// the lane order of the 16x16x4 form (state = lane >> 4, site = lane & 15) -- one register
// of a message per instruction, 16 cycles instead of 64.  A tile-step then costs
// ceil(n/4)^2 x 16 cycles instead of ceil(n/16) ceil(n/4) x 64: 25 x 16 against 10 x 64 at
// 20 states, where the second 16-row tile of the 16x16x4 form is 3/4 padding.  Same k
// order, same messages, same observation layout; the A operand comes from the model's
// quad-block table (rt_model::d_Pquad), each lane fetching element (lane >> 4, lane & 3)
// of a 16-double block.  (The instruction's A-broadcast controls cbsz / abid have no
// effect on the f64 form -- probed -- so the block is fetched replicated.)
// ---------------------------------------------------------------------------
// 4x4x4-block form, two tile groups in turn ("ping-pong", T even)
// ---------------------------------------------------------------------------
// One wave, one step: chain (T * KS^2 block MFMAs), then fold the result into the parent's
// accumulator, prepare the next operands (accumulator x observation), park the next blocks
// of P -- 450 to 900 cycles of the 2 500 a step took at T = 4 with the matrix pipe idle
// (step stamps, tools/trace_c5.py).  Two waves per SIMD (T = 2) did not hide it: both run
// the same program from the same start and stay in phase -- both in their chains (sharing
// the pipe), then both outside (pipe idle): 2 C + E per pair of steps, the same as one
// wave with twice the tiles (measured: 68 us either way; 60 us with the observations
// served from L2, so the HBM stream is not what holds it).  Here the wave's T tiles are
// two groups, and the program itself is out of phase: while group A's chain of step i
// runs, the statements of group B's fold (step i - 1) and operand preparation (step i)
// are issued one or two behind each block of MFMAs; under group B's chain, group A's
// fold of step i, the park of the next blocks of P and group A's operands of step i + 1.
// Any step order works (the other group's chain always stands between a group's chain
// and its next one).  Same k order and fold order per tile as rt_jit_mfma_source:
// bit-identical results.
// RESULT: correct (the probe verification and the tests pass with it) and the tail between
// chains is gone (80 cycles), but a chain of 2-MFMA blocks with a statement behind each
// runs at 20-24 cycles per MFMA instead of 17.5, and a step takes the same 2 200 cycles.
// Kept behind RAOTEH_JIT_PINGPONG=1 as the record of the experiment.
static std::string rt_jit_mfma_quad_pp_source(const std::vector<rt_op> &ops, int n, int K, int T,
                                              int D)
{
    const int KS = (n + 3) / 4;
    const int KP = (KS + 1) / 2;
    const int TG = T / 2;
    const int NB = KS * KS;
    const int nrec = (int)ops.size();
    int nslots = 1;
    for (const rt_op &op : ops) {
        if (op.pop >= 0) nslots = std::max(nslots, op.pop + 1);
        if (op.dst >= 0) nslots = std::max(nslots, (op.dst & 255) + 1);
    }
    const int QS = ((KS * KS * 16 + 127) / 128) * 128;    // rt_quad_stride(n)
    const int QL = QS / 128;
    const char *qa_env = getenv("RAOTEH_JIT_QAHEAD");
    const int QA = std::max(1, qa_env ? atoi(qa_env) : 4);
    const char *pd_env = getenv("RAOTEH_JIT_PARKDELAY");
    const int PD = pd_env ? std::max(0, atoi(pd_env)) : 1;
    std::ostringstream o;
    o << "// generated by raoteh_amd/csrc/jit.hip (MFMA family, 4x4x4 blocks, two tile groups in "
         "turn): " << nrec << " steps, " << n << " states, " << K << " observed nodes, " << T
      << " tiles per wave, prefetch " << D << " leaves\n";
    o << "typedef double rt_d2 __attribute__((ext_vector_type(2)));\n";
    const char *trace_env = getenv("RAOTEH_JIT_TRACE");
    const bool trace = trace_env != nullptr;
    const long trace_wg = trace ? atol(trace_env) : 0;
    if (trace) o << "__device__ unsigned long long rt_trace[" << (nrec + 1) * 3 << "];\n";
    auto stamp = [&](int i, int which) {
        if (!trace) return;
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[" << i * 3 + which
          << "] = __builtin_readcyclecounter();\n";
    };
    o << "extern \"C\" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu("
      << (T <= 2 ? "2, 2" : "1, 1") << ")))\n"
         "rt_jit_prune(const double *__restrict__ Pfrag, const rt_d2 *__restrict__ obs,\n"
         "             const double *__restrict__ root_w, double *__restrict__ loglik,\n"
         "             int *__restrict__ status, double *__restrict__ partial,\n"
         "             long nsites, long nblocks, long first_tile, long stride)\n{\n"
         "    if (blockIdx.x % stride) return;      // a sparse launch: every stride-th workgroup works\n";
    o << "    const int lane = threadIdx.x;\n";
    if (trace)      // the constant 100 MHz clock next to the shader clock: the core frequency
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[" << nrec * 3 + 1
          << "] = __builtin_amdgcn_s_memrealtime();\n";
    o << "    const long tbase = first_tile + (long)(blockIdx.x / stride) * " << T << ";\n";
    o << "    typedef const __attribute__((address_space(1))) rt_d2 *rt_glb2;\n";
    o << "    __shared__ __attribute__((aligned(16))) double qa0[" << QS << "];\n";
    o << "    __shared__ __attribute__((aligned(16))) double qa1[" << QS << "];\n";
    o << "    rt_glb2 ag = (rt_glb2)Pfrag + lane;          // [step][QS doubles]\n";
    o << "    const int alane = (lane >> 4) * 4 + (lane & 3);\n";
    for (int t = 0; t < T; ++t) {
        o << "    const long tile" << t << " = tbase + " << t << ";\n";
        o << "    rt_glb2 g" << t << " = (rt_glb2)obs + (size_t)(tile" << t << " < nblocks ? tile" << t
          << " : nblocks - 1) * " << (long)K * KP * 64 << " + lane;\n";
    }
    for (int j = 0; j < KS; ++j)
        o << "    const bool rowok" << j << " = " << 4 * j << " + (lane >> 4) < " << n << ";\n";
    for (int t = 0; t < T; ++t) {
        o << "    double lik" << t << " = 0.0;\n    bool negative" << t << " = false;\n";
        for (int sl = 0; sl < nslots; ++sl)
            for (int j = 0; j < KS; ++j)
                o << "    double a" << sl << "_" << t << "_" << j << " = 1.0;\n";
        for (int j = 0; j < KS; ++j)
            o << "    double x" << t << "_" << j << " = 0.0, c" << t << "_" << j << " = 0.0;\n";
    }
    auto emit_obs_load = [&](int k) {
        for (int t = 0; t < T; ++t)
            for (int q = 0; q < KP; ++q) {
                const std::string at = "g" + std::to_string(t) + "[" +
                                       std::to_string(((long)k * KP + q) * 64) + "]";
                if ((KS & 1) && q == KP - 1)
                    o << "    const rt_d2 o" << k << "_" << t << "_" << q << " = "
                      << half_pair_load(at, true) << ";\n";
                else
                    o << "    const rt_d2 o" << k << "_" << t << "_" << q
                      << " = __builtin_nontemporal_load(&" << at << ");\n";
            }
    };
    auto emit_a_load = [&](int i) {
        for (int j = 0; j < QL; ++j)
            o << "    const rt_d2 V" << i << "_" << j << " = ag[" << ((long)i * QS / 2 + j * 64)
              << "];\n";
    };
    typedef std::vector<std::string> stmts;
    auto park = [&](int i, stmts &out) {           // blocks of step i: registers -> qa<i & 1>
        for (int j = 0; j < QL; ++j)
            out.push_back("((rt_d2 *)qa" + std::to_string(i & 1) + ")[" + std::to_string(j * 64) +
                          " + lane] = V" + std::to_string(i) + "_" + std::to_string(j) + ";");
    };
    auto operands = [&](int g, int i, stmts &out) {   // x of group g for step i
        const rt_op &op = ops[(size_t)i];
        for (int t = g * TG; t < (g + 1) * TG; ++t)
            for (int j = 0; j < KS; ++j) {
                std::ostringstream e;
                e << "x" << t << "_" << j << " = ";
                std::ostringstream obs_j;
                if (op.obs >= 0)
                    obs_j << "o" << op.obs << "_" << t << "_" << (j >> 1) << ((j & 1) ? ".y" : ".x");
                if (op.pop >= 0) {
                    e << "a" << op.pop << "_" << t << "_" << j;
                    if (op.obs >= 0) e << " * " << obs_j.str();
                } else if (op.obs >= 0) {
                    e << obs_j.str();
                } else {
                    e << "1.0";
                }
                e << ";";
                out.push_back(e.str());
            }
    };
    auto fold = [&](int g, int i, stmts &out) {       // result of group g's chain of step i
        const rt_op &op = ops[(size_t)i];
        const int d = op.dst & 255;
        const bool first = (op.dst >> 8) != 0;
        for (int t = g * TG; t < (g + 1) * TG; ++t)
            for (int j = 0; j < KS; ++j)
                out.push_back("a" + std::to_string(d) + "_" + std::to_string(t) + "_" +
                              std::to_string(j) + (first ? " = c" : " *= c") + std::to_string(t) +
                              "_" + std::to_string(j) + ";");
    };
    // chain of group g for step i; the statements of `shadow` are spread behind its blocks
    auto chain = [&](int g, int i, const stmts &shadow) {
        auto q_read = [&](int b) {
            const int kk = b / KS, rq = b % KS;
            o << "    const double Q" << i << "_" << g << "_" << rq << "_" << kk << " = qa" << (i & 1)
              << "[" << (rq * KS + kk) * 16 << " + alane];\n";
        };
        for (int b = 0; b < std::min(QA, NB); ++b) q_read(b);
        o << "    __builtin_amdgcn_sched_barrier(0);\n";
        size_t done = 0;
        for (int b = 0; b < NB; ++b) {
            const int kk = b / KS, rq = b % KS;
            if (b + QA < NB) q_read(b + QA);
            for (int t = g * TG; t < (g + 1) * TG; ++t) {
                o << "    c" << t << "_" << rq << " = __builtin_amdgcn_mfma_f64_4x4x4f64(Q" << i << "_"
                  << g << "_" << rq << "_" << kk << ", x" << t << "_" << kk << ", ";
                if (kk == 0) o << "0.0";
                else o << "c" << t << "_" << rq;
                o << ", 0, 0, 0);\n";
            }
            const size_t upto = shadow.size() * (size_t)(b + 1) / (size_t)NB;
            for (; done < upto; ++done) o << "    " << shadow[done] << "\n";
            o << "    __builtin_amdgcn_sched_barrier(0);\n";
        }
    };
    for (int k = 0; k < std::min(D, K); ++k) emit_obs_load(k);
    for (int i = 0; i < std::min(1 + PD, nrec); ++i)
        if (ops[(size_t)i].dst >= 0) emit_a_load(i);
    {
        stmts pre;
        if (nrec > 0 && ops[0].dst >= 0) park(0, pre);
        operands(0, 0, pre);
        for (const std::string &st : pre) o << "    " << st << "\n";
    }
    stmts pending;                                 // group B's fold of the previous step
    for (int i = 0; i < nrec; ++i) {
        const rt_op &op = ops[(size_t)i];
        o << "    // step " << i << ": node " << op.node << "\n";
        if (!getenv("RAOTEH_JIT_NO_PINS")) {
            o << "    asm volatile(\"\" : \"+v\"(ag)";
            for (int t = 0; t < T; ++t) o << ", \"+v\"(g" << t << ")";
            o << " : \"v\"(x0_0));\n";
        }
        o << "    __builtin_amdgcn_sched_barrier(0);\n";
        stamp(i, 0);
        if (i + 1 + PD < nrec && ops[(size_t)(i + 1 + PD)].dst >= 0) emit_a_load(i + 1 + PD);
        if (op.obs >= 0 && op.obs + D < K) emit_obs_load(op.obs + D);
        if (op.dst < 0) {
            // root reduction (_mc0_dense.py:184-209): flush what is pending, then all tiles
            stmts rest = pending;
            pending.clear();
            operands(1, i, rest);
            for (const std::string &st : rest) o << "    " << st << "\n";
            for (int j = 0; j < KS; ++j)
                o << "    const double w" << j << " = rowok" << j << " ? root_w[" << 4 * j
                  << " + (lane >> 4)] : 0.0;\n";
            for (int t = 0; t < T; ++t) {
                o << "    {\n    double sacc = 0.0;\n";
                for (int j = 0; j < KS; ++j) {
                    o << "    negative" << t << " |= rowok" << j << " && (x" << t << "_" << j
                      << " < 0.0);\n";
                    o << "    sacc += w" << j << " * fmax(x" << t << "_" << j << ", 0.0);\n";
                }
                o << "    sacc += __shfl_xor(sacc, 16, 64);\n"
                     "    sacc += __shfl_xor(sacc, 32, 64);\n"
                     "    lik" << t << " = sacc;\n    }\n";
            }
            continue;
        }
        stmts under_a = pending;
        pending.clear();
        operands(1, i, under_a);
        chain(0, i, under_a);
        stamp(i, 1);
        stmts under_b;
        fold(0, i, under_b);
        if (i + 1 < nrec && ops[(size_t)(i + 1)].dst >= 0) park(i + 1, under_b);
        if (i + 1 < nrec) operands(0, i + 1, under_b);
        chain(1, i, under_b);
        stamp(i, 2);
        fold(1, i, pending);
    }
    for (const std::string &st : pending) o << "    " << st << "\n";
    stamp(nrec, 0);
    if (trace)
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[" << nrec * 3 + 2
          << "] = __builtin_amdgcn_s_memrealtime();\n";
    for (int t = 0; t < T; ++t) {
        o << "    {\n"
             "    const long site = tile" << t << " * 16 + (lane & 15);\n"
             "    const bool ok = lik" << t << " > 0.0;\n"
             "    double sum = 0.0, nzero = 0.0;\n"
             "    if (lane < 16 && tile" << t << " < nblocks && site < nsites) {\n"
             "        loglik[site] = ok ? log(lik" << t << ") : -__builtin_inf();\n"
             "        status[site] = (ok ? " << RT_SITE_OK << " : " << RT_SITE_ZERO_PROB
          << ") | (negative" << t << " ? " << RT_SITE_NEGATIVE << " : 0);\n"
             "        sum = ok ? log(lik" << t << ") : 0.0;\n"
             "        nzero = ok ? 0.0 : 1.0;\n"
             "    }\n"
             "    for (int off = 32; off > 0; off >>= 1) {\n"
             "        sum += __shfl_xor(sum, off, 64);\n"
             "        nzero += __shfl_xor(nzero, off, 64);\n"
             "    }\n"
             "    if (lane == 0 && tile" << t << " < nblocks) {\n"
             "        partial[tile" << t << " * 2] = sum;\n"
             "        partial[tile" << t << " * 2 + 1] = nzero;\n"
             "    }\n"
             "    }\n";
    }
    o << "}\n";
    return o.str();
}

// sparse (quad form only; 1: one observed state per leaf, 2: allowed sets of one or two states;
// the batch's `sparse_ok`): a leaf's message is a column of its transition matrix or the sum of
// two, and the 4 x 4 blocks of that matrix are parked in this wave's LDS buffer anyway -- a leaf
// step reads its KS entries per lane and tile from there (qa[(j KS + state / 4) 16 + (state & 3) 4
// + (lane >> 4)]) instead of running KS^2 block MFMAs, and its leaf vector never crosses HBM:
// protein and compound-model alignments (20 states observed at every leaf, C5's allowed pairs).
std::string rt_jit_mfma_source(const std::vector<rt_op> &ops, int n, int K, int T, int D, int LA,
                               bool quad, int sparse)
{
    if (sparse && !quad) return std::string();
    // RAOTEH_JIT_PINGPONG=1: two tile groups in turn (T even).  Not the default: measured
    // no faster (C5, T = 4: 69.8 us against 67.4; T = 2 on 2 048 tiles: 44.8 against 40.4)
    {
        const char *pp_env = getenv("RAOTEH_JIT_PINGPONG");
        if (quad && T >= 2 && T % 2 == 0 && pp_env && atoi(pp_env) != 0 &&
            !getenv("RAOTEH_JIT_FAKE_LEAFMAJOR") && !getenv("RAOTEH_JIT_FAKE_ONETILE"))
            return rt_jit_mfma_quad_pp_source(ops, n, K, T, D);
    }
    const int NT = (n + 15) / 16;
    const int KS = (n + 3) / 4;
    const int KP = (KS + 1) / 2;
    const int nrec = (int)ops.size();
    int nslots = 1;
    for (const rt_op &op : ops) {
        if (op.pop >= 0) nslots = std::max(nslots, op.pop + 1);
        if (op.dst >= 0) nslots = std::max(nslots, (op.dst & 255) + 1);
    }
    std::ostringstream o;
    o << "// generated by raoteh_amd/csrc/jit.hip (MFMA family" << (quad ? ", 4x4x4 blocks" : "")
      << (sparse == 2 ? ", leaf state pairs" : sparse ? ", leaf states" : "")
      << "): " << nrec << " steps, " << n
      << " states, " << K << " observed nodes, " << T << " tiles per wave, prefetch " << D
      << " leaves / " << LA << " P records\n";
    o << "typedef double rt_d2 __attribute__((ext_vector_type(2)));\n";
    o << "typedef double rt_d4 __attribute__((ext_vector_type(4)));\n";
    // RAOTEH_JIT_TRACE=<workgroup>: diagnostics (tools/trace_c5.py) -- that wave stamps the
    // shader clock at the start of every step, before its first MFMA and after its last
    // MFMA has been issued: rt_trace[step][3]
    const char *trace_env = getenv("RAOTEH_JIT_TRACE");
    const bool trace = trace_env != nullptr;
    const long trace_wg = trace ? atol(trace_env) : 0;
    if (trace) o << "__device__ unsigned long long rt_trace[" << (nrec + 1) * 3 << "];\n";
    auto stamp = [&](int i, int which) {
        if (!trace) return;
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[" << i * 3 + which
          << "] = __builtin_readcyclecounter();\n";
    };
    // up to two tiles per wave: two waves per SIMD (256 VGPRs); more tiles: one wave
    // per SIMD and the whole 512-register file (accumulators of T tiles)
    const char *weu = getenv("RAOTEH_JIT_WAVES_EU");      // diagnostics: "1, 1" / "2, 2"
    o << "extern \"C\" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu("
      << (weu ? weu : (quad && T == 1 && KS <= 5) ? "3, 3" : T <= 2 ? "2, 2" : "1, 1") << ")))\n"
         "rt_jit_prune(const double *__restrict__ Pfrag, const rt_d2 *__restrict__ obs,\n"
         "             const double *__restrict__ root_w, double *__restrict__ loglik,\n"
         "             int *__restrict__ status, double *__restrict__ partial,\n"
         "             long nsites, long nblocks, long first_tile, long stride"
      << (sparse ? ", const unsigned *__restrict__ leafw, const double *__restrict__ Pesd" : "")
      << ")\n{\n"
         "    if (blockIdx.x % stride) return;      // a sparse launch: every stride-th workgroup works\n";
    o << "    const int lane = threadIdx.x;\n";
    if (trace)      // the constant 100 MHz clock next to the shader clock: the core frequency
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[" << nrec * 3 + 1
          << "] = __builtin_amdgcn_s_memrealtime();\n";
    o << "    const long tbase = first_tile + (long)(blockIdx.x / stride) * " << T << ";\n";
    o << "    typedef const __attribute__((address_space(1))) rt_d2 *rt_glb2;\n";
    const int QS = ((KS * KS * 16 + 127) / 128) * 128;    // rt_quad_stride(n)
    const int QL = QS / 128;                              // 1 KiB wave loads per step
    if (quad) {
        // the 4 x 4 blocks of P_e pass through LDS: fetched once per step as QL coalesced
        // 1 KiB loads (one step ahead), parked in the wave's own double buffer, and read
        // back block by block as each lane's element (lane >> 4, lane & 3) -- 16 distinct
        // addresses per read.  (Fetching every block replicated from global memory, 25
        // loads of 512 B per step at 20 states, made the kernel wait on the texture
        // addresser: 87 us at T = 4, 196 us at T = 1.)
        o << "    __shared__ __attribute__((aligned(16))) double qa0[" << QS << "];\n";
        o << "    __shared__ __attribute__((aligned(16))) double qa1[" << QS << "];\n";
        o << "    rt_glb2 ag = (rt_glb2)Pfrag + lane;          // [step][QS doubles]\n";
        o << "    const int alane = (lane >> 4) * 4 + (lane & 3);\n";
    } else {
        o << "    rt_glb2 ag = (rt_glb2)Pfrag + lane;      // [step][m][k-pair][lane][2]\n";
    }
    for (int t = 0; t < T; ++t) {
        o << "    const long tile" << t << " = tbase + " << t << ";\n";
        // tiles past the end re-read the last one; their results are dropped
        // RAOTEH_JIT_FAKE_LEAFMAJOR: timing experiment only (wrong results): address the
        // observations as if they were stored [leaf][tile] instead of [tile][leaf]
        // RAOTEH_JIT_FAKE_ONETILE: timing experiment only (wrong results): every wave reads
        // the observations of tiles 0..T-1 (L2-resident) -- the kernel without its HBM stream
        o << "    rt_glb2 g" << t << " = (rt_glb2)obs + (size_t)("
          << (getenv("RAOTEH_JIT_FAKE_ONETILE") ? std::to_string(t) + " + 0 * tile" + std::to_string(t)
              : "tile" + std::to_string(t) + " < nblocks ? tile" + std::to_string(t) + " : nblocks - 1")
          << ") * " << (getenv("RAOTEH_JIT_FAKE_LEAFMAJOR") ? (long)KP * 64 : (long)K * KP * 64)
          << " + lane;\n";
    }
    // the root weights are loaded where the root step uses them, not here: KS doubles
    // that stay live through the whole walk are the first values the register allocator
    // splits and spills once the file is full, and ROCm 7.2's allocator can lose half of
    // such a 64-bit value (see verify_jit_kernel in api.hip for the incident)
    // (RAOTEH_JIT_W_AT_START=1 restores the old placement: the reproducer of
    // tests/soak/spill_probe.py)
    const bool w_at_start = getenv("RAOTEH_JIT_W_AT_START") != nullptr;
    for (int j = 0; j < KS; ++j) {
        o << "    const bool rowok" << j << " = " << 4 * j << " + (lane >> 4) < " << n << ";\n";
        if (w_at_start)
            o << "    const double w" << j << " = rowok" << j << " ? root_w[" << 4 * j
              << " + (lane >> 4)] : 0.0;\n";
    }
    for (int t = 0; t < T; ++t) {
        o << "    double lik" << t << " = 0.0;\n    bool negative" << t << " = false;\n";
        for (int s = 0; s < nslots; ++s)
            for (int j = 0; j < KS; ++j)
                o << "    double a" << s << "_" << t << "_" << j << " = 1.0;\n";
    }
    auto emit_obs_load = [&](int k) {
        for (int t = 0; t < T; ++t)
            for (int q = 0; q < KP; ++q)
                if (getenv("RAOTEH_JIT_FAKE_LEAFMAJOR"))
                    o << "    const rt_d2 o" << k << "_" << t << "_" << q
                      << " = __builtin_nontemporal_load(&g" << t << "[" << (long)k * KP * 64
                      << " * nblocks + " << q * 64 << "]);\n";
                else if ((KS & 1) && q == KP - 1)
                    o << "    const rt_d2 o" << k << "_" << t << "_" << q << " = "
                      << half_pair_load("g" + std::to_string(t) + "[" +
                                        std::to_string(((long)k * KP + q) * 64) + "]", true)
                      << ";\n";
                else
                o << "    const rt_d2 o" << k << "_" << t << "_" << q
                  << " = __builtin_nontemporal_load(&g" << t << "[" << ((long)k * KP + q) * 64
                  << "]);\n";     // read once: keep L2 for the A fragments
    };
    auto emit_a_load = [&](int i) {
        if (quad) {
            for (int j = 0; j < QL; ++j)
                o << "    const rt_d2 V" << i << "_" << j << " = ag[" << ((long)i * QS / 2 + j * 64)
                  << "];\n";
            return;
        }
        for (int m = 0; m < NT; ++m)
            for (int q = 0; q < KP; ++q) {
                const long at = (((long)i * NT + m) * KP + q) * 64;
                if ((KS & 1) && q == KP - 1)
                    o << "    const rt_d2 A" << i << "_" << m << "_" << q << " = "
                      << half_pair_load("ag[" + std::to_string(at) + "]", false) << ";\n";
                else
                    o << "    const rt_d2 A" << i << "_" << m << "_" << q << " = ag[" << at << "];\n";
            }
    };
    auto is_sparse_leaf = [&](const rt_op &op) {
        return sparse && op.pop < 0 && op.obs >= 0 && op.dst >= 0;
    };
    // (sparse) the state words: four leaves a word (pairs: two), requested WA leaves ahead
    const int per_word = sparse == 2 ? 2 : 4;
    const int KW = (K + per_word - 1) / per_word;
    const int WA = 6;
    std::vector<char> word_seen((size_t)std::max(KW, 1), 0);
    auto emit_words_upto = [&](int k) {          // the words of stream positions <= k
        if (!sparse) return;
        for (int w = 0; w <= std::min(k, K - 1) / per_word; ++w) {
            if (word_seen[(size_t)w]) continue;
            word_seen[(size_t)w] = 1;
            for (int t = 0; t < T; ++t)
                o << "    const unsigned lw" << w << "_" << t << " = leafw[((size_t)(tile" << t
                  << " < nblocks ? tile" << t << " : nblocks - 1) * " << KW << " + " << w
                  << ") * 16 + (lane & 15)];\n";
        }
    };
    // quad form: the blocks of step i + LA are parked at the end of step i and fetched PD
    // steps before that (RAOTEH_JIT_PARKDELAY, default 1: with the fetch at the top of the
    // same step the park waited on it at every step -- an L2 round trip is longer than
    // one chain of 2 x 25 block MFMAs)
    const char *pd_env = getenv("RAOTEH_JIT_PARKDELAY");
    const int PD = quad ? (pd_env ? std::max(0, atoi(pd_env)) : 1) : 0;
    if (!sparse)
        for (int k = 0; k < std::min(D, K); ++k) emit_obs_load(k);
    emit_words_upto(WA);
    for (int i = 0; i < std::min(LA + PD, nrec); ++i)
        if (ops[(size_t)i].dst >= 0) emit_a_load(i);

    auto emit_q_park = [&](int i) {        // blocks of step i: registers -> qa<i & 1>
        for (int j = 0; j < QL; ++j)
            o << "    ((rt_d2 *)qa" << (i & 1) << ")[" << j * 64 << " + lane] = V" << i << "_" << j
              << ";\n";
    };
    if (quad)
        for (int i = 0; i < std::min(LA, nrec); ++i)
            if (ops[(size_t)i].dst >= 0) emit_q_park(i);
    std::string dep = "lane";
    for (int i = 0; i < nrec; ++i) {
        const rt_op &op = ops[(size_t)i];
        // pin this step's loads to this step (see rt_jit_lane_source);
        // RAOTEH_JIT_NO_PINS / RAOTEH_JIT_NO_SCHED_BARRIER: diagnostics
        // (tests/soak/spill_probe.py)
        if (!getenv("RAOTEH_JIT_NO_PINS")) {
            o << "    asm volatile(\"\" : \"+v\"(ag)";
            for (int t = 0; t < T; ++t) o << ", \"+v\"(g" << t << ")";
            o << " : \"v\"(" << dep << "));\n";
        }
        if (!getenv("RAOTEH_JIT_NO_SCHED_BARRIER"))
            o << "    __builtin_amdgcn_sched_barrier(0);\n";
        stamp(i, 0);
        if (i + LA + PD < nrec && ops[(size_t)(i + LA + PD)].dst >= 0) emit_a_load(i + LA + PD);
        if (!sparse && op.obs >= 0 && op.obs + D < K) emit_obs_load(op.obs + D);
        if (sparse && op.obs >= 0) emit_words_upto(op.obs + WA);
        if (op.dst >= 0) dep = "a" + std::to_string(op.dst & 255) + "_0_0";
        if (is_sparse_leaf(op)) {
            // ---- a leaf with observed state(s): columns of P from the parked blocks
            o << "    {   // step " << i << ": leaf " << op.node << " (column" << (sparse == 2 ? "s" : "")
              << " of P)\n";
            const int w = op.obs / per_word;
            const int sh = sparse == 2 ? 16 * (op.obs & 1) : 8 * (op.obs & 3);
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int t = 0; t < T; ++t) {
                o << "    const int st" << t << " = (int)((lw" << w << "_" << t << " >> " << sh << ") & 255u);\n";
                o << "    const double *qp" << t << " = qa" << (i & 1) << " + (st" << t << " >> 2) * 16 + (st"
                  << t << " & 3) * 4 + (lane >> 4);\n";
                if (sparse == 2) {
                    o << "    const int sq" << t << " = (int)((lw" << w << "_" << t << " >> " << sh + 8
                      << ") & 255u);\n";
                    o << "    const int sr" << t << " = sq" << t << " == 255 ? st" << t << " : sq" << t << ";\n";
                    o << "    const double *qq" << t << " = qa" << (i & 1) << " + (sr" << t << " >> 2) * 16 + (sr"
                      << t << " & 3) * 4 + (lane >> 4);\n";
                }
                for (int j = 0; j < KS; ++j) {
                    o << "    const double pc" << t << "_" << j << " = qp" << t << "[" << j * KS * 16 << "]";
                    if (sparse == 2)
                        o << " + (sq" << t << " == 255 ? 0.0 : qq" << t << "[" << j * KS * 16 << "])";
                    o << ";\n";
                }
            }
            // the next step's blocks (fetched at the top of this step) go to the other buffer
            if (i + LA < nrec && ops[(size_t)(i + LA)].dst >= 0) emit_q_park(i + LA);
            for (int t = 0; t < T; ++t)
                for (int j = 0; j < KS; ++j)
                    o << "    a" << d << "_" << t << "_" << j << (first ? " = " : " *= ") << "pc" << t << "_"
                      << j << ";\n";
            o << "    }\n";
            continue;
        }
        o << "    {   // step " << i << ": node " << op.node << "\n";
        for (int t = 0; t < T; ++t) {
            for (int j = 0; j < KS; ++j) {
                std::ostringstream obs_j;
                if (op.obs >= 0)
                    obs_j << "o" << op.obs << "_" << t << "_" << (j >> 1) << ((j & 1) ? ".y" : ".x");
                o << "    const double x" << t << "_" << j << " = ";
                if (op.pop >= 0) {
                    o << "a" << op.pop << "_" << t << "_" << j;
                    if (op.obs >= 0) o << " * " << obs_j.str();
                } else if (op.obs >= 0) {
                    o << obs_j.str();
                } else {
                    o << "1.0";
                }
                o << ";\n";
            }
        }
        if (op.dst < 0) {
            // root reduction (_mc0_dense.py:184-209), as prune_mfma_solo_kernel
            for (int j = 0; j < KS && !w_at_start; ++j)
                o << "    const double w" << j << " = rowok" << j << " ? root_w[" << 4 * j
                  << " + (lane >> 4)] : 0.0;\n";
            for (int t = 0; t < T; ++t) {
                o << "    {\n    double sacc = 0.0;\n";
                for (int j = 0; j < KS; ++j) {
                    o << "    negative" << t << " |= rowok" << j << " && (x" << t << "_" << j
                      << " < 0.0);\n";
                    o << "    sacc += w" << j << " * fmax(x" << t << "_" << j << ", 0.0);\n";
                }
                o << "    sacc += __shfl_xor(sacc, 16, 64);\n"
                     "    sacc += __shfl_xor(sacc, 32, 64);\n"
                     "    lik" << t << " = sacc;\n    }\n";
            }
        } else if (quad) {
            // one accumulator (register of the message) per row quad and tile; the KS * T
            // chains advance side by side, k-step outermost
            for (int t = 0; t < T; ++t)
                for (int rq = 0; rq < KS; ++rq) o << "    double c" << t << "_" << rq << " = 0.0;\n";
            // The block reads are issued QA blocks ahead of their MFMAs and pinned there
            // (sched_barrier(0) after every block): left to itself the scheduler, with the
            // register file full, put each read right in front of its use and the wave
            // waited out an LDS round trip 13 times per step with the matrix pipe drained
            // (2 650 cycles per step against 1 650 of MFMA work at T = 4).
            // RAOTEH_JIT_QAHEAD=0: the old placement.
            const char *qa_env = getenv("RAOTEH_JIT_QAHEAD");
            const int QA = qa_env ? atoi(qa_env) : 4;
            const int NB = KS * KS;
            auto emit_q_read = [&](int b) {
                const int kk = b / KS, rq = b % KS;
                o << "    const double Q" << i << "_" << rq << "_" << kk << " = qa" << (i & 1)
                  << "[" << (rq * KS + kk) * 16 << " + alane];\n";
            };
            for (int b = 0; b < std::min(QA, NB); ++b) emit_q_read(b);
            if (QA > 0) o << "    __builtin_amdgcn_sched_barrier(0);\n";
            if (trace) {
                // the stamp after the operands: x of every tile is in registers here
                o << "    asm volatile(\"\" :: \"v\"(x" << T - 1 << "_" << KS - 1 << "));\n";
                stamp(i, 1);
            }
            for (int b = 0; b < NB; ++b) {
                const int kk = b / KS, rq = b % KS;
                if (QA <= 0) emit_q_read(b);
                else if (b + QA < NB) emit_q_read(b + QA);
                for (int t = 0; t < T; ++t)
                    o << "    c" << t << "_" << rq << " = __builtin_amdgcn_mfma_f64_4x4x4f64(Q"
                      << i << "_" << rq << "_" << kk << ", x" << t << "_" << kk << ", c" << t
                      << "_" << rq << ", 0, 0, 0);\n";
                if (QA > 0) o << "    __builtin_amdgcn_sched_barrier(0);\n";
            }
            stamp(i, 2);
            // the next step's blocks (fetched at the top of this step) go to the other buffer
            if (i + LA < nrec && ops[(size_t)(i + LA)].dst >= 0) emit_q_park(i + LA);
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int t = 0; t < T; ++t)
                for (int j = 0; j < KS; ++j)
                    o << "    a" << d << "_" << t << "_" << j << (first ? " = " : " *= ") << "c" << t
                      << "_" << j << ";\n";
        } else {
            for (int t = 0; t < T; ++t)
                for (int m = 0; m < NT; ++m)
                    o << "    rt_d4 c" << t << "_" << m << " = {0.0, 0.0, 0.0, 0.0};\n";
            // independent chains side by side: k-step outermost
            for (int kk = 0; kk < KS; ++kk)
                for (int m = 0; m < NT; ++m)
                    for (int t = 0; t < T; ++t)
                        o << "    c" << t << "_" << m << " = __builtin_amdgcn_mfma_f64_16x16x4f64(A"
                          << i << "_" << m << "_" << (kk >> 1) << ((kk & 1) ? ".y" : ".x") << ", x"
                          << t << "_" << kk << ", c" << t << "_" << m << ", 0, 0, 0);\n";
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int t = 0; t < T; ++t)
                for (int j = 0; j < KS; ++j) {
                    o << "    a" << d << "_" << t << "_" << j << (first ? " = " : " *= ") << "c" << t
                      << "_" << (j >> 2) << "[" << (j & 3) << "];\n";
                }
        }
        o << "    }\n";
    }

    stamp(nrec, 0);
    if (trace)
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[" << nrec * 3 + 2
          << "] = __builtin_amdgcn_s_memrealtime();\n";
    // lanes 0..15 own the 16 sites of a tile (finish_site / wave_sum of prune.hip)
    for (int t = 0; t < T; ++t) {
        o << "    {\n"
             "    const long site = tile" << t << " * 16 + (lane & 15);\n"
             "    const bool ok = lik" << t << " > 0.0;\n"
             "    double sum = 0.0, nzero = 0.0;\n"
             "    if (lane < 16 && tile" << t << " < nblocks && site < nsites) {\n"
             "        loglik[site] = ok ? log(lik" << t << ") : -__builtin_inf();\n"
             "        status[site] = (ok ? " << RT_SITE_OK << " : " << RT_SITE_ZERO_PROB
          << ") | (negative" << t << " ? " << RT_SITE_NEGATIVE << " : 0);\n"
             "        sum = ok ? log(lik" << t << ") : 0.0;\n"
             "        nzero = ok ? 0.0 : 1.0;\n"
             "    }\n"
             "    for (int off = 32; off > 0; off >>= 1) {\n"
             "        sum += __shfl_xor(sum, off, 64);\n"
             "        nzero += __shfl_xor(nzero, off, 64);\n"
             "    }\n"
             "    if (lane == 0 && tile" << t << " < nblocks) {\n"
             "        partial[tile" << t << " * 2] = sum;\n"
             "        partial[tile" << t << " * 2 + 1] = nzero;\n"
             "    }\n"
             "    }\n";
    }
    o << "}\n";
    return o.str();
}

// ---------------------------------------------------------------------------
// 32 < n <= 64: tree-specialised split-M MFMA kernel, one workgroup = T tiles
// ---------------------------------------------------------------------------
//
// The decomposition of prune_mfma_kernel (prune.hip): NT waves per workgroup,
// wave m owns rows 16m..16m+15 of every message and its own slice of P_e (KS
// A-fragment doubles per lane and step); the waves publish their four rows of x
// through LDS and read all of x back as B operands.  Specialised for one tree:
//   * T site tiles per workgroup: one A fetch, one barrier and one program
//     step per T chains of KS MFMAs (the interpreter: per chain; it requests
//     2.3 GB of A fragments per launch on config 3);
//   * the x exchange buffer is double-buffered: ONE barrier per step;
//   * pending accumulators (own rows: 4 doubles per lane and tile) in named
//     registers, no LDS stack, no program fetch, no branches.
// Same accumulation order as the interpreter: bit-identical results.
// sparse = observed STATES at the leaves (type x, _mcx.py:12-23: the reference's own fast case):
// the product of a leaf's edge with a one-hot vector is a COLUMN of P_e, so a leaf step is four
// 8-byte gathers from the transition matrices in the reference's order (P[node][row][state],
// L2-resident) instead of an x exchange, a barrier and KS MFMAs -- half of the steps of a binary
// tree.  The matrix pipe computes that product as fma(P, 1, 0) plus exact zeros, so the gathered
// column is its result bit for bit (the probe verification checks it against the interpreter on
// the dense expansion).  Leaf states arrive as bytes, four stream positions per word:
// leafw[tile][ceil(K/4)][16 sites].  Only for batches whose observed nodes are all leaves and
// whose states are all observed (api.hip).
std::string rt_jit_mfma_split_source(const std::vector<rt_op> &ops, int n, int K, int T, int D,
                                     int LA, int sparse)
{
    const int NT = (n + 15) / 16;
    const int KS = (n + 3) / 4;
    const int KP = (KS + 1) / 2;
    const int nrec = (int)ops.size();
    const int XT = NT * 4 * 64;               // doubles of one tile's x image
    int nslots = 1;
    for (const rt_op &op : ops) {
        if (op.pop >= 0) nslots = std::max(nslots, op.pop + 1);
        if (op.dst >= 0) nslots = std::max(nslots, (op.dst & 255) + 1);
    }
    std::ostringstream o;
    o << "// generated by raoteh_amd/csrc/jit.hip (split-M MFMA family"
      << (sparse == 2 ? ", leaf state pairs" : sparse ? ", leaf states" : "")
      << "): " << nrec << " steps, "
      << n << " states, " << K << " observed nodes, " << T << " tiles per workgroup of " << NT
      << " waves, prefetch " << D << " leaves / " << LA << " P records\n";
    o << "typedef double rt_d2 __attribute__((ext_vector_type(2)));\n";
    o << "typedef double rt_d4 __attribute__((ext_vector_type(4)));\n";
    // RAOTEH_JIT_TRACE=<workgroup>: diagnostics (tools/trace_c3.py) -- the waves of that
    // workgroup stamp the shader clock at the start of every step, after its barrier and
    // after its last MFMA has been issued: rt_trace[wave][step][3]
    const char *trace_env = getenv("RAOTEH_JIT_TRACE");
    const bool trace = trace_env != nullptr;
    const long trace_wg = trace ? atol(trace_env) : 0;
    if (trace)
        o << "__device__ unsigned long long rt_trace[" << NT * (nrec + 1) * 3 << "];\n";
    auto stamp = [&](int i, int which) {
        if (!trace) return;
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[(m * " << (nrec + 1)
          << " + " << i << ") * 3 + " << which << "] = __builtin_readcyclecounter();\n";
    };
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * NT
      << ") __attribute__((amdgpu_waves_per_eu("
      << (NT > 4 ? "2, 2" : T == 1 ? "3, 3" : T == 2 ? "2, 2" : "1, 1") << ")))\n"
         "rt_jit_prune(const double *__restrict__ Pfrag, const rt_d2 *__restrict__ obs,\n"
         "             const double *__restrict__ root_w, double *__restrict__ loglik,\n"
         "             int *__restrict__ status, double *__restrict__ partial,\n"
         "             long nsites, long nblocks"
      << (sparse ? ", const unsigned *__restrict__ leafw, const double *__restrict__ Pesd" : "")
      << ")\n{\n";
    o << "    __shared__ double xb[2][" << T << "][" << XT << "];   // [buffer][tile][k-step][lane]\n";
    o << "    __shared__ double red[" << T << "][" << NT << "][16];\n";
    o << "    const int lane = threadIdx.x & 63;\n";
    o << "    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // row tile of this wave\n";
    o << "    const long tbase = (long)blockIdx.x * " << T << ";\n";
    o << "    typedef const __attribute__((address_space(1))) rt_d2 *rt_glb2;\n";
    o << "    rt_glb2 ag = (rt_glb2)Pfrag + (m * " << KP * 64 << " + lane);   // [step][m][k-pair][lane][2]\n";
    // own rows 4m..4m+3 of a leaf vector = pairs 2m, 2m+1 of its B-operand image
    o << "    const bool pair1 = 2 * m + 1 < " << KP << ";\n";
    for (int t = 0; t < T; ++t) {
        o << "    const long tile" << t << " = tbase + " << t << ";\n";
        o << "    rt_glb2 g" << t << " = (rt_glb2)obs + (size_t)(tile" << t << " < nblocks ? tile" << t
          << " : nblocks - 1) * " << (long)K * KP * 64 << " + (2 * m * 64 + lane);\n";
    }
    for (int r = 0; r < 4; ++r)
        o << "    const bool rowok" << r << " = 16 * m + " << 4 * r << " + (lane >> 4) < " << n << ";\n";
    for (int t = 0; t < T; ++t) {
        o << "    double lik" << t << " = 0.0;\n    bool negative" << t << " = false;\n";
        for (int s = 0; s < nslots; ++s)
            for (int r = 0; r < 4; ++r)
                o << "    double a" << s << "_" << t << "_" << r << " = 1.0;\n";
    }
    o << "    const rt_d2 zero2 = {0.0, 0.0};\n";
    auto emit_obs_load = [&](int k) {
        for (int t = 0; t < T; ++t) {
            o << "    const rt_d2 o" << k << "_" << t << "_0 = __builtin_nontemporal_load(&g" << t
              << "[" << (long)k * KP * 64 << "]);\n";
            o << "    const rt_d2 o" << k << "_" << t << "_1 = pair1 ? __builtin_nontemporal_load(&g"
              << t << "[" << ((long)k * KP + 1) * 64 << "]) : zero2;\n";
        }
    };
    auto emit_a_load = [&](int i) {
        for (int q = 0; q < KP; ++q) {
            const long at = ((long)i * NT * KP + q) * 64;
            if ((KS & 1) && q == KP - 1)
                o << "    const rt_d2 A" << i << "_" << q << " = "
                  << half_pair_load("ag[" + std::to_string(at) + "]", false) << ";\n";
            else
                o << "    const rt_d2 A" << i << "_" << q << " = ag[" << at << "];\n";
        }
    };
    auto is_sparse_leaf = [&](const rt_op &op) {
        return sparse && op.pop < 0 && op.obs >= 0 && op.dst >= 0;
    };
    if (sparse) {
        // the state bytes of every leaf of the workgroup's tiles, four per word (two per word
        // for allowed sets of one or two states: 16 bits per leaf)
        const int KW = sparse == 2 ? (K + 1) / 2 : (K + 3) / 4;
        for (int t = 0; t < T; ++t)
            for (int w = 0; w < KW; ++w)
                o << "    const unsigned lw" << w << "_" << t << " = leafw[((size_t)(tile" << t
                  << " < nblocks ? tile" << t << " : nblocks - 1) * " << KW << " + " << w
                  << ") * 16 + (lane & 15)];\n";
        o << "    const rt_d4 *pcm = (const rt_d4 *)Pesd + (4 * m + (lane >> 4));\n";
    }
    // the four entries of this lane of column `state` of P_node, from the step's A-fragment
    // record (zero in the padded rows): the four row-lanes of a site share a 64-byte sector
    auto emit_gather = [&](int i) {
        const rt_op &op = ops[(size_t)i];
        const int w = sparse == 2 ? op.obs >> 1 : op.obs >> 2;
        const int sh = sparse == 2 ? 16 * (op.obs & 1) : 8 * (op.obs & 3);
        for (int t = 0; t < T; ++t) {
            o << "    const int st" << i << "_" << t << " = (int)((lw" << w << "_" << t
              << " >> " << sh << ") & 255u);\n";
            o << "    const rt_d4 pf" << i << "_" << t << " = pcm[(" << (long)i * n << " + st" << i << "_" << t
              << ") * " << 4 * NT << "];\n";
            if (sparse == 2) {
                // the second allowed state (255: none -- the first column again, not added)
                o << "    const int sq" << i << "_" << t << " = (int)((lw" << w << "_" << t << " >> " << sh + 8
                  << ") & 255u);\n";
                o << "    const int sr" << i << "_" << t << " = sq" << i << "_" << t << " == 255 ? st" << i << "_"
                  << t << " : sq" << i << "_" << t << ";\n";
                o << "    const rt_d4 pg" << i << "_" << t << " = pcm[(" << (long)i * n << " + sr" << i << "_" << t
                  << ") * " << 4 * NT << "];\n";
            }
            for (int r = 0; r < 4; ++r) {
                o << "    const double pc" << i << "_" << t << "_" << r << " = pf" << i << "_" << t << "["
                  << r << "]";
                if (sparse == 2)
                    o << " + (sq" << i << "_" << t << " == 255 ? 0.0 : pg" << i << "_" << t << "[" << r
                      << "])";
                o << ";\n";
            }
        }
    };
    if (!sparse)
        for (int k = 0; k < std::min(D, K); ++k) emit_obs_load(k);
    // (sparse: the steps that multiply, in order -- the A fragments run LA of THOSE ahead)
    std::vector<int> msteps;
    for (int i = 0; i < nrec; ++i)
        if (!is_sparse_leaf(ops[(size_t)i])) msteps.push_back(i);
    if (sparse) {
        for (int q = 0; q < std::min(LA, (int)msteps.size()); ++q)
            if (ops[(size_t)msteps[(size_t)q]].dst >= 0) emit_a_load(msteps[(size_t)q]);
    } else {
        for (int i = 0; i < std::min(LA, nrec); ++i)
            if (ops[(size_t)i].dst >= 0) emit_a_load(i);
    }
    // gathers are requested one matrix step ahead of the leaf step that folds them: those of
    // the leaves in front of the first matrix step here, the others at the matrix step before
    if (sparse)
        for (int i = 0; i < nrec && is_sparse_leaf(ops[(size_t)i]); ++i) emit_gather(i);

    std::string dep = "lane";
    int nmfma = 0;                           // matrix steps so far (the x buffers alternate over THESE)
    for (int i = 0; i < nrec; ++i) {
        const rt_op &op = ops[(size_t)i];
        if (is_sparse_leaf(op)) {
            // ---- a leaf with an observed state: its message is a column of P
            o << "    {   // step " << i << ": leaf " << op.node << " (column of P)\n";
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < 4; ++r)
                    o << "    a" << d << "_" << t << "_" << r << (first ? " = " : " *= ") << "pc" << i
                      << "_" << t << "_" << r << ";\n";
            o << "    }\n";
            continue;
        }
        o << "    asm volatile(\"\" : \"+v\"(ag)";
        for (int t = 0; t < T; ++t) o << ", \"+v\"(g" << t << ")";
        o << " : \"v\"(" << dep << "));\n";
        o << "    __builtin_amdgcn_sched_barrier(0);\n";
        if (sparse) {
            // the A fragments of the next matrix steps, and the columns of the leaves that
            // follow this step
            const size_t q = (size_t)(std::find(msteps.begin(), msteps.end(), i) - msteps.begin());
            if (q + (size_t)LA < msteps.size() && ops[(size_t)msteps[q + (size_t)LA]].dst >= 0)
                emit_a_load(msteps[q + (size_t)LA]);
            for (int j = i + 1; j < nrec && is_sparse_leaf(ops[(size_t)j]); ++j) emit_gather(j);
        } else {
            if (i + LA < nrec && ops[(size_t)(i + LA)].dst >= 0) emit_a_load(i + LA);
            if (op.obs >= 0 && op.obs + D < K) emit_obs_load(op.obs + D);
        }
        if (op.dst >= 0) dep = "a" + std::to_string(op.dst & 255) + "_0_0";
        o << "    {   // step " << i << ": node " << op.node << "\n";
        stamp(i, 0);
        // own rows of x = (internal ? accumulator : 1) * (observed ? leaf vector : 1)
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 4; ++r) {
                std::ostringstream obs_r;
                if (op.obs >= 0)
                    obs_r << "o" << op.obs << "_" << t << "_" << (r >> 1) << ((r & 1) ? ".y" : ".x");
                o << "    const double x" << t << "_" << r << " = ";
                if (op.pop >= 0) {
                    o << "a" << op.pop << "_" << t << "_" << r;
                    if (op.obs >= 0) o << " * " << obs_r.str();
                } else if (op.obs >= 0) {
                    o << obs_r.str();
                } else {
                    o << "1.0";
                }
                o << ";\n";
            }
        if (op.dst < 0) {
            // root reduction (_mc0_dense.py:184-209), as prune_mfma_kernel
            for (int r = 0; r < 4; ++r)
                o << "    const double w" << r << " = rowok" << r << " ? root_w[16 * m + " << 4 * r
                  << " + (lane >> 4)] : 0.0;\n";
            for (int t = 0; t < T; ++t) {
                o << "    {\n    double sacc = 0.0;\n";
                for (int r = 0; r < 4; ++r) {
                    o << "    negative" << t << " |= rowok" << r << " && (x" << t << "_" << r
                      << " < 0.0);\n";
                    o << "    sacc += w" << r << " * fmax(x" << t << "_" << r << ", 0.0);\n";
                }
                o << "    sacc += __shfl_xor(sacc, 16, 64);\n"
                     "    sacc += __shfl_xor(sacc, 32, 64);\n"
                     "    if (lane < 16) red[" << t << "][m][lane] = sacc;\n    }\n";
            }
            o << "    __syncthreads();\n";
            for (int t = 0; t < T; ++t) {
                o << "    if (m == 0 && lane < 16) {\n        double tot = 0.0;\n";
                for (int mm = 0; mm < NT; ++mm)
                    o << "        tot += red[" << t << "][" << mm << "][lane];\n";
                o << "        lik" << t << " = tot;\n    }\n";
            }
        } else {
            const int b = sparse ? (nmfma++ & 1) : (i & 1);
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < 4; ++r)
                    o << "    xb[" << b << "][" << t << "][(4 * m + " << r << ") * 64 + lane] = x" << t
                      << "_" << r << ";\n";
            o << "    __syncthreads();      // x of every wave, every tile, is in LDS\n";
            stamp(i, 1);
            for (int t = 0; t < T; ++t) o << "    rt_d4 c" << t << " = {0.0, 0.0, 0.0, 0.0};\n";
            for (int kk = 0; kk < KS; ++kk)
                for (int t = 0; t < T; ++t)
                    o << "    c" << t << " = __builtin_amdgcn_mfma_f64_16x16x4f64(A" << i << "_"
                      << (kk >> 1) << ((kk & 1) ? ".y" : ".x") << ", xb[" << b << "][" << t << "]["
                      << kk * 64 << " + lane], c" << t << ", 0, 0, 0);\n";
            stamp(i, 2);
            const int d = op.dst & 255;
            const bool first = (op.dst >> 8) != 0;
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < 4; ++r)
                    o << "    a" << d << "_" << t << "_" << r << (first ? " = " : " *= ") << "c" << t
                      << "[" << r << "];\n";
        }
        o << "    }\n";
    }
    stamp(nrec, 0);

    // lanes 0..15 of wave 0 own the 16 sites of a tile
    for (int t = 0; t < T; ++t) {
        o << "    {\n"
             "    const long site = tile" << t << " * 16 + (lane & 15);\n"
             "    const bool ok = lik" << t << " > 0.0;\n"
             "    double sum = 0.0, nzero = 0.0;\n"
             "    if (m == 0 && lane < 16 && tile" << t << " < nblocks && site < nsites) {\n"
             "        loglik[site] = ok ? log(lik" << t << ") : -__builtin_inf();\n"
             "        status[site] = (ok ? " << RT_SITE_OK << " : " << RT_SITE_ZERO_PROB
          << ") | (negative" << t << " ? " << RT_SITE_NEGATIVE << " : 0);\n"
             "        sum = ok ? log(lik" << t << ") : 0.0;\n"
             "        nzero = ok ? 0.0 : 1.0;\n"
             "    }\n"
             "    for (int off = 32; off > 0; off >>= 1) {\n"
             "        sum += __shfl_xor(sum, off, 64);\n"
             "        nzero += __shfl_xor(nzero, off, 64);\n"
             "    }\n"
             "    if (m == 0 && lane == 0 && tile" << t << " < nblocks) {\n"
             "        partial[tile" << t << " * 2] = sum;\n"
             "        partial[tile" << t << " * 2 + 1] = nzero;\n"
             "    }\n"
             "    }\n";
    }
    o << "}\n";
    return o.str();
}

// ---------------------------------------------------------------------------
// 32 < n <= 64: the split-M kernel, software-pipelined (what C3 / C4 run)
// ---------------------------------------------------------------------------
//
// Same decomposition and arithmetic as rt_jit_mfma_split_source (NT waves, wave m owns
// rows 16m..16m+15, x exchanged through LDS), but the instruction stream is arranged so
// that the matrix pipe does not wait for the exchange.  Per-step clock stamps of the
// kernel above (tools/trace_c3.py, T = 3, one wave per SIMD): 3 220 cycles of back-to-back
// MFMAs per step, then 570 cycles from the last MFMA to the next step (drain, fold into
// the parent's accumulator, fetches) and 550 to the end of the x exchange (multiply,
// LDS write, barrier) with the pipe idle -- 26 % of the step.  Three changes remove the
// serial part:
//   * the STEP ORDER is a list schedule of the tree instead of its post-order: whenever
//     another step is ready, a node's step does not follow its last child's directly
//     (siblings keep their relative order, so every accumulator is folded in the same
//     order as before: results stay bit-identical with the interpreter).  With that,
//     almost no step consumes what the step before it produced;
//   * x of step i+1 is then known BEFORE the MFMAs of step i are issued: it is published
//     (own rows -> xb[(i+1) & 1]) in the shadow of step i's chain, and the barrier that
//     ends step i finds it there;
//   * the fold of step i's result into its parent's accumulator is deferred into the
//     shadow of step i+1's chain (it needs the MFMA results, which drain while the next
//     chain starts).
// A step that does consume its predecessor's result (unavoidable at the top of the tree)
// takes the old serial route for that one step.  One barrier per step.
namespace {

struct pipe_step {
    rt_op op;          // node / obs / pop / dst with slots re-assigned for the new order
    bool late;         // consumes the result of the step before it
};

// list schedule + fresh accumulator slots; returns the steps in issue order
std::vector<pipe_step> pipeline_order(const std::vector<rt_op> &ops, int *nslots_out)
{
    const int nrec = (int)ops.size();
    // the tree from the post-order schedule: a step's parent is the later step that pops
    // the slot it wrote
    std::vector<int> parent((size_t)nrec, -1), nchild((size_t)nrec, 0), prev_sib((size_t)nrec, -1);
    {
        std::vector<int> writer_of_slot(256, -1);      // last pending children per slot
        std::vector<std::vector<int>> pending(256);
        for (int i = 0; i < nrec; ++i) {
            const rt_op &op = ops[(size_t)i];
            if (op.pop >= 0) {
                for (int c : pending[(size_t)op.pop]) parent[(size_t)c] = i;
                nchild[(size_t)i] = (int)pending[(size_t)op.pop].size();
                for (size_t k = 1; k < pending[(size_t)op.pop].size(); ++k)
                    prev_sib[(size_t)pending[(size_t)op.pop][k]] = pending[(size_t)op.pop][k - 1];
                pending[(size_t)op.pop].clear();
            }
            if (op.dst >= 0) pending[(size_t)(op.dst & 255)].push_back(i);
        }
    }
    std::vector<char> done((size_t)nrec, 0);
    std::vector<int> left = nchild;                      // children not yet scheduled
    std::vector<int> order;
    order.reserve((size_t)nrec);
    int last = -1;
    for (int count = 0; count < nrec; ++count) {
        int pick = -1, fallback = -1;
        for (int i = 0; i < nrec; ++i) {
            if (done[(size_t)i] || left[(size_t)i] != 0) continue;
            if (prev_sib[(size_t)i] >= 0 && !done[(size_t)prev_sib[(size_t)i]]) continue;
            if (fallback < 0) fallback = i;
            if (last >= 0 && parent[(size_t)last] == i) continue;    // would consume `last`
            pick = i;
            break;
        }
        if (pick < 0) pick = fallback;
        // bound the look-ahead: a step far beyond the post-order front would keep many
        // accumulators alive; fall back to the front when the pick is > 8 nodes away
        if (fallback >= 0 && pick - fallback > 8) pick = fallback;
        done[(size_t)pick] = 1;
        if (parent[(size_t)pick] >= 0) left[(size_t)parent[(size_t)pick]] -= 1;
        order.push_back(pick);
        last = pick;
    }
    // fresh slots in the new order
    std::vector<int> slot_of((size_t)nrec, -1);
    std::vector<char> used(256, 0);
    int nslots = 1;
    std::vector<pipe_step> out;
    out.reserve((size_t)nrec);
    for (size_t k = 0; k < order.size(); ++k) {
        const int i = order[k];
        pipe_step st;
        st.op = ops[(size_t)i];
        st.late = k > 0 && parent[(size_t)order[k - 1]] == i;
        if (st.op.pop >= 0) {
            st.op.pop = slot_of[(size_t)i];
            used[(size_t)st.op.pop] = 0;                 // free after this step has read it
        }
        if (st.op.dst >= 0) {
            const int p = parent[(size_t)i];
            bool first = false;
            if (slot_of[(size_t)p] < 0) {
                int sl = 0;
                while (used[(size_t)sl]) ++sl;
                used[(size_t)sl] = 1;
                slot_of[(size_t)p] = sl;
                nslots = std::max(nslots, sl + 1);
                first = true;
            }
            st.op.dst = slot_of[(size_t)p] | (first ? 256 : 0);
        }
        out.push_back(st);
    }
    *nslots_out = nslots;
    return out;
}

// The schedule cut at the root into two programs that run as separate workgroups:
// A = the subtrees of all children of the root but the last, B = the subtree of the last
// child, each followed by the root's step.  The root's accumulator is folded in child
// order, ((c1 * c2) ...) * ck, so (product of A's children) * (B's child) is the
// interpreter's product bit for bit.  False: the root has fewer than two children.
bool split_at_root(const std::vector<rt_op> &ops, std::vector<rt_op> *A, std::vector<rt_op> *B)
{
    const int nrec = (int)ops.size();
    if (nrec < 3 || ops[(size_t)nrec - 1].dst >= 0 || ops[(size_t)nrec - 1].pop < 0) return false;
    std::vector<int> parent((size_t)nrec, -1);
    {
        std::vector<std::vector<int>> pending(256);
        for (int i = 0; i < nrec; ++i) {
            const rt_op &op = ops[(size_t)i];
            if (op.pop >= 0) {
                for (int c : pending[(size_t)op.pop]) parent[(size_t)c] = i;
                pending[(size_t)op.pop].clear();
            }
            if (op.dst >= 0) pending[(size_t)(op.dst & 255)].push_back(i);
        }
    }
    int last_child = -1, nchildren = 0;
    for (int i = 0; i < nrec - 1; ++i)
        if (parent[(size_t)i] == nrec - 1) { last_child = i; ++nchildren; }
    if (nchildren < 2) return false;
    A->clear();
    B->clear();
    for (int i = 0; i < nrec - 1; ++i) {
        int top = i;
        while (parent[(size_t)top] != nrec - 1) {
            if (parent[(size_t)top] < 0) return false;      // not a tree below the root
            top = parent[(size_t)top];
        }
        (top == last_child ? B : A)->push_back(ops[(size_t)i]);
    }
    A->push_back(ops[(size_t)nrec - 1]);
    B->push_back(ops[(size_t)nrec - 1]);
    return true;
}

// Observed states at the leaves (the batch's `sparse_ok`): a leaf's message is a column of its
// transition matrix, so a leaf is not a step of the pipeline at all.  The leaves are taken out of
// the schedule and become factors of their parents' expressions, in the order the accumulator
// was folded in (the same product, bit for bit):
//   lead[node of k]   the leaves folded into k's parent's accumulator since the previous matrix
//                     sibling: fold(k) becomes  a = ((a * pc1) * pc2 ...) * c_k  (or, first into
//                     the slot,  a = (pc1 * pc2 ...) * c_k);
//   trail[node of p]  the leaves after p's last matrix child: x(p) = ((a * pc1) * pc2) ...; a
//                     node all of whose children are leaves pops nothing.
struct leaf_ref {
    int node, obs;
};
struct sparse_plan {
    std::vector<std::vector<leaf_ref>> lead, trail;     // by node
};

bool sparse_reduce(const std::vector<rt_op> &ops, std::vector<rt_op> *out, sparse_plan *plan)
{
    const int nrec = (int)ops.size();
    int maxnode = 0;
    for (const rt_op &op : ops) maxnode = std::max(maxnode, (int)op.node);
    plan->lead.resize(std::max(plan->lead.size(), (size_t)maxnode + 1));
    plan->trail.resize(std::max(plan->trail.size(), (size_t)maxnode + 1));
    auto is_leaf = [&](const rt_op &op) { return op.pop < 0 && op.obs >= 0 && op.dst >= 0; };
    std::vector<std::vector<int>> pending(256);
    std::vector<int> pop_eff((size_t)nrec, -1);
    for (int i = 0; i < nrec; ++i) {
        const rt_op &op = ops[(size_t)i];
        if (op.obs >= 0 && !is_leaf(op)) return false;         // an observed inner node
        pop_eff[(size_t)i] = op.pop;
        if (op.pop >= 0) {
            std::vector<leaf_ref> run;
            bool matrix_child = false;
            for (int c : pending[(size_t)op.pop]) {
                const rt_op &ch = ops[(size_t)c];
                if (is_leaf(ch)) {
                    run.push_back({(int)ch.node, (int)ch.obs});
                } else {
                    plan->lead[(size_t)ch.node] = run;
                    run.clear();
                    matrix_child = true;
                }
            }
            plan->trail[(size_t)op.node] = run;
            if (!matrix_child) pop_eff[(size_t)i] = -1;
            pending[(size_t)op.pop].clear();
        }
        if (op.dst >= 0) {
            if (op.dst >> 8) pending[(size_t)(op.dst & 255)].clear();
            pending[(size_t)(op.dst & 255)].push_back(i);
        }
    }
    out->clear();
    for (int i = 0; i < nrec; ++i) {
        if (is_leaf(ops[(size_t)i])) continue;
        rt_op r = ops[(size_t)i];
        r.pop = pop_eff[(size_t)i];
        out->push_back(r);
    }
    return !out->empty();
}

}  // namespace

bool rt_split_at_root(const std::vector<rt_op> &ops, std::vector<rt_op> *A, std::vector<rt_op> *B)
{
    return split_at_root(ops, A, B);
}

// work of the two root programs as steps (A, B); (0, 0) when the schedule cannot be cut
void rt_jit_root_halves(const std::vector<rt_op> &ops, int *stepsA, int *stepsB)
{
    std::vector<rt_op> A, B;
    *stepsA = *stepsB = 0;
    if (!split_at_root(ops, &A, &B)) return;
    *stepsA = (int)A.size() - 1;
    *stepsB = (int)B.size() - 1;
}

// halves: the two root programs of split_at_root as the even / odd workgroups of one
// launch: twice as many workgroups of half the length, so that a batch of a few tiles per
// CU spreads evenly (625 tiles on 256 CUs: three on the busiest CU and 2.44 on average;
// 1 250 half-tiles: 5 halves = 2.5 -- as 250 workgroups of T = 5 half-tiles each, one per
// CU, or as 1 250 workgroups of one, three at a time per CU).  Each writes its root accumulator (own
// rows) to halfbuf[tile][half][k-step][lane]; the module's second kernel, rt_jit_combine,
// multiplies the two and runs the root step and the site epilogue unchanged.
// Root halves, folded: the second workgroup of a pair to arrive finishes the pair's tiles
// itself (root step + site epilogue), so that a pruning launch is ONE kernel; the pair meets
// at a counter in device memory, the shares cross the two workgroups' XCDs as agent-scope
// atomics (write-through / L2-bypassing).
// BUILT, MEASURED, AND NOT THE DEFAULT (RAOTEH_JIT_FOLD=1 turns it on; bit-identical, in the
// -m gpu suite): config 3, 250 workgroups of five half-tiles -- the folded kernel takes
// 176.7 us where the two kernels take 171.1 + 4.1, the step 202.7 us against 203.4: the last
// arrivers read ten shares per lane past the L2 and run the root step while the rest of the
// chip has nothing left to do, which costs what the second launch cost.  With an agent-scope
// release fence per workgroup (an L2 write-back each) the kernel takes 189 us.
bool rt_jit_fold_enabled()
{
    const char *v = getenv("RAOTEH_JIT_FOLD");
    return v && atoi(v) != 0;
}

std::string rt_jit_mfma_split_pipelined_source(const std::vector<rt_op> &ops, int n, int K, int T,
                                               int D, int LA, bool halves, int sparse)
{
    (void)LA;
    const bool fold = halves && !sparse && rt_jit_fold_enabled();
    // x of a step is published one step early, so its leaf vector must be in registers a
    // step earlier than in the serial kernel: at least two leaves ahead
    D = std::max(D, 2);
    const int NT = (n + 15) / 16;
    const int KS = (n + 3) / 4;
    const int KP = (KS + 1) / 2;
    const int XT = NT * 4 * 64;               // doubles of one tile's x image
    // n = 4 KS - 3: the last k-step holds ONE real state (the codon model: state 60 of 61), an
    // MFMA of which three quarters multiply zeros.  Then the chain stops one k-step early and
    // the last state's term is added on the vector ALU when the result is folded:
    // c[r] = fma(P[16 m + 4 r + (lane >> 4)][n - 1], x[n - 1][lane & 15], c[r]) -- the matrix
    // pipe adds the k-steps of a chain in order, each as fused multiply-adds in k order, and
    // the three padded terms are exact zeros, so this is the chain's last MFMA bit for bit
    // (the probe verification agrees: bit-identical at T = 5).  The four P values of a lane sit
    // in its wave's A fragment of that k-step at lanes 4 r + (lane >> 4).
    // MEASURED, AND NOT THE DEFAULT (RAOTEH_JIT_LASTK=1 turns it on): config 3, root halves,
    // T = 5: 176.3 us against 173.1 with the MFMA -- the 16th MFMA of a chain costs 70 pipe
    // cycles, its replacement 8 cross-lane reads per step + 5 LDS reads and 20 v_fma_f64 per
    // step in the shadow, and 16 more live registers in a kernel that already keeps 250
    // values in AGPRs; at T = 1 and T = 2 the register budget no longer holds (scratch: the
    // kernel is rejected and the interpreter runs).
    const bool lastk = (n % 4 == 1) && KS >= 4 && !sparse && getenv("RAOTEH_JIT_LASTK") &&
                       atoi(getenv("RAOTEH_JIT_LASTK")) != 0;
    const int KSM = lastk ? KS - 1 : KS;      // k-steps on the matrix pipe
    int nslots = 1;
    std::vector<std::vector<pipe_step>> programs;
    // (sparse: leaves as factors, see sparse_reduce; one plan per program -- the root's step
    // belongs to both root programs, with different children)
    std::vector<sparse_plan> plans(2);
    size_t cur_prog = 0;
    // gathers of the leaves a step names are requested this many steps ahead
    int GA = 2;
    if (const char *v = getenv("RAOTEH_JIT_GATHER_AHEAD")) GA = std::max(1, std::min(4, atoi(v)));
    if (halves) {
        std::vector<rt_op> opsA, opsB;
        if (!split_at_root(ops, &opsA, &opsB)) return std::string();
        if (sparse) {
            std::vector<rt_op> ra, rb;
            if (!sparse_reduce(opsA, &ra, &plans[0]) || !sparse_reduce(opsB, &rb, &plans[1]))
                return std::string();
            opsA.swap(ra);
            opsB.swap(rb);
        }
        int sa = 1, sb = 1;
        programs.push_back(pipeline_order(opsA, &sa));
        programs.push_back(pipeline_order(opsB, &sb));
        nslots = std::max(sa, sb);
    } else if (sparse) {
        std::vector<rt_op> red;
        if (!sparse_reduce(ops, &red, &plans[0])) return std::string();
        programs.push_back(pipeline_order(red, &nslots));
    } else {
        programs.push_back(pipeline_order(ops, &nslots));
    }
    auto lead_of = [&](const rt_op &op) -> const std::vector<leaf_ref> & {
        static const std::vector<leaf_ref> none;
        const sparse_plan &plan = plans[cur_prog];
        return sparse && (size_t)op.node < plan.lead.size() ? plan.lead[(size_t)op.node] : none;
    };
    auto trail_of = [&](const rt_op &op) -> const std::vector<leaf_ref> & {
        static const std::vector<leaf_ref> none;
        const sparse_plan &plan = plans[cur_prog];
        return sparse && (size_t)op.node < plan.trail.size() ? plan.trail[(size_t)op.node] : none;
    };
    int nrec_max = 0, nrec_all = 0, nlate = 0;
    for (const auto &pr : programs) {
        nrec_max = std::max(nrec_max, (int)pr.size());
        nrec_all += (int)pr.size();
        for (const pipe_step &p : pr) nlate += p.late;
    }
    // a slot popped by step i is read when x(i) is published, which may be one step
    // earlier than in program order of the folds: the deferred fold of step i-1 never
    // targets it (that would make step i late)
    std::ostringstream o;
    o << "// generated by raoteh_amd/csrc/jit.hip (split-M MFMA family, pipelined"
      << (halves ? ", root halves" : "") << (sparse == 2 ? ", leaf state pairs" : sparse ? ", leaf states" : "")
      << "): " << nrec_all
      << " steps (" << nlate << " serial), " << n << " states, " << K << " observed nodes, " << T
      << " tiles per workgroup of " << NT << " waves, " << nslots << " accumulator slots, prefetch "
      << D << " leaves\n";
    o << "typedef double rt_d2 __attribute__((ext_vector_type(2)));\n";
    o << "typedef double rt_d4 __attribute__((ext_vector_type(4)));\n";
    const char *trace_env = getenv("RAOTEH_JIT_TRACE");
    const bool trace = trace_env != nullptr;
    const long trace_wg = trace ? atol(trace_env) : 0;
    if (trace) o << "__device__ unsigned long long rt_trace[" << NT * (nrec_max + 1) * 3 << "];\n";
    auto stamp = [&](int i, int which) {
        if (!trace) return;
        o << "    if (blockIdx.x == " << trace_wg << " && lane == 0) rt_trace[(m * " << (nrec_max + 1)
          << " + " << i << ") * 3 + " << which << "] = __builtin_readcyclecounter();\n";
    };
    // NT <= 4 waves: one wave per SIMD and workgroup, three / two / one workgroups per CU at
    // T = 1 / 2 / more tiles.  NT = 5..8 (64 < n <= 128): two waves on some or all SIMDs and
    // the A fragments of two steps alone are up to 128 registers: one workgroup per CU
    o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * NT
      << ") __attribute__((amdgpu_waves_per_eu("
      << (NT > 4 ? "2, 2" : T == 1 ? "3, 3" : T == 2 ? "2, 2" : "1, 1")
      << ")))\n"
         "rt_jit_prune(const double *__restrict__ Pfrag, const rt_d2 *__restrict__ obs,\n"
         "             const double *__restrict__ root_w, double *__restrict__ loglik,\n"
         "             int *__restrict__ status, double *__restrict__ partial,\n"
         "             long nsites, long nblocks"
      << (halves ? ", double *__restrict__ halfbuf" : "")
      << (fold ? ", int *__restrict__ counters" : "")
      << (sparse ? ", const unsigned *__restrict__ leafw, const double *__restrict__ Pesd" : "")
      << ")\n{\n";
    // two OBJECTS, not one array of two: step i reads xb<i & 1> while x of step i + 1 is
    // written to the other one, and only for distinct objects does the compiler know that
    // an LDS read may be hoisted above an earlier LDS write (with one array every read of
    // the second half of a chain was issued right behind a write and waited for in full)
    // Three of them: the barrier that ends step i stands BEFORE the last k-step of its chain
    // (the first operands of step i + 1 are then fetched under those MFMAs), so a fast wave
    // may already be publishing x of step i + 2 while a slow one still reads x of step i.
    o << "    __shared__ double xb0[" << T << "][" << XT << "];   // [tile][k-step][lane]\n";
    o << "    __shared__ double xb1[" << T << "][" << XT << "];\n";
    o << "    __shared__ double xb2[" << T << "][" << XT << "];\n";
    o << "    __shared__ double red[" << T << "][" << NT << "][16];\n";
    o << "    const int lane = threadIdx.x & 63;\n";
    o << "    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // row tile of this wave\n";
    if (halves) {
        o << "    const long tbase = (long)(blockIdx.x >> 1) * " << T << ";\n";
        o << "    const int half = (int)(blockIdx.x & 1);      // which root program\n";
    } else {
        o << "    const long tbase = (long)blockIdx.x * " << T << ";\n";
    }
    o << "    typedef const __attribute__((address_space(1))) rt_d2 *rt_glb2;\n";
    o << "    rt_glb2 ag = (rt_glb2)Pfrag + (m * " << KP * 64 << " + lane);   // [step][m][k-pair][lane][2]\n";
    o << "    const bool pair1 = 2 * m + 1 < " << KP << ";\n";
    for (int t = 0; t < T; ++t) {
        o << "    const long tile" << t << " = tbase + " << t << ";\n";
        o << "    rt_glb2 g" << t << " = (rt_glb2)obs + (size_t)(tile" << t << " < nblocks ? tile" << t
          << " : nblocks - 1) * " << (long)K * KP * 64 << " + (2 * m * 64 + lane);\n";
    }
    for (int r = 0; r < 4; ++r)
        o << "    const bool rowok" << r << " = 16 * m + " << 4 * r << " + (lane >> 4) < " << n << ";\n";
    for (int t = 0; t < T; ++t) {
        o << "    double lik" << t << " = 0.0;\n    bool negative" << t << " = false;\n";
        for (int sl = 0; sl < nslots; ++sl)
            for (int r = 0; r < 4; ++r)
                o << "    double a" << sl << "_" << t << "_" << r << " = 1.0;\n";
    }
    o << "    const rt_d2 zero2 = {0.0, 0.0};\n";
    // (sparse) the leaf-column table (rt_model::d_Pcol, passed as Pesd): column s of step rec at
    // [(rec n + s) RN + 16 m + 4 (lane >> 4) + r] -- this lane's four rows are one 32-byte load
    if (sparse)
        o << "    const rt_d4 *pcm = (const rt_d4 *)Pesd + (4 * m + (lane >> 4));\n";
    // the root step: x of the root -> weighted sum over the states (_mc0_dense.py:184-209,
    // as prune_mfma_kernel); xp = name prefix of the root's x values (xp_<tile>_<row>)
    auto emit_root_reduce = [&](std::ostream &os, const std::string &xp) {
        for (int r = 0; r < 4; ++r)
            os << "    const double w" << r << " = rowok" << r << " ? root_w[16 * m + " << 4 * r
               << " + (lane >> 4)] : 0.0;\n";
        for (int t = 0; t < T; ++t) {
            os << "    {\n    double sacc = 0.0;\n";
            for (int r = 0; r < 4; ++r) {
                os << "    negative" << t << " |= rowok" << r << " && (" << xp << "_" << t << "_" << r
                   << " < 0.0);\n";
                os << "    sacc += w" << r << " * fmax(" << xp << "_" << t << "_" << r << ", 0.0);\n";
            }
            os << "    sacc += __shfl_xor(sacc, 16, 64);\n"
                  "    sacc += __shfl_xor(sacc, 32, 64);\n"
                  "    if (lane < 16) red[" << t << "][m][lane] = sacc;\n    }\n";
        }
        os << "    __syncthreads();\n";
        for (int t = 0; t < T; ++t) {
            os << "    if (m == 0 && lane < 16) {\n        double tot = 0.0;\n";
            for (int mm = 0; mm < NT; ++mm)
                os << "        tot += red[" << t << "][" << mm << "][lane];\n";
            os << "        lik" << t << " = tot;\n    }\n";
        }
    };
    // lanes 0..15 of wave 0 own the 16 sites of a tile
    auto emit_site_epilogue = [&](std::ostream &os) {
        for (int t = 0; t < T; ++t) {
            os << "    {\n"
                  "    const long site = tile" << t << " * 16 + (lane & 15);\n"
                  "    const bool ok = lik" << t << " > 0.0;\n"
                  "    double sum = 0.0, nzero = 0.0;\n"
                  "    if (m == 0 && lane < 16 && tile" << t << " < nblocks && site < nsites) {\n"
                  "        loglik[site] = ok ? log(lik" << t << ") : -__builtin_inf();\n"
                  "        status[site] = (ok ? " << RT_SITE_OK << " : " << RT_SITE_ZERO_PROB
               << ") | (negative" << t << " ? " << RT_SITE_NEGATIVE << " : 0);\n"
                  "        sum = ok ? log(lik" << t << ") : 0.0;\n"
                  "        nzero = ok ? 0.0 : 1.0;\n"
                  "    }\n"
                  "    for (int off = 32; off > 0; off >>= 1) {\n"
                  "        sum += __shfl_xor(sum, off, 64);\n"
                  "        nzero += __shfl_xor(nzero, off, 64);\n"
                  "    }\n"
                  "    if (m == 0 && lane == 0 && tile" << t << " < nblocks) {\n"
                  "        partial[tile" << t << " * 2] = sum;\n"
                  "        partial[tile" << t << " * 2 + 1] = nzero;\n"
                  "    }\n"
                  "    }\n";
        }
    };
    // the P record of a step is addressed by the step's position in the ORIGINAL schedule
    // (Pfrag is written in that order): recover it from the node
    std::vector<int> rec_of_node;
    {
        int maxnode = 0;
        for (const rt_op &op : ops) maxnode = std::max(maxnode, (int)op.node);
        rec_of_node.assign((size_t)maxnode + 1, -1);
        for (size_t i = 0; i < ops.size(); ++i) rec_of_node[(size_t)ops[i].node] = (int)i;
    }
    for (size_t prog = 0; prog < programs.size(); ++prog) {
    cur_prog = prog;
    const std::vector<pipe_step> &st = programs[prog];
    const int nrec = (int)st.size();
    if (halves) o << (prog == 0 ? "    if (half == 0) {\n" : "    } else {\n");
    // observation stream positions in ISSUE order (the batch was packed in post-order
    // stream order: position = op.obs)
    std::vector<int> obs_order;
    for (const pipe_step &p : st)      // (halves: the root's own observation is the combine kernel's)
        if (p.op.obs >= 0 && !(halves && p.op.dst < 0)) obs_order.push_back(p.op.obs);
    std::vector<int> obs_rank((size_t)std::max(K, 1), -1);
    for (size_t k = 0; k < obs_order.size(); ++k) obs_rank[(size_t)obs_order[k]] = (int)k;
    auto emit_obs_load = [&](std::ostream &os, int pos) {       // pos = stream position (op.obs)
        for (int t = 0; t < T; ++t) {
            os << "    const rt_d2 o" << pos << "_" << t << "_0 = __builtin_nontemporal_load(&g" << t
               << "[" << (long)pos * KP * 64 << "]);\n";
            os << "    const rt_d2 o" << pos << "_" << t << "_1 = pair1 ? __builtin_nontemporal_load(&g"
               << t << "[" << ((long)pos * KP + 1) * 64 << "]) : zero2;\n";
        }
    };
    auto emit_a_load = [&](std::ostream &os, int k) {           // k = issue index
        const int rec = rec_of_node[(size_t)st[(size_t)k].op.node];
        for (int q = 0; q < KP; ++q) {
            const long at = ((long)rec * NT * KP + q) * 64;
            if ((KS & 1) && q == KP - 1)
                os << "    const rt_d2 A" << k << "_" << q << " = "
                   << half_pair_load("ag[" + std::to_string(at) + "]", false) << ";\n";
            else
                os << "    const rt_d2 A" << k << "_" << q << " = ag[" << at << "];\n";
        }
    };
    // (lastk) this lane's four entries of column n - 1 of P, from the A fragment of issue step k
    auto emit_pcol = [&](std::ostream &os, int k) {
        if (!lastk) return;
        for (int r = 0; r < 4; ++r)
            os << "    const double pc" << k << "_" << r << " = __shfl(A" << k << "_" << ((KS - 1) >> 1)
               << (((KS - 1) & 1) ? ".y" : ".x") << ", " << 4 * r << " + (lane >> 4), 64);\n";
    };
    // (sparse) the state words of the leaves, four leaves a word, loaded where first needed in
    // this program; then this lane's four entries of column `state` of the leaf's P (zero in the
    // padded rows): pc<leaf node>_<tile>_<row>
    const int KW = sparse == 2 ? (K + 1) / 2 : (K + 3) / 4;
    std::vector<char> word_seen((size_t)std::max(KW, 1), 0), leaf_seen;
    auto emit_gather = [&](std::ostream &os, const leaf_ref &lf) {
        if ((size_t)lf.node >= leaf_seen.size()) leaf_seen.resize((size_t)lf.node + 1, 0);
        if (leaf_seen[(size_t)lf.node]) return;
        leaf_seen[(size_t)lf.node] = 1;
        const int w = sparse == 2 ? lf.obs >> 1 : lf.obs >> 2;
        const int shf = sparse == 2 ? 16 * (lf.obs & 1) : 8 * (lf.obs & 3);
        if (!word_seen[(size_t)w]) {
            word_seen[(size_t)w] = 1;
            for (int t = 0; t < T; ++t)
                os << "    const unsigned lw" << prog << "_" << w << "_" << t << " = leafw[((size_t)(tile" << t
                   << " < nblocks ? tile" << t << " : nblocks - 1) * " << KW << " + " << w
                   << ") * 16 + (lane & 15)];\n";
        }
        // Column `state` of P from the leaf's A-fragment record (Pfrag[rec][m][q][lane][e2] =
        // P[16 m + (lane & 15)][4 (2 q + e2) + (lane >> 4)], zero in the padding): the four
        // row-lanes of a site then read one 64-byte sector together, where the row-major P costs
        // a sector per lane -- the gathers, not the products, bounded the first version.
        const int rec = rec_of_node[(size_t)lf.node];
        for (int t = 0; t < T; ++t) {
            os << "    const int st" << lf.node << "_" << t << " = (int)((lw" << prog << "_" << w << "_" << t
               << " >> " << shf << ") & 255u);\n";
            os << "    const rt_d4 pf" << lf.node << "_" << t << " = pcm[(" << (long)rec * n << " + st" << lf.node
               << "_" << t << ") * " << 4 * NT << "];\n";
            if (sparse == 2) {
                // the second allowed state (255: none -- the first column again, not added)
                os << "    const int sq" << lf.node << "_" << t << " = (int)((lw" << prog << "_" << w << "_" << t
                   << " >> " << shf + 8 << ") & 255u);\n";
                os << "    const int sr" << lf.node << "_" << t << " = sq" << lf.node << "_" << t
                   << " == 255 ? st" << lf.node << "_" << t << " : sq" << lf.node << "_" << t << ";\n";
                os << "    const rt_d4 pg" << lf.node << "_" << t << " = pcm[(" << (long)rec * n << " + sr" << lf.node
                   << "_" << t << ") * " << 4 * NT << "];\n";
            }
            for (int r = 0; r < 4; ++r) {
                os << "    const double pc" << lf.node << "_" << t << "_" << r << " = pf" << lf.node << "_" << t
                   << "[" << r << "]";
                if (sparse == 2)
                    os << " + (sq" << lf.node << "_" << t << " == 255 ? 0.0 : pg" << lf.node << "_" << t << "["
                       << r << "])";
                os << ";\n";
            }
        }
    };
    auto emit_gathers_of = [&](std::ostream &os, int k) {       // everything issue step k names
        if (!sparse || k < 0 || k >= nrec) return;
        for (const leaf_ref &lf : lead_of(st[(size_t)k].op)) emit_gather(os, lf);
        for (const leaf_ref &lf : trail_of(st[(size_t)k].op)) emit_gather(os, lf);
    };
    // own rows of x for issue step k
    auto emit_x = [&](std::ostream &os, int k) {
        const rt_op &op = st[(size_t)k].op;
        const std::vector<leaf_ref> &tr = trail_of(op);
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 4; ++r) {
                std::ostringstream obs_r;
                if (op.obs >= 0)
                    obs_r << "o" << op.obs << "_" << t << "_" << (r >> 1) << ((r & 1) ? ".y" : ".x");
                os << "    const double x" << k << "_" << t << "_" << r << " = ";
                bool any = false;
                if (op.pop >= 0) {
                    os << "a" << op.pop << "_" << t << "_" << r;
                    if (op.obs >= 0) os << " * " << obs_r.str();
                    any = true;
                } else if (op.obs >= 0) {
                    os << obs_r.str();
                    any = true;
                }
                // (left to right: the order the leaves were folded into the accumulator)
                for (const leaf_ref &lf : tr) {
                    os << (any ? " * " : "") << "pc" << lf.node << "_" << t << "_" << r;
                    any = true;
                }
                if (!any) os << "1.0";
                os << ";\n";
            }
    };
    auto emit_publish = [&](std::ostream &os, int k) {          // x(k) -> xb<k % 3>
        emit_x(os, k);
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 4; ++r)
                os << "    xb" << (k % 3) << "[" << t << "][(4 * m + " << r << ") * 64 + lane] = x" << k
                   << "_" << t << "_" << r << ";\n";
    };
    auto emit_fold = [&](std::ostream &os, int k) {             // c(k) into the parent's accumulator
        const rt_op &op = st[(size_t)k].op;
        const int d = op.dst & 255;
        const bool first = (op.dst >> 8) != 0;
        if (lastk)
            for (int t = 0; t < T; ++t) {
                os << "    const double xl" << k << "_" << t << " = xb" << k % 3 << "[" << t << "]["
                   << (KS - 1) * 64 << " + (lane & 15)];\n";
                for (int r = 0; r < 4; ++r)
                    os << "    c" << k << "_" << t << "[" << r << "] = fma(pc" << k << "_" << r << ", xl" << k
                       << "_" << t << ", c" << k << "_" << t << "[" << r << "]);\n";
            }
        const std::vector<leaf_ref> &ld = lead_of(op);
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 4; ++r) {
                os << "    a" << d << "_" << t << "_" << r << " = ";
                // (the leaves folded since the previous matrix sibling, then this result: the
                // accumulator's own order)
                bool any = false;
                if (!first) {
                    os << "a" << d << "_" << t << "_" << r;
                    any = true;
                }
                for (const leaf_ref &lf : ld) {
                    os << (any ? " * " : "") << "pc" << lf.node << "_" << t << "_" << r;
                    any = true;
                }
                os << (any ? " * " : "") << "c" << k << "_" << t << "[" << r << "];\n";
            }
    };
    // ---- prologue -----------------------------------------------------------------------
    for (int k = 0; k < GA; ++k) emit_gathers_of(o, k);
    for (int k = 0; k < std::min(D, (int)obs_order.size()); ++k) emit_obs_load(o, obs_order[(size_t)k]);
    if (nrec > 0 && st[0].op.dst >= 0) { emit_a_load(o, 0); emit_pcol(o, 0); }
    o << "    {\n";
    if (st[0].op.dst >= 0) {
        emit_publish(o, 0);
        o << "    }\n    __syncthreads();\n";
    } else {
        o << "    }\n";
    }
    bool fold_pending = false;                // fold of step i-1 still to be emitted
    bool prefetched = false;                  // the first two k-steps' operands of step i are in bp<i>
    std::string dep = "lane";
    for (int i = 0; i < nrec; ++i) {
        const rt_op &op = st[(size_t)i].op;
        o << "    asm volatile(\"\" : \"+v\"(ag)";
        for (int t = 0; t < T; ++t) o << ", \"+v\"(g" << t << ")";
        o << " : \"v\"(" << dep << "));\n";
        o << "    __builtin_amdgcn_sched_barrier(0);\n";
        o << "    // ---- step " << i << ": node " << op.node << (st[(size_t)i].late ? " (serial)" : "")
          << "\n";
        stamp(i, 0);
        if (op.dst < 0) {
            // root: everything folded, reduce (_mc0_dense.py:184-209), as prune_mfma_kernel
            if (fold_pending) { emit_fold(o, i - 1); fold_pending = false; }
            if (halves) {
                // this program's share of the root's accumulator (own rows) -> halfbuf
                // (the buffer is padded to whole groups of T tiles: no bounds check)
                for (int t = 0; t < T; ++t) {
                    o << "    {\n    double *hb = halfbuf + ((size_t)tile" << t << " * 2 + " << prog
                      << ") * " << XT << " + (4 * m) * 64 + lane;\n";
                    for (int r = 0; r < 4; ++r) {
                        if (fold) {
                            o << "    __hip_atomic_store(&hb[" << r * 64 << "], a" << op.pop << "_" << t
                              << "_" << r << ", __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n";
                        } else {
                            // (sparse: the root's leaf children of this program are factors)
                            o << "    hb[" << r * 64 << "] = ";
                            bool any = false;
                            if (op.pop >= 0) {
                                o << "a" << op.pop << "_" << t << "_" << r;
                                any = true;
                            }
                            for (const leaf_ref &lf : trail_of(op)) {
                                o << (any ? " * " : "") << "pc" << lf.node << "_" << t << "_" << r;
                                any = true;
                            }
                            if (!any) o << "1.0";
                            o << ";\n";
                        }
                    }
                    o << "    }\n";
                }
                continue;
            }
            emit_x(o, i);
            emit_root_reduce(o, "x" + std::to_string(i));
            continue;
        }
        dep = "c" + std::to_string(i) + "_0[0]";
        const bool have_next = i + 1 < nrec;
        const bool next_root = have_next && st[(size_t)(i + 1)].op.dst < 0;
        const bool next_late = have_next && st[(size_t)(i + 1)].late;
        const int b = i % 3;
        // everything that is not this step's chain goes into its shadow: the fold of the
        // step before, x of the step after, the fetches for later steps -- one statement
        // after each MFMA (in-order issue: a block of them between two MFMAs would hold
        // the pipe up, which is what the first version of this kernel did: 84 cycles per
        // MFMA instead of 67)
        std::ostringstream sh;
        // (the fetches first: a leaf vector is declared before anything that may name it)
        if (op.obs >= 0) {
            const int rank = obs_rank[(size_t)op.obs];
            if (rank + D < (int)obs_order.size()) emit_obs_load(sh, obs_order[(size_t)(rank + D)]);
        }
        if (i + 1 < nrec && st[(size_t)(i + 1)].op.dst >= 0) emit_a_load(sh, i + 1);
        emit_gathers_of(sh, i + GA);
        if (fold_pending) { emit_fold(sh, i - 1); fold_pending = false; }
        if (have_next && !next_root && !next_late) emit_publish(sh, i + 1);
        // (behind everything else: the fragment it reads was requested at the top of this shadow)
        if (i + 1 < nrec && st[(size_t)(i + 1)].op.dst >= 0) emit_pcol(sh, i + 1);
        std::vector<std::string> shadow;
        {
            const std::string all = sh.str();
            size_t pos = 0;
            while (pos < all.size()) {
                const size_t nl = all.find('\n', pos);
                shadow.push_back(all.substr(pos, nl - pos + 1));
                pos = nl + 1;
            }
        }
        for (int t = 0; t < T; ++t) o << "    rt_d4 c" << i << "_" << t << " = {0.0, 0.0, 0.0, 0.0};\n";
        // MFMAs, MFMA results and LDS writes keep their order across these barriers; LDS
        // reads, global reads and scalar instructions may move (0x100 | 0x20 | 0x4)
        const int nmfma = KSM * T;
        const int lead = std::min(2 * T, nmfma);          // MFMAs before the first shadow slice
        const int slots = std::max(1, nmfma - lead - 2 * T);   // ... none behind the last 2 T
        size_t next_sh = 0;
        int issued = 0;
        const bool early_barrier = have_next && !next_root && !next_late && KS >= 4;
        for (int kk = 0; kk < KSM; ++kk)
            for (int t = 0; t < T; ++t) {
                if (early_barrier && issued == nmfma - T) {
                    // the shadow is empty: x of step i + 1 is on its way to LDS
                    while (next_sh < shadow.size()) o << shadow[next_sh++];
                    o << "    __syncthreads();      // x of step " << i + 1 << " is in LDS\n";
                    stamp(i + 1, 1);
                    for (int k2 = 0; k2 < 2; ++k2)
                        for (int t2 = 0; t2 < T; ++t2)
                            o << "    const double bp" << i + 1 << "_" << t2 << "_" << k2 << " = xb"
                              << (i + 1) % 3 << "[" << t2 << "][" << k2 * 64 << " + lane];\n";
                    o << "    __builtin_amdgcn_sched_barrier(0x124);\n";
                }
                std::ostringstream bop;
                if (prefetched && kk < 2) bop << "bp" << i << "_" << t << "_" << kk;
                else bop << "xb" << b << "[" << t << "][" << kk * 64 << " + lane]";
                o << "    c" << i << "_" << t << " = __builtin_amdgcn_mfma_f64_16x16x4f64(A" << i << "_"
                  << (kk >> 1) << ((kk & 1) ? ".y" : ".x") << ", " << bop.str() << ", c" << i << "_"
                  << t << ", 0, 0, 0);\n";
                ++issued;
                if (issued >= lead && next_sh < shadow.size()) {
                    // spread what is left evenly over the MFMAs that are left
                    const int left = std::max(1, slots - (issued - lead));
                    size_t take = (shadow.size() - next_sh + (size_t)left - 1) / (size_t)left;
                    if (issued >= nmfma - 2 * T) take = shadow.size() - next_sh;
                    o << "    __builtin_amdgcn_sched_barrier(0x124);\n";
                    for (size_t q = 0; q < take && next_sh < shadow.size(); ++q) o << shadow[next_sh++];
                    o << "    __builtin_amdgcn_sched_barrier(0x124);\n";
                }
            }
        while (next_sh < shadow.size()) o << shadow[next_sh++];
        stamp(i, 2);
        if (have_next && (next_late || next_root)) {
            // the next step consumes this result: fold now (serial), then publish
            emit_fold(o, i);
            if (!next_root) emit_publish(o, i + 1);
        } else {
            fold_pending = true;
        }
        if (have_next && !next_root && !early_barrier) {
            o << "    __syncthreads();      // x of step " << i + 1 << " is in LDS\n";
            stamp(i + 1, 1);
        }
        prefetched = early_barrier;
    }
    stamp(nrec, 0);
    }   // programs
    if (halves) {
        if (fold) {
            const rt_op &root = ops.back();
            // The shares are agent-scope atomic stores (write-through: they do not sit dirty
            // in this XCD's L2), so what the release needs is that they have completed before
            // the counter moves: a workgroup-scope fence (s_waitcnt) does that.  An agent-scope
            // release fence adds an L2 write-back per workgroup: measured 189 us instead of
            // 175 for the kernel (RAOTEH_JIT_FOLD_FENCE=agent brings it back for A/B runs).
            const char *fs = getenv("RAOTEH_JIT_FOLD_FENCE");
            const std::string scope = (fs && strcmp(fs, "agent") == 0) ? "agent" : "workgroup";
            o << "    }\n";
            o << "    // ---- the pair's second workgroup to arrive finishes its tiles\n"
                 "    __shared__ int arrived;\n"
                 "    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"" << scope << "\");\n"
                 "    __syncthreads();\n"
                 "    if (threadIdx.x == 0)\n"
                 "        arrived = __hip_atomic_fetch_add(&counters[blockIdx.x >> 1], 1, __ATOMIC_RELAXED,\n"
                 "                                         __HIP_MEMORY_SCOPE_AGENT);\n"
                 "    __syncthreads();\n"
                 "    if (arrived == 0) return;\n"
                 "    if (threadIdx.x == 0)          // (for the next launch)\n"
                 "        __hip_atomic_store(&counters[blockIdx.x >> 1], 0, __ATOMIC_RELAXED,\n"
                 "                           __HIP_MEMORY_SCOPE_AGENT);\n"
                 "    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"" << scope << "\");\n";
            for (int t = 0; t < T; ++t) {
                o << "    const double *hx" << t << " = halfbuf + (size_t)tile" << t << " * " << 2 * XT
                  << " + (4 * m) * 64 + lane;\n";
                if (root.obs >= 0) {
                    o << "    const rt_d2 ob" << t << "_0 = g" << t << "[" << (long)root.obs * KP * 64 << "];\n";
                    o << "    const rt_d2 ob" << t << "_1 = pair1 ? g" << t << "["
                      << ((long)root.obs * KP + 1) * 64 << "] : zero2;\n";
                }
                for (int r = 0; r < 4; ++r) {
                    o << "    const double xr_" << t << "_" << r << " = __hip_atomic_load(&hx" << t << "["
                      << r * 64 << "], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * __hip_atomic_load(&hx"
                      << t << "[" << XT + r * 64 << "], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)";
                    if (root.obs >= 0) o << " * ob" << t << "_" << (r >> 1) << ((r & 1) ? ".y" : ".x");
                    o << ";\n";
                }
            }
            emit_root_reduce(o, "xr");
            emit_site_epilogue(o);
            o << "}\n";
        } else {
            o << "    }\n}\n";
        }
        // ---- second kernel of the module: root step + site epilogue from the two halves,
        // one tile per workgroup whatever T is
        T = 1;
        const rt_op &root = ops.back();
        o << "extern \"C\" __global__ void __launch_bounds__(" << 64 * NT << ")\n"
             "rt_jit_combine(const double *__restrict__ halfbuf, const rt_d2 *__restrict__ obs,\n"
             "               const double *__restrict__ root_w, double *__restrict__ loglik,\n"
             "               int *__restrict__ status, double *__restrict__ partial,\n"
             "               long nsites, long nblocks)\n{\n";
        o << "    __shared__ double red[1][" << NT << "][16];\n";
        o << "    const int lane = threadIdx.x & 63;\n";
        o << "    const int m = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);\n";
        o << "    const long tile0 = (long)blockIdx.x;\n";
        for (int r = 0; r < 4; ++r)
            o << "    const bool rowok" << r << " = 16 * m + " << 4 * r << " + (lane >> 4) < " << n << ";\n";
        o << "    double lik0 = 0.0;\n    bool negative0 = false;\n";
        o << "    const double *ha = halfbuf + (size_t)tile0 * " << 2 * XT << " + (4 * m) * 64 + lane;\n";
        if (root.obs >= 0) {
            o << "    typedef const __attribute__((address_space(1))) rt_d2 *rt_glb2;\n";
            o << "    const bool pair1 = 2 * m + 1 < " << KP << ";\n";
            o << "    const rt_d2 zero2 = {0.0, 0.0};\n";
            o << "    rt_glb2 g0 = (rt_glb2)obs + (size_t)tile0 * " << (long)K * KP * 64
              << " + (2 * m * 64 + lane);\n";
            o << "    const rt_d2 ob_0 = g0[" << (long)root.obs * KP * 64 << "];\n";
            o << "    const rt_d2 ob_1 = pair1 ? g0[" << ((long)root.obs * KP + 1) * 64 << "] : zero2;\n";
        }
        for (int r = 0; r < 4; ++r) {
            o << "    const double xr_0_" << r << " = ha[" << r * 64 << "] * ha[" << XT + r * 64 << "]";
            if (root.obs >= 0) o << " * ob_" << (r >> 1) << ((r & 1) ? ".y" : ".x");
            o << ";\n";
        }
        emit_root_reduce(o, "xr");
        emit_site_epilogue(o);
        o << "}\n";
        return o.str();
    }
    emit_site_epilogue(o);
    o << "}\n";
    return o.str();
}

namespace {

// ---- persistent code-object cache -----------------------------------------------------------
// A tree-specialised kernel costs 0.4-3.5 s of hiprtc on a cold process (61 states, five
// half-tiles per workgroup: 3.4 s).  The code object of every kernel that compiled is kept
// in a user cache directory, keyed by a hash of everything that decides it -- the source
// text, the compiler options, the hiprtc version and the target -- and a later process (or
// another context of this one) loads it with hipModuleLoadData in a few milliseconds.
//   RAOTEH_JIT_CACHE_DIR   the directory (default $XDG_CACHE_HOME/raoteh_amd/jit or
//                          ~/.cache/raoteh_amd/jit; created on demand)
//   RAOTEH_JIT_CACHE=0     neither read nor written
// A cached object is data this library wrote for itself; it is still checked like a fresh
// one (scratch attribute, probe verification against the interpreter kernel) before use.
std::string jit_cache_dir()
{
    const char *off = getenv("RAOTEH_JIT_CACHE");
    if (off && atoi(off) == 0 && off[0] != '\0') return std::string();
    if (const char *d = getenv("RAOTEH_JIT_CACHE_DIR")) return d;
    if (const char *x = getenv("XDG_CACHE_HOME")) return std::string(x) + "/raoteh_amd/jit";
    if (const char *h = getenv("HOME")) return std::string(h) + "/.cache/raoteh_amd/jit";
    return std::string();
}

void make_dirs(const std::string &path)
{
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0700);
}

// 128 bits of FNV-1a over (source, options, hiprtc version, target)
std::string jit_cache_key(const std::string &src, bool vgpr_form)
{
    int major = 0, minor = 0;
    hiprtcVersion(&major, &minor);
    char tail[96];
    snprintf(tail, sizeof(tail), "|gfx950|-O3|c++17|vgpr-form=%d|hiprtc %d.%d|layout 1", (int)vgpr_form,
             major, minor);
    unsigned long long h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
    auto feed = [&](const char *p, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            h1 = (h1 ^ (unsigned char)p[i]) * 0x100000001b3ull;
            h2 = (h2 ^ (unsigned char)p[i]) * 0x100000001b3ull + 0x9E3779B97F4A7C15ull;
        }
    };
    feed(src.data(), src.size());
    feed(tail, strlen(tail));
    char out[40];
    snprintf(out, sizeof(out), "%016llx%016llx", h1, h2);
    return out;
}

bool disk_cache_load(const std::string &key, std::vector<char> &code)
{
    const std::string dir = jit_cache_dir();
    if (dir.empty()) return false;
    FILE *f = fopen((dir + "/" + key + ".hsaco").c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long sz = ftell(f);
        if (sz > 64 && sz < (256l << 20) && fseek(f, 0, SEEK_SET) == 0) {
            code.resize((size_t)sz);
            ok = fread(code.data(), 1, (size_t)sz, f) == (size_t)sz &&
                 memcmp(code.data(), "\x7f" "ELF", 4) == 0;
        }
    }
    fclose(f);
    return ok;
}

void disk_cache_store(const std::string &key, const std::vector<char> &code)
{
    const std::string dir = jit_cache_dir();
    if (dir.empty()) return;
    make_dirs(dir);
    char tmp[64];
    snprintf(tmp, sizeof(tmp), "/.%s.%ld.tmp", key.substr(0, 16).c_str(), (long)getpid());
    const std::string tpath = dir + tmp, fpath = dir + "/" + key + ".hsaco";
    FILE *f = fopen(tpath.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    if (fclose(f) != 0 || !ok || rename(tpath.c_str(), fpath.c_str()) != 0) remove(tpath.c_str());
}

// source -> code object: the disk cache, else hiprtc (no lock held: a compile takes seconds)
int jit_compile(const std::string &src, bool mfma, std::vector<char> &code, bool *from_disk)
{
    // MFMA family: results straight into VGPRs (the default picks AGPRs and reads
    // every result back with two v_accvgpr_read_b32: 1 300 moves for 1 240 MFMAs)
    const bool vgpr_form = mfma && !getenv("RAOTEH_JIT_NO_VGPR_FORM");
    const std::string key = jit_cache_key(src, vgpr_form);
    *from_disk = disk_cache_load(key, code);
    if (*from_disk) return RT_OK;
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "rt_jit_prune.hip", 0, nullptr, nullptr) !=
        HIPRTC_SUCCESS) {
        rt_set_error("hiprtcCreateProgram failed");
        return RT_ERR_HIP;
    }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm",
                          "-amdgpu-mfma-vgpr-form=1"};
    const hiprtcResult cr = hiprtcCompileProgram(prog, vgpr_form ? 5 : 3, opts);
    if (cr != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        rt_set_error("hiprtc: %s: %.400s", hiprtcGetErrorString(cr), log.c_str());
        hiprtcDestroyProgram(&prog);
        return RT_ERR_HIP;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    disk_cache_store(key, code);
    return RT_OK;
}

// the cached entry of (ctx, src), under the lock: RT_OK + *fn, RT_ERR_UNSUPPORTED (rejected),
// or 1 = not in the cache
int jit_lookup_locked(const rt_ctx *ctx, const std::string &src, void **fn)
{
    auto it = g_jit_cache.find(std::make_pair(ctx, src));
    if (it == g_jit_cache.end()) return 1;
    if (!it->second.fn || it->second.rejected) {
        rt_set_error(it->second.rejected
                         ? "tree-specialised kernel rejected earlier (differs from the "
                           "interpreter kernel on the probe batch: miscompiled)"
                         : "tree-specialised kernel rejected earlier (register spills)");
        return RT_ERR_UNSUPPORTED;
    }
    it->second.refs += 1;
    *fn = (void *)it->second.fn;
    return RT_OK;
}

}  // namespace

// compile (or fetch from the per-context cache / the disk cache) and return the function
int rt_jit_get(rt_ctx *ctx, const std::string &src, void **fn, bool mfma, double *compile_s)
{
    const auto t_begin = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        const int rc = jit_lookup_locked(ctx, src, fn);
        if (rc != 1) return rc;
        // A module stays loaded while a batch may still launch it (refs).  When the cache
        // is full, the kernels no batch refers to any more are dropped (a program that
        // keeps changing its tree, e.g. MCMC over topologies); if every one is in use the
        // new batch runs the interpreter kernels.
        if (g_jit_cache.size() >= 256) {
            for (auto e = g_jit_cache.begin(); e != g_jit_cache.end();) {
                if (e->second.refs == 0) {
                    if (e->second.module) hipModuleUnload(e->second.module);
                    e = g_jit_cache.erase(e);
                } else {
                    ++e;
                }
            }
            if (g_jit_cache.size() >= 256) {
                rt_set_error("tree-specialised kernel cache is full (256 kernels in use)");
                return RT_ERR_UNSUPPORTED;
            }
        }
    }
    std::vector<char> code;
    bool from_disk = false;
    RT_TRY(jit_compile(src, mfma, code, &from_disk));
    jit_entry e;
    RT_HIP(hipSetDevice(ctx->device));           // (a background compile thread starts without one)
    RT_HIP(hipModuleLoadData(&e.module, code.data()));
    RT_HIP(hipModuleGetFunction(&e.fn, e.module, "rt_jit_prune"));
    // A kernel that needs scratch memory (register spills) is never run: it would be
    // slower than the interpreter kernels, and with the whole register file in use
    // and -amdgpu-mfma-vgpr-form such a kernel has produced wrong results
    // (tests/soak/soak.py: 32 states, 4 tiles per wave).  The rejection is cached too.
    int scratch = 0;
    if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, e.fn) != hipSuccess)
        scratch = -1;
    if (scratch != 0 && !getenv("RAOTEH_JIT_ALLOW_SCRATCH")) {   // the override: diagnostics only
        hipModuleUnload(e.module);
        e.module = nullptr;
        e.fn = nullptr;
    }
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    {
        // another thread (a background compile of the same source) may have got here first
        void *other = nullptr;
        const int rc = jit_lookup_locked(ctx, src, &other);
        if (rc != 1) {
            if (e.module) hipModuleUnload(e.module);
            if (rc == RT_OK) *fn = other;
            return rc;
        }
    }
    e.refs = e.fn ? 1 : 0;
    g_jit_cache[std::make_pair((const rt_ctx *)ctx, src)] = e;
    if (!e.fn) {
        rt_set_error("tree-specialised kernel rejected: %d bytes of scratch per work-item", scratch);
        return RT_ERR_UNSUPPORTED;
    }
    *fn = (void *)e.fn;
    if (compile_s)
        *compile_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return RT_OK;
}

// ---- background compilation -----------------------------------------------------------------
// rt_sites_create does not wait for hiprtc: a job on a host thread of THIS process works
// through the candidate sources of the batch (fewer tiles if one spills) -- compile or disk
// cache, module load, scratch check, into the per-context cache -- while the batch runs the
// interpreter kernel; the caller's thread swaps the kernel in at a later rt_prune / rt_step
// (api.hip rt_sites_jit_poll: probe verification first).  Never a re-exec, never a fork.
struct rt_jit_job {
    rt_ctx *ctx = nullptr;
    std::vector<std::string> sources;
    bool mfma = false;
    std::thread worker;
    std::atomic<int> done{0};
    int chosen = -1;              // index of the first candidate that compiled without scratch
    int rc = RT_ERR_UNSUPPORTED;
    double seconds = 0.0;
    std::string error;
    // (a job nobody waited for is joined when the last reference goes -- at the latest when
    // this library's statics are destroyed at exit, which is before the HIP runtime's, loaded
    // earlier: a joinable std::thread must not be destroyed)
    ~rt_jit_job()
    {
        if (worker.joinable()) worker.join();
    }
};

namespace {
std::mutex g_jobs_mutex;
std::vector<std::shared_ptr<rt_jit_job>> g_jobs;     // every job not yet joined
}

std::shared_ptr<rt_jit_job> rt_jit_start(rt_ctx *ctx, std::vector<std::string> sources, bool mfma)
{
    auto job = std::make_shared<rt_jit_job>();
    job->ctx = ctx;
    job->sources = std::move(sources);
    job->mfma = mfma;
    rt_jit_job *j = job.get();
    job->worker = std::thread([j]() {
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < j->sources.size(); ++k) {
            void *fn = nullptr;
            const int rc = rt_jit_get(j->ctx, j->sources[k], &fn, j->mfma, nullptr);
            j->rc = rc;
            if (rc == RT_OK) {
                rt_jit_ref(j->ctx, fn, -1);       // the batches take their own references
                j->chosen = (int)k;
                break;
            }
            j->error = rt_last_error();
            if (rc != RT_ERR_UNSUPPORTED) break;  // a compiler error, not a spill
        }
        j->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        j->done.store(1, std::memory_order_release);
    });
    std::lock_guard<std::mutex> lock(g_jobs_mutex);
    // finished jobs nobody holds any more (their batches are gone): join and drop them, so
    // that a long run over many trees does not keep every job's source texts
    for (auto it = g_jobs.begin(); it != g_jobs.end();) {
        if ((*it)->done.load(std::memory_order_acquire) && it->use_count() == 1) {
            if ((*it)->worker.joinable()) (*it)->worker.join();
            it = g_jobs.erase(it);
        } else {
            ++it;
        }
    }
    g_jobs.push_back(job);
    return job;
}

// background compiles still running (whatever their context): rt_sites_create starts no
// further one beyond the limit of api.hip -- every job is a host thread inside hiprtc
int rt_jit_jobs_pending()
{
    std::lock_guard<std::mutex> lock(g_jobs_mutex);
    int pending = 0;
    for (const auto &j : g_jobs)
        if (!j->done.load(std::memory_order_acquire)) ++pending;
    return pending;
}

// A compile thread must not be inside hiprtc when the process runs its exit handlers (the
// compiler's own lazily created statics are destroyed before this library's): a host program
// that lets batches compile in the background calls this before it exits -- the Python
// binding registers it with atexit.
extern "C" int rt_jit_wait_all(void)
{
    rt_jit_join_all(nullptr);
    return RT_OK;
}

bool rt_jit_job_done(rt_jit_job *job, bool wait)
{
    if (!job) return true;
    if (wait) {
        std::lock_guard<std::mutex> lock(g_jobs_mutex);      // one joiner at a time
        if (job->worker.joinable()) job->worker.join();
    }
    return job->done.load(std::memory_order_acquire) != 0;
}

void rt_jit_job_result(rt_jit_job *job, int *rc, int *chosen, double *seconds, std::string *error)
{
    *rc = job->rc;
    *chosen = job->chosen;
    *seconds = job->seconds;
    *error = job->error;
}

// rt_ctx_destroy: no thread of this context may outlive it (ctx == nullptr: every job)
void rt_jit_join_all(const rt_ctx *ctx)
{
    std::vector<std::shared_ptr<rt_jit_job>> mine;
    {
        std::lock_guard<std::mutex> lock(g_jobs_mutex);
        for (auto it = g_jobs.begin(); it != g_jobs.end();) {
            if (!ctx || (*it)->ctx == ctx || (*it)->done.load()) {
                mine.push_back(*it);
                it = g_jobs.erase(it);
            } else {
                ++it;
            }
        }
    }
    for (auto &j : mine)
        if (j->worker.joinable()) j->worker.join();
}

// 1: compiled and usable, -1: compiled and rejected (spills / failed verification), 0: unknown
int rt_jit_cached(const rt_ctx *ctx, const std::string &src)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    auto it = g_jit_cache.find(std::make_pair(ctx, src));
    if (it == g_jit_cache.end()) return 0;
    return it->second.fn && !it->second.rejected ? 1 : -1;
}

// a batch (or its clone) takes / gives back its reference to a kernel
void rt_jit_ref(const rt_ctx *ctx, void *fn, int delta)
{
    if (!fn) return;
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (auto &kv : g_jit_cache) {
        if (kv.first.first == ctx && (void *)kv.second.fn == fn) {
            kv.second.refs += delta;
            return;
        }
    }
}

int rt_jit_companion(const rt_ctx *ctx, void *fn, const char *name, void **out)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (auto &kv : g_jit_cache)
        if (kv.first.first == ctx && (void *)kv.second.fn == fn) {
            hipFunction_t f = nullptr;
            RT_HIP(hipModuleGetFunction(&f, kv.second.module, name));
            *out = (void *)f;
            return RT_OK;
        }
    rt_set_error("kernel not found");
    return RT_ERR_INVALID;
}

int rt_jit_verified(const rt_ctx *ctx, void *fn)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (auto &kv : g_jit_cache)
        if (kv.first.first == ctx && (void *)kv.second.fn == fn) return kv.second.verified ? 1 : 0;
    return 0;
}

void rt_jit_set_verified(const rt_ctx *ctx, void *fn, bool ok)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (auto &kv : g_jit_cache)
        if (kv.first.first == ctx && (void *)kv.second.fn == fn) {
            kv.second.verified = ok;
            kv.second.rejected = !ok;       // the module stays loaded until the context goes
            return;
        }
}

// diagnostics: copy a __device__ variable of the kernel's module to the host
int rt_jit_read_global(const rt_ctx *ctx, void *fn, const char *name, void *dst, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (auto &kv : g_jit_cache)
        if (kv.first.first == ctx && (void *)kv.second.fn == fn) {
            hipDeviceptr_t ptr = nullptr;
            size_t sz = 0;
            RT_HIP(hipModuleGetGlobal(&ptr, &sz, kv.second.module, name));
            RT_REQUIRE(bytes <= sz, "global %s has %zu bytes", name, sz);
            RT_HIP(hipMemcpy(dst, ptr, bytes, hipMemcpyDeviceToHost));
            return RT_OK;
        }
    rt_set_error("kernel not found");
    return RT_ERR_INVALID;
}

void rt_jit_release(const rt_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_jit_mutex);
    for (auto it = g_jit_cache.begin(); it != g_jit_cache.end();) {
        if (it->first.first == ctx) {
            if (it->second.module) hipModuleUnload(it->second.module);
            it = g_jit_cache.erase(it);
        } else {
            ++it;
        }
    }
}

int rt_launch_prune_jit(rt_model *m, rt_sites *s, const rt_fuse_args *fuse)
{
    if (s->jit_fused && s->layout == RT_LAYOUT_LANE) {
        // lane kernel with the fused prologue: the same kernel runs a plain pruning launch
        // (fq == nullptr: it copies the resident P table) and a whole step
        static const rt_fuse_args none;
        const rt_fuse_args &f = fuse ? *fuse : none;
        const double *Pord = m->d_Pfrag, *obs = s->d_obs, *root_w = m->d_root;
        double *loglik = s->d_loglik, *partial = s->d_partial;
        int *status = s->d_status;
        long nsites = (long)s->nsites, nblocks = (long)s->nblocks;
        const double *fq = f.expm ? m->d_Q : nullptr, *ftt = m->d_t_step;
        const int *fqidx = m->d_qidx_step;
        double *fP = m->d_P, *fPord = m->d_Pfrag;
        int *finfo = m->d_info;
        const double *rp = f.red.partial;
        long rn = f.red.npartials;
        double *rtot = f.red.totals;
        double rns = f.red.nsites;
        void *args[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks,
                        &fq, &fqidx, &ftt, &fP, &fPord, &finfo, &rp, &rn, &rtot, &rns};
        const int wg = s->jit_waves;
        const unsigned groups = (unsigned)((s->nblocks + wg - 1) / wg) + (rp ? 1u : 0u);
        if (m->ctx->ev_start)
            RT_HIP(hipExtModuleLaunchKernel((hipFunction_t)s->jit_fn, groups * 64u * wg, 1, 1, 64 * wg,
                                            1, 1, 0, m->ctx->stream, args, nullptr,
                                            m->ctx->ev_start, m->ctx->ev_stop, 0));
        else
            RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_fn, groups, 1, 1, 64 * wg, 1, 1, 0,
                                         m->ctx->stream, args, nullptr));
        return RT_OK;
    }
    const double *Pord = s->jit_quad ? m->d_Pquad : m->d_Pfrag;
    const double *obs = s->d_obs;
    const double *root_w = m->d_root;
    double *loglik = s->d_loglik;
    int *status = s->d_status;
    double *partial = s->d_partial;
    long nsites = (long)s->nsites;
    long nblocks = (long)s->nblocks;
    long tile0 = 0, stride1 = 1;     // declared by the one-wave MFMA family only
    // the column-gathering split-M kernels declare the leaf-state words and the transition
    // matrices in the reference's order in those two places
    const unsigned *leafw = s->d_leafw;
    const double *Pesd = (s->jit_sparse && !(m->n <= 32 && s->mfma_solo) && m->d_Pcol) ? m->d_Pcol : m->d_P;
    void *args_dense[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks, &tile0,
                          &stride1};
    void *args_sparse[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks, &leafw,
                           &Pesd};
    // (the one-wave family's leaf-state kernels: its own two arguments, then these)
    void *args_solo_sparse[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks,
                                &tile0, &stride1, &leafw, &Pesd};
    const bool solo_family = m->n <= 32 && s->mfma_solo;
    void **args = s->jit_sparse ? (solo_family ? args_solo_sparse : args_sparse) : args_dense;
    if (s->jit_fn2) {
        // main kernel: jit_split_tiles tiles, jit_tiles per wave (one wave per SIMD); the rest
        // one tile per wave on the side stream, at the same time.  The timing events, when
        // this launch is sampled, stand around both.
        rt_ctx *ctx = m->ctx;
        if (!ctx->stream2) {
            RT_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            RT_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            RT_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        if (ctx->ev_start) RT_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
        RT_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        RT_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        const unsigned main_groups = (unsigned)(s->jit_split_tiles / s->jit_tiles);
        RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_fn, main_groups, 1, 1, 64, 1, 1, 0,
                                     ctx->stream, args, nullptr));
        // the few tail waves as every stride-th workgroup of a launch as wide as the main one:
        // launched dense, the dispatcher put all of them on the first free slots it found -- a
        // handful of CUs -- and those SIMDs carried six tiles (87 us instead of 67)
        long tail0 = (long)s->jit_split_tiles;
        const long tail_waves = (s->nblocks - s->jit_split_tiles + s->jit_tiles2 - 1) / s->jit_tiles2;
        long stride = std::max<long>(1, (long)main_groups / tail_waves);
        if (const char *v = getenv("RAOTEH_JIT_TAIL_STRIDE")) stride = std::max(1, atoi(v));
        void *args2[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks, &tail0,
                         &stride};
        RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_fn2, (unsigned)(tail_waves * stride), 1, 1,
                                     64, 1, 1, 0, ctx->stream2, args2, nullptr));
        RT_HIP(hipEventRecord(ctx->ev_join, ctx->stream2));
        RT_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        if (ctx->ev_stop) RT_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
        return RT_OK;
    }
    if (s->jit_halves) {
        // split-M family, root halves: 2 workgroups per tile, then the combine kernel (one
        // per tile); in a sampled launch each kernel gets its own begin / end stamps (the
        // pruning slot holds the first kernel, RT_K_COMBINE the second)
        double *half = s->d_half;
        const double *chalf = s->d_half;
        int *counters = s->d_half_count;
        void *hargs_dense[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks, &half,
                               &counters};
        void *hargs_sparse[] = {&Pord, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks,
                                &half, &leafw, &Pesd};
        void **hargs = s->jit_sparse ? hargs_sparse : hargs_dense;
        if (s->jit_fold) {
            // one kernel: the second workgroup of every pair finishes the pair's tiles
            const unsigned tpb = 64u * (unsigned)s->jit_waves;
            const unsigned groups = (unsigned)((s->nblocks + s->jit_tiles - 1) / s->jit_tiles);
            if (m->ctx->ev_start)
                RT_HIP(hipExtModuleLaunchKernel((hipFunction_t)s->jit_fn, 2u * groups * tpb, 1, 1, tpb,
                                                1, 1, 0, m->ctx->stream, hargs, nullptr,
                                                m->ctx->ev_start, m->ctx->ev_stop, 0));
            else
                RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_fn, 2u * groups, 1, 1, tpb, 1, 1, 0,
                                             m->ctx->stream, hargs, nullptr));
            return RT_OK;
        }
        void *cargs[] = {&chalf, &obs, &root_w, &loglik, &status, &partial, &nsites, &nblocks};
        const unsigned tpb = 64u * (unsigned)s->jit_waves;
        const unsigned tiles = (unsigned)s->nblocks;
        const unsigned groups = (unsigned)((s->nblocks + s->jit_tiles - 1) / s->jit_tiles);
        if (m->ctx->ev_start) {
            hipEvent_t ca = nullptr, cb = nullptr;
            rt_time_extra_begin(m->ctx, RT_K_COMBINE, "rt_jit_combine", &ca, &cb);
            RT_HIP(hipExtModuleLaunchKernel((hipFunction_t)s->jit_fn, 2u * groups * tpb, 1, 1, tpb, 1, 1,
                                            0, m->ctx->stream, hargs, nullptr, m->ctx->ev_start,
                                            m->ctx->ev_stop, 0));
            RT_HIP(hipExtModuleLaunchKernel((hipFunction_t)s->jit_combine, tiles * tpb, 1, 1, tpb, 1, 1,
                                            0, m->ctx->stream, cargs, nullptr, ca, cb, 0));
            rt_time_extra_end(m->ctx, RT_K_COMBINE, ca, cb);
        } else {
            RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_fn, 2u * groups, 1, 1, tpb, 1, 1, 0,
                                         m->ctx->stream, hargs, nullptr));
            RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_combine, tiles, 1, 1, tpb, 1, 1, 0,
                                         m->ctx->stream, cargs, nullptr));
        }
        return RT_OK;
    }
    // lane family: jit_waves waves of one site block each per workgroup; MFMA family,
    // n <= 32: one wave of jit_tiles site tiles per workgroup; n > 32: jit_waves = NT
    // waves share jit_tiles tiles
    const int wg = (s->layout == RT_LAYOUT_LANE || !s->mfma_solo) ? s->jit_waves : 1;
    const long per = s->layout == RT_LAYOUT_LANE ? wg : s->jit_tiles;
    // global size in work-items; the events (null unless this launch is sampled) get
    // the kernel's own begin / end
    const unsigned groups = (unsigned)((s->nblocks + per - 1) / per);
    if (m->ctx->ev_start)
        RT_HIP(hipExtModuleLaunchKernel((hipFunction_t)s->jit_fn, groups * 64u * wg, 1, 1, 64 * wg,
                                        1, 1, 0, m->ctx->stream, args, nullptr,
                                        m->ctx->ev_start, m->ctx->ev_stop, 0));
    else
        RT_HIP(hipModuleLaunchKernel((hipFunction_t)s->jit_fn, groups, 1, 1, 64 * wg, 1, 1, 0,
                                     m->ctx->stream, args, nullptr));
    return RT_OK;
}
