/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the tree-CTMC likelihood hot path of argriffing/raoteh,
 * scalar, single thread.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load liboracle.so.  Parity status: PINNED --
 * tests/test_oracle_c.py checks every function against the golden vectors
 * generated from the reference (tests/golden, tools/gen_golden.py).
 *
 * Reference lines restated (relative to the reference repository root):
 *   orc_expm            scipy.linalg.expm(Q*t) as called at
 *                       raoteh/sampler/_mjp_dense.py:24-25 -- algorithm: Higham
 *                       2005 scaling and squaring with [m/m] Pade, m in
 *                       {3,5,7,9,13} (scipy implements the Al-Mohy/Higham 2009
 *                       refinement of the same method; both are accurate to
 *                       rounding, so results agree to ~1e-15 relative to |P|)
 *   orc_upward          pyfelscore.mcy_esd_get_node_to_pmap as called at
 *                       raoteh/sampler/_mcy_dense.py:286; twins _mcx.py:188-210,
 *                       _mcy.py:657-679; type-z factor _mcz.py:159-160
 *   orc_root            raoteh/sampler/_mc0_dense.py:184-209
 *   orc_site_faithful   one call of _mjp_dense.get_likelihood
 *                       (raoteh/sampler/_mjp_dense.py:362-407): expm of EVERY
 *                       edge, then the upward pass, for ONE site
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double THETA[5] = {1.495585217958292e-2, 2.539398330063230e-1,
                                9.504178996162932e-1, 2.097847961257068e0,
                                5.371920351148152e0};
static const double B3[] = {120., 60., 12., 1.};
static const double B5[] = {30240., 15120., 3360., 420., 30., 1.};
static const double B7[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
static const double B9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240.,
                            2162160., 110880., 3960., 90., 1.};
static const double B13[] = {64764752532480000., 32382376266240000., 7771770303897600.,
                             1187353796428800., 129060195264000., 10559470521600.,
                             670442572800., 33522128640., 1323241920., 40840800.,
                             960960., 16380., 182., 1.};

static void matmul(int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < n; ++i) {
        double *c = C + (size_t)i * n;
        for (int j = 0; j < n; ++j) c[j] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double a = A[(size_t)i * n + k];
            const double *b = B + (size_t)k * n;
            for (int j = 0; j < n; ++j) c[j] += a * b[j];
        }
    }
}

/* solve M X = R in place (X overwrites R) by Gaussian elimination with partial
 * pivoting; returns 0 or -1 if singular */
static int solve(int n, double *M, double *R)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(M[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(M[(size_t)i * n + k]) > best) { best = fabs(M[(size_t)i * n + k]); p = i; }
        if (!(best > 0.0)) return -1;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                double t = M[(size_t)k * n + j]; M[(size_t)k * n + j] = M[(size_t)p * n + j]; M[(size_t)p * n + j] = t;
                t = R[(size_t)k * n + j]; R[(size_t)k * n + j] = R[(size_t)p * n + j]; R[(size_t)p * n + j] = t;
            }
        for (int i = k + 1; i < n; ++i) {
            const double f = M[(size_t)i * n + k] / M[(size_t)k * n + k];
            if (f == 0.0) continue;
            for (int j = k + 1; j < n; ++j) M[(size_t)i * n + j] -= f * M[(size_t)k * n + j];
            for (int j = 0; j < n; ++j) R[(size_t)i * n + j] -= f * R[(size_t)k * n + j];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        for (int j = 0; j < n; ++j) {
            double s = R[(size_t)k * n + j];
            for (int i = k + 1; i < n; ++i) s -= M[(size_t)k * n + i] * R[(size_t)i * n + j];
            R[(size_t)k * n + j] = s / M[(size_t)k * n + k];
        }
    }
    return 0;
}

/* P = expm(Q * t); info[0] = Pade degree, info[1] = squarings.  work: 7*n*n doubles */
int orc_expm(int n, const double *Q, double t, double *P, double *work, int *info)
{
    const size_t nn = (size_t)n * n;
    double *A = work, *A2 = A + nn, *A4 = A2 + nn, *A6 = A4 + nn, *U = A6 + nn,
           *V = U + nn, *W = V + nn;
    for (size_t e = 0; e < nn; ++e) A[e] = Q[e] * t;
    double nrm = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += fabs(A[(size_t)i * n + j]);
        if (s > nrm) nrm = s;
    }
    int m = 13, s = 0;
    if (nrm <= THETA[0]) m = 3;
    else if (nrm <= THETA[1]) m = 5;
    else if (nrm <= THETA[2]) m = 7;
    else if (nrm <= THETA[3]) m = 9;
    else if (nrm > THETA[4]) {
        s = (int)ceil(log2(nrm / THETA[4]));
        if (s < 0) s = 0;
    }
    if (info) { info[0] = m; info[1] = s; }
    if (s) {
        const double sc = ldexp(1.0, -s);
        for (size_t e = 0; e < nn; ++e) A[e] *= sc;
    }
    matmul(n, A, A, A2);
    if (m == 13) {
        matmul(n, A2, A2, A4);
        matmul(n, A4, A2, A6);
        for (size_t e = 0; e < nn; ++e) W[e] = B13[13] * A6[e] + B13[11] * A4[e] + B13[9] * A2[e];
        matmul(n, A6, W, U);
        for (size_t e = 0; e < nn; ++e) U[e] += B13[7] * A6[e] + B13[5] * A4[e] + B13[3] * A2[e];
        for (int i = 0; i < n; ++i) U[(size_t)i * n + i] += B13[1];
        memcpy(W, U, nn * sizeof(double));
        matmul(n, A, W, U);
        for (size_t e = 0; e < nn; ++e) W[e] = B13[12] * A6[e] + B13[10] * A4[e] + B13[8] * A2[e];
        matmul(n, A6, W, V);
        for (size_t e = 0; e < nn; ++e) V[e] += B13[6] * A6[e] + B13[4] * A4[e] + B13[2] * A2[e];
        for (int i = 0; i < n; ++i) V[(size_t)i * n + i] += B13[0];
    } else {
        const double *b = m == 3 ? B3 : m == 5 ? B5 : m == 7 ? B7 : B9;
        double *A8 = P;                        /* borrowed as scratch */
        if (m >= 5) matmul(n, A2, A2, A4);
        if (m >= 7) matmul(n, A4, A2, A6);
        if (m >= 9) matmul(n, A6, A2, A8);
        for (size_t e = 0; e < nn; ++e) {
            double w = b[3] * A2[e], v = b[2] * A2[e];
            if (m >= 5) { w += b[5] * A4[e]; v += b[4] * A4[e]; }
            if (m >= 7) { w += b[7] * A6[e]; v += b[6] * A6[e]; }
            if (m >= 9) { w += b[9] * A8[e]; v += b[8] * A8[e]; }
            W[e] = w; V[e] = v;
        }
        for (int i = 0; i < n; ++i) { W[(size_t)i * n + i] += b[1]; V[(size_t)i * n + i] += b[0]; }
        matmul(n, A, W, U);
    }
    for (size_t e = 0; e < nn; ++e) { const double u = U[e], v = V[e]; A2[e] = v - u; P[e] = v + u; }
    if (solve(n, A2, P) != 0) return -1;
    for (int q = 0; q < s; ++q) { matmul(n, P, P, W); memcpy(P, W, nn * sizeof(double)); }
    return 0;
}

/* upward pass for one site.  obs (optional) f64[nnodes][n] likelihood per node
 * and state (ones where unobserved); mask (optional) int64[nnodes][n].
 * pmap f64[nnodes][n] is written for every node. */
void orc_upward(int64_t nnodes, int n, const int64_t *idx, const int64_t *ptr,
                const double *esd, const int64_t *mask, const double *obs, double *pmap)
{
    for (int64_t v = nnodes - 1; v >= 0; --v) {
        double *L = pmap + (size_t)v * n;
        for (int s = 0; s < n; ++s) L[s] = 1.0;
        for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) {
            const int64_t c = idx[e];
            const double *Pc = esd + (size_t)c * n * n;
            const double *Lc = pmap + (size_t)c * n;
            for (int s = 0; s < n; ++s) {
                double sum = 0.0;
                for (int sp = 0; sp < n; ++sp) sum += Pc[(size_t)s * n + sp] * Lc[sp];
                L[s] *= sum;
            }
        }
        if (obs) for (int s = 0; s < n; ++s) L[s] *= obs[(size_t)v * n + s];
        if (mask) for (int s = 0; s < n; ++s) if (!mask[(size_t)v * n + s]) L[s] = 0.0;
    }
}

double orc_root(int n, const double *root_pmap, const double *root_w)
{
    double lik = 0.0;
    for (int s = 0; s < n; ++s) {
        const double x = root_pmap[s] > 0.0 ? root_pmap[s] : 0.0;
        lik += (root_w ? root_w[s] : 1.0) * x;
    }
    return lik;
}

/* amortised batch: esd given; obs_dense f64[nsites][nobs][n] at nodes obs_nodes */
int orc_batch_loglik(int64_t nnodes, int n, const int64_t *idx, const int64_t *ptr,
                     const double *esd, int64_t nobs, const int64_t *obs_nodes,
                     const double *obs_dense, int64_t nsites, const double *root_w,
                     double *loglik, int32_t *status)
{
    double *pmap = (double *)malloc((size_t)nnodes * n * sizeof(double));
    double *obs = (double *)malloc((size_t)nnodes * n * sizeof(double));
    if (!pmap || !obs) { free(pmap); free(obs); return -1; }
    for (int64_t i = 0; i < nsites; ++i) {
        for (size_t e = 0; e < (size_t)nnodes * n; ++e) obs[e] = 1.0;
        for (int64_t k = 0; k < nobs; ++k)
            memcpy(obs + (size_t)obs_nodes[k] * n, obs_dense + ((size_t)i * nobs + k) * n,
                   (size_t)n * sizeof(double));
        orc_upward(nnodes, n, idx, ptr, esd, NULL, obs, pmap);
        const double lik = orc_root(n, pmap, root_w);
        loglik[i] = lik > 0.0 ? log(lik) : -INFINITY;
        status[i] = lik > 0.0 ? 0 : 1;
    }
    free(pmap); free(obs);
    return 0;
}

/* reference-faithful batch: the E per-edge expm calls are repeated for every
 * site, as the reference's per-site entry point does.  Q f64[nq][n][n],
 * node_q int64[nnodes], t f64[nnodes] (entry 0 ignored). */
int orc_batch_loglik_faithful(int64_t nnodes, int n, const int64_t *idx, const int64_t *ptr,
                              const double *Q, const int64_t *node_q, const double *t,
                              int64_t nobs, const int64_t *obs_nodes, const double *obs_dense,
                              int64_t nsites, const double *root_w, double *loglik,
                              int32_t *status)
{
    const size_t nn = (size_t)n * n;
    double *esd = (double *)calloc((size_t)nnodes * nn, sizeof(double));
    double *work = (double *)malloc(7 * nn * sizeof(double));
    if (!esd || !work) { free(esd); free(work); return -1; }
    int rc = 0;
    for (int64_t i = 0; i < nsites && rc == 0; ++i) {
        for (int64_t v = 1; v < nnodes; ++v)
            if (orc_expm(n, Q + (size_t)node_q[v] * nn, t[v], esd + (size_t)v * nn, work, NULL)) rc = -2;
        if (rc == 0)
            rc = orc_batch_loglik(nnodes, n, idx, ptr, esd, nobs, obs_nodes,
                                  obs_dense + (size_t)i * nobs * n, 1, root_w, loglik + i,
                                  status + i);
    }
    free(esd); free(work);
    return rc;
}
