// Layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950 (the local guides give the
// 16x16x4 f64 maps only).  D = A * B per block, 4 blocks.  For every pair (la, lb) one
// wave sets A = [lane == la], B = [lane == lb], C = 0 and records which lanes of D
// become 1: that pins which (block, i, k) an A lane holds, which (block, k, j) a B lane
// holds and which (block, i, j) a D lane holds.  Repeated with cbsz = 2 and abid = 0..3
// (A broadcast from one block to all four).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma4_layout mfma4_layout.hip
// Output: a summary of the inferred maps + a self-check of a 4 x (4 x 16) product
// computed through the inferred maps against a scalar reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CBSZ, int ABID>
__global__ void probe(unsigned long long *out)
{
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    const int lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0;
    const double b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    const unsigned long long m = __ballot(d != 0.0);
    if (lane == 0) out[blockIdx.x] = m;
}

// product through the inferred layout: t[r][s] = sum_k P[r][k] x[k][s], 4 rows, 4 k, 16 sites
__global__ void product_check(const double *P, const double *x, double *t, int cbsz_mode)
{
    const int lane = threadIdx.x;
    const int b = lane >> 4, k = (lane >> 2) & 3, i = lane & 3, j = lane & 3;
    // A lane (b, k, i): P[i][k] (same in every block, or block 0 only with broadcast)
    const double a = (cbsz_mode && b != 0) ? 1e300 : P[i * 4 + k];
    // B lane (b, k, j): x[k][4b + j]
    const double bb = x[k * 16 + 4 * b + j];
    double d;
    if (cbsz_mode) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, 2, 0, 0);
    else d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, 0, 0, 0);
    // D lane (b, i, j) under the hypothesis lane = 16 b + 4 i + j
    const int di = (lane >> 2) & 3, dj = lane & 3;
    t[di * 16 + 4 * b + dj] = d;
}

static void decode(const char *title, const std::vector<unsigned long long> &m)
{
    printf("== %s\n", title);
    // for each A lane: the set of B lanes it interacts with and the D lanes hit
    int ok_rows = 0;
    for (int la = 0; la < 64; ++la) {
        int nb = 0;
        for (int lb = 0; lb < 64; ++lb) nb += m[la * 64 + lb] != 0;
        ok_rows += nb > 0;
    }
    printf("A lanes that interact with some B lane: %d of 64\n", ok_rows);
    for (int la = 0; la < 64; la += 1) {
        if (la >= 20 && la < 60) continue;
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (m[la * 64 + lb]) {
                printf(" B%d->D[", lb);
                bool first = true;
                for (int d = 0; d < 64; ++d)
                    if (m[la * 64 + lb] >> d & 1ull) { printf(first ? "%d" : ",%d", d); first = false; }
                printf("]");
            }
        printf("\n");
    }
}

int main()
{
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, 4096 * 8));
    std::vector<unsigned long long> m(4096);
    probe<0, 0><<<4096, 64>>>(d_out);
    CK(hipMemcpy(m.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost));
    decode("cbsz=0 abid=0", m);
    // hypothesis check: A lane 16b + 4k + i, B lane 16b' + 4k' + j interact iff b == b' and
    // k == k', hitting D lane 16b + 4i + j
    int bad = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int b = la >> 4, k = (la >> 2) & 3, i = la & 3;
            const int b2 = lb >> 4, k2 = (lb >> 2) & 3, j = lb & 3;
            unsigned long long want = (b == b2 && k == k2) ? 1ull << (16 * b + 4 * i + j) : 0ull;
            bad += m[la * 64 + lb] != want;
        }
    printf("hypothesis A=16b+4k+i, B=16b+4k+j, D=16b+4i+j: %d mismatches\n", bad);
    // alternative: A lane 16b + 4i + k
    bad = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int b = la >> 4, i = (la >> 2) & 3, k = la & 3;
            const int b2 = lb >> 4, k2 = (lb >> 2) & 3, j = lb & 3;
            unsigned long long want = (b == b2 && k == k2) ? 1ull << (16 * b + 4 * i + j) : 0ull;
            bad += m[la * 64 + lb] != want;
        }
    printf("hypothesis A=16b+4i+k, B=16b+4k+j, D=16b+4i+j: %d mismatches\n", bad);
    probe<2, 0><<<4096, 64>>>(d_out);
    CK(hipMemcpy(m.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost));
    decode("cbsz=2 abid=0", m);
    probe<2, 1><<<4096, 64>>>(d_out);
    CK(hipMemcpy(m.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost));
    decode("cbsz=2 abid=1", m);
    probe<2, 3><<<4096, 64>>>(d_out);
    CK(hipMemcpy(m.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost));
    decode("cbsz=2 abid=3", m);
    probe<1, 0><<<4096, 64>>>(d_out);
    CK(hipMemcpy(m.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost));
    decode("cbsz=1 abid=0", m);

    // numeric check of a product through the first hypothesis, with and without broadcast
    double hP[16], hx[64], ht[64], *dP, *dx, *dt;
    for (int e = 0; e < 16; ++e) hP[e] = 0.25 + 0.1 * e;
    for (int e = 0; e < 64; ++e) hx[e] = 1.0 + 0.01 * e * e;
    CK(hipMalloc(&dP, sizeof hP)); CK(hipMalloc(&dx, sizeof hx)); CK(hipMalloc(&dt, sizeof ht));
    CK(hipMemcpy(dP, hP, sizeof hP, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; ++mode) {
        product_check<<<1, 64>>>(dP, dx, dt, mode);
        CK(hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < 16; ++s) {
                double w = 0.0;
                for (int k = 0; k < 4; ++k) w += hP[r * 4 + k] * hx[k * 16 + s];
                const double e = ht[r * 16 + s] - w;
                worst = e < 0 ? (-e > worst ? -e : worst) : (e > worst ? e : worst);
            }
        printf("product check (%s): max |err| = %.3g\n", mode ? "A broadcast from block 0, cbsz=2" : "A replicated", worst);
    }
    return 0;
}
