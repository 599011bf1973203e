// Microbenchmark: HBM read ceiling for the lane kernel's access pattern
// (one wave = 64 sites, K slots of 2 x 1 KiB fully coalesced pieces each),
// for several grid shapes and unroll depths.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ inline double2 ld_nt(const double2 *p)
{
    const d2v v = __builtin_nontemporal_load((const d2v *)p);
    return make_double2(v.x, v.y);
}

template <int R, int WPB, bool NT = false>
__global__ void __launch_bounds__(WPB * 64) stream_kernel(const double2 *__restrict__ obs, int K, double *__restrict__ out, long nblocks)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long gw = (long)blockIdx.x * WPB + wave;
    if (gw >= nblocks) return;
    const double2 *g = obs + (size_t)gw * K * 2 * 64 + lane;
    double2 ring[R][2];
#pragma unroll
#define LD(p) (NT ? ld_nt(&(p)) : (p))
    for (int j = 0; j < R; ++j) { ring[j][0] = LD(g[(j * 2) * 64]); ring[j][1] = LD(g[(j * 2 + 1) * 64]); }
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int k = k0 + j;
            acc += ring[j][0].x * ring[j][1].y + ring[j][0].y * ring[j][1].x;
            if (k + R < K) { ring[j][0] = LD(g[((k + R) * 2) * 64]); ring[j][1] = LD(g[((k + R) * 2 + 1) * 64]); }
        }
    }
    out[gw * 64 + lane] = acc;
}

template <int R, int WPB, bool NT = false>
float run(const double2 *d, int K, double *out, long nblocks, int nbuf, size_t stride, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid((unsigned)((nblocks + WPB - 1) / WPB));
    for (int i = 0; i < 3; ++i) stream_kernel<R, WPB, NT><<<grid, WPB * 64>>>(d + (i % nbuf) * stride, K, out, nblocks);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) stream_kernel<R, WPB, NT><<<grid, WPB * 64>>>(d + (i % nbuf) * stride, K, out, nblocks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3f;
}

int main()
{
    const int K = 64; const long nblocks = 1563; const int nbuf = 4;
    const size_t per = (size_t)nblocks * K * 2 * 64;     // double2 per buffer
    double2 *d; double *out;
    CK(hipMalloc(&d, per * nbuf * sizeof(double2)));
    CK(hipMalloc(&out, nblocks * 64 * 8));
    CK(hipMemset(d, 0, per * nbuf * sizeof(double2)));
    const double mb = per * 16.0 / 1e6;
    printf("bytes per launch %.1f MB\n", mb);
#define RUN(R, W) { float us = run<R, W>(d, K, out, nblocks, nbuf, per, 50); printf("R=%2d waves/block=%d  %.1f us  %.2f TB/s\n", R, W, us, mb / us); }
    RUN(2, 1) RUN(4, 1) RUN(8, 1) RUN(16, 1) RUN(4, 4) RUN(8, 4) RUN(16, 4) RUN(8, 2)
#define RUNNT(R, W) { float us = run<R, W, true>(d, K, out, nblocks, nbuf, per, 50); printf("R=%2d waves/block=%d non-temporal  %.1f us  %.2f TB/s\n", R, W, us, mb / us); }
    RUNNT(4, 1) RUNNT(8, 1) RUNNT(8, 4)
    // bigger problem: 4x the sites (the ceiling when the chip is full)
    hipFree(d); hipFree(out);
    const long nb2 = 1563 * 4;
    const size_t per2 = (size_t)nb2 * K * 2 * 64;
    CK(hipMalloc(&d, per2 * 2 * sizeof(double2)));
    CK(hipMalloc(&out, nb2 * 64 * 8));
    CK(hipMemset(d, 0, per2 * 2 * sizeof(double2)));
    const double mb2 = per2 * 16.0 / 1e6;
    { float us = run<8, 4>(d, K, out, nb2, 2, per2, 20); printf("4x sites R=8 W=4: %.1f us %.2f TB/s\n", us, mb2 / us); }
    { float us = run<4, 1>(d, K, out, nb2, 2, per2, 20); printf("4x sites R=4 W=1: %.1f us %.2f TB/s\n", us, mb2 / us); }
    return 0;
}
