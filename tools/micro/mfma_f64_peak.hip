// Microbenchmark: measured peak of the f64 matrix pipe of an MI355X (gfx950).
//
// SURVEY.md section 8(d) asks for the measured rate of v_mfma_f64_16x16x4_f64 to be
// recorded (the local micro-architecture guide has no f64 MFMA row; the datasheet
// says 78.6 TFLOP/s).  Every wave runs C independent accumulation chains of the
// instruction in a loop, W waves per SIMD, one workgroup of 4 * W waves per CU slot,
// enough workgroups to fill every CU.  Also times v_mfma_f64_4x4x4_4b_f64 (the
// 4-block form) and a plain v_fma_f64 loop for comparison.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
// Prints one JSON object (profiles/r02_mfma_f64_peak.json is a copy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ void __launch_bounds__(256) mfma16_kernel(double *out, int iters, double a0, double b0)
{
    d4 acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C>
__global__ void __launch_bounds__(256) mfma4_kernel(double *out, int iters, double a0, double b0)
{
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0;
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += acc[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C>
__global__ void __launch_bounds__(256) fma_kernel(double *out, int iters, double a0, double b0)
{
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = c;
    const double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fma(acc[c], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += acc[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double time_us(K kern, int grid, double *out, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms * 1e3 < best) best = ms * 1e3;
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8));
    const int iters = 4000;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"results\": [\n", prop.gcnArchName,
           cus, prop.clockRate / 1000);
    bool first = true;
    double best16 = 0.0;
#define REPORT(name, chains, wps, flops_per_inst, us)                                         \
    do {                                                                                      \
        const double insts = (double)cus * 4 * (wps) * iters * 8.0 * (chains);                \
        const double tf = insts * (flops_per_inst) / ((us) * 1e-6) / 1e12;                    \
        printf("%s  {\"inst\": \"%s\", \"chains_per_wave\": %d, \"waves_per_simd\": %d, "    \
               "\"us\": %.1f, \"tflops\": %.2f, \"cycles_per_inst_per_simd\": %.2f}",         \
               first ? "" : ",\n", name, chains, wps, us, tf,                                 \
               (us) * 1e-6 * prop.clockRate * 1e3 / (iters * 8.0 * (chains) * (wps)));        \
        first = false;                                                                        \
        if (flops_per_inst == 2048.0 && tf > best16) best16 = tf;                             \
    } while (0)
    // 16x16x4: 2 * 16 * 16 * 4 = 2048 flops per instruction
    { double us = time_us(mfma16_kernel<1>, cus * 1, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 1, 1, 2048.0, us); }
    { double us = time_us(mfma16_kernel<2>, cus * 1, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 2, 1, 2048.0, us); }
    { double us = time_us(mfma16_kernel<4>, cus * 1, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 4, 1, 2048.0, us); }
    { double us = time_us(mfma16_kernel<1>, cus * 2, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 1, 2, 2048.0, us); }
    { double us = time_us(mfma16_kernel<2>, cus * 2, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 2, 2, 2048.0, us); }
    { double us = time_us(mfma16_kernel<4>, cus * 2, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 4, 2, 2048.0, us); }
    { double us = time_us(mfma16_kernel<1>, cus * 4, out, iters); REPORT("v_mfma_f64_16x16x4_f64", 1, 4, 2048.0, us); }
    // 4x4x4, 4 blocks: 4 * 2 * 4 * 4 * 4 = 512 flops per instruction
    { double us = time_us(mfma4_kernel<1>, cus * 1, out, iters); REPORT("v_mfma_f64_4x4x4_4b_f64", 1, 1, 512.0, us); }
    { double us = time_us(mfma4_kernel<4>, cus * 1, out, iters); REPORT("v_mfma_f64_4x4x4_4b_f64", 4, 1, 512.0, us); }
    { double us = time_us(mfma4_kernel<4>, cus * 2, out, iters); REPORT("v_mfma_f64_4x4x4_4b_f64", 4, 2, 512.0, us); }
    // vector FMA: 64 lanes * 2 flops
    { double us = time_us(fma_kernel<8>, cus * 1, out, iters); REPORT("v_fma_f64", 8, 1, 128.0, us); }
    { double us = time_us(fma_kernel<8>, cus * 2, out, iters); REPORT("v_fma_f64", 8, 2, 128.0, us); }
    printf("\n], \"f64_mfma_16x16x4_peak_tflops\": %.2f}\n", best16);
    hipFree(out);
    return 0;
}
