// Fixed-order reduction of the per-wave partial sums of a batch -> totals[3], as a
// device function so that the stand-alone kernel (prune.hip) and the extra workgroup
// an expm launch may carry (expm.hip) run the same arithmetic: the totals are bitwise
// the same whichever of the two computed them.  256 threads.
#pragma once

__device__ __forceinline__ void rt_reduce_partials_body(const double *__restrict__ partial,
                                                        long npartials,
                                                        double *__restrict__ totals,
                                                        double nsites)
{
    __shared__ double ssum[256];
    __shared__ double szero[256];
    // thread t adds partials t, t + 256, ... in that order (the order fixes the
    // rounding: totals are bitwise reproducible); the loads of eight of them are
    // issued together so that the pass costs one L2 round trip per 2 048 partials
    double s = 0.0, z = 0.0;
    const double2 *p2 = (const double2 *)partial;
    for (long base = threadIdx.x; base < npartials; base += 256 * 8) {
        double2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long i = base + 256 * j;
            v[j] = i < npartials ? p2[i] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (base + 256 * j < npartials) {
                s += v[j].x;
                z += v[j].y;
            }
        }
    }
    ssum[threadIdx.x] = s;
    szero[threadIdx.x] = z;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            ssum[threadIdx.x] += ssum[threadIdx.x + w];
            szero[threadIdx.x] += szero[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = ssum[0];
        totals[1] = szero[0];
        totals[2] = nsites;
    }
}
