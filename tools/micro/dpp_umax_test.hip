// Unit check of the DPP wave-wide unsigned max used for the pivot search of the
// expm kernel (row_shr 1/2/4/8 + row_bcast 15/31 scan, result in lane 63) against
// a ds_bpermute (shuffle) reduction, on random data.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned wave_umax_dpp(unsigned v)
{
#define RT_DPP(ctrl, rmask) (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
    unsigned t;
    t = RT_DPP(0x111, 0xf); v = v > t ? v : t;   // row_shr:1
    t = RT_DPP(0x112, 0xf); v = v > t ? v : t;   // row_shr:2
    t = RT_DPP(0x114, 0xf); v = v > t ? v : t;   // row_shr:4
    t = RT_DPP(0x118, 0xf); v = v > t ? v : t;   // row_shr:8
    t = RT_DPP(0x142, 0xa); v = v > t ? v : t;   // row_bcast:15
    t = RT_DPP(0x143, 0xc); v = v > t ? v : t;   // row_bcast:31
#undef RT_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ void k(const unsigned *in, unsigned *out_dpp, unsigned *out_shfl)
{
    unsigned v = in[blockIdx.x * 64 + threadIdx.x];
    unsigned a = wave_umax_dpp(v);
    unsigned b = v;
    for (int o = 32; o > 0; o >>= 1) { unsigned t = __shfl_xor(b, o, 64); b = b > t ? b : t; }
    if (threadIdx.x == 0) { out_dpp[blockIdx.x] = a; out_shfl[blockIdx.x] = b; }
}

int main()
{
    const int nb = 4096;
    std::vector<unsigned> h(nb * 64);
    srand(1);
    for (auto &x : h) x = ((unsigned)rand() << 8) ^ (unsigned)rand();
    for (int i = 0; i < 64; ++i) h[i] = (i == 17) ? 0xffffffffu : 0;      // single winner
    for (int i = 0; i < 64; ++i) h[64 + i] = (i == 63) ? 5 : 0;
    for (int i = 0; i < 64; ++i) h[128 + i] = (i == 0) ? 5 : 0;
    unsigned *d, *a, *b;
    hipMalloc(&d, h.size() * 4); hipMalloc(&a, nb * 4); hipMalloc(&b, nb * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<nb, 64>>>(d, a, b);
    std::vector<unsigned> ha(nb), hb(nb);
    hipMemcpy(ha.data(), a, nb * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, nb * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < nb; ++i) {
        unsigned want = 0;
        for (int j = 0; j < 64; ++j) want = h[i * 64 + j] > want ? h[i * 64 + j] : want;
        if (ha[i] != want || hb[i] != want) { if (bad < 5) printf("block %d: dpp %u shfl %u want %u\n", i, ha[i], hb[i], want); ++bad; }
    }
    printf("dpp_umax_test: %d mismatches of %d\n", bad, nb);
    return bad != 0;
}
