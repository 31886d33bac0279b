// Do the lanes of ONE ds_add_rtn_u32 that hit the same LDS address get their pre-add values in ascending lane order?
// (Unspecified by the ISA; the ingest kernels' fast ranking path relies on it after checking it at start-up.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(const unsigned* __restrict__ addr, unsigned* __restrict__ out, int groups) {
    __shared__ unsigned cnt[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) cnt[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < groups; ++g) {
        const unsigned a = addr[(blockIdx.x * groups + g) * 64 + lane];
        out[(blockIdx.x * groups + g) * 64 + lane] = atomicAdd(&cnt[a], 1u);
    }
}
int main() {
    const int blocks = 2048, groups = 64, n = blocks * groups * 64;
    unsigned* h = (unsigned*)malloc(n * 4), *o = (unsigned*)malloc(n * 4);
    srand(1);
    for (int b = 0; b < blocks; ++b)
        for (int g = 0; g < groups; ++g) {
            const int kind = (b + g) % 5;      // all equal / two values / 16 values / random 256 / runs
            for (int l = 0; l < 64; ++l) {
                unsigned a;
                if (kind == 0) a = 7; else if (kind == 1) a = (rand() & 1) * 33; else if (kind == 2) a = rand() & 15;
                else if (kind == 3) a = rand() & 255; else a = (l / 5) & 255;
                h[(b * groups + g) * 64 + l] = a;
            }
        }
    unsigned *da, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(da, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, da, dout, groups);
    hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned cnt[256] = {0};
        for (int g = 0; g < groups; ++g)
            for (int l = 0; l < 64; ++l) {
                const int i = (b * groups + g) * 64 + l;
                if (o[i] != cnt[h[i]]) { if (bad < 5) printf("block %d group %d lane %d addr %u got %u want %u\n", b, g, l, h[i], o[i], cnt[h[i]]); ++bad; }
                cnt[h[i]]++;
            }
    }
    printf("%ld of %d pre-add values differ from ascending-lane order\n", bad, n);
    return bad != 0;
}
