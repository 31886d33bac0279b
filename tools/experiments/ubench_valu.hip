// Micro-benchmark: issue cost of the VALU instructions the trace kernel is made of, at 1 wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int CHAINS>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed) {
    double a[CHAINS];
    float f[CHAINS];
    int sel = threadIdx.x & 7;
    for (int c = 0; c < CHAINS; ++c) { a[c] = seed + c + threadIdx.x; f[c] = (float)a[c]; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == 0) a[c] = fma(a[c], 1.0000001, 0.5);            // v_fma_f64
                if (MODE == 1) a[c] = a[c] * 1.0000001;                      // v_mul_f64
                if (MODE == 2) a[c] = fmax(a[c], seed + r);                  // v_max_f64
                if (MODE == 3) f[c] = fmaf(f[c], 1.0000001f, 0.5f);          // v_fma_f32
                if (MODE == 4) a[c] = (sel == r) ? seed : a[c];              // cmp + 2 cndmask
                if (MODE == 5) a[c] = a[c] + 0.25;                           // v_add_f64
                if (MODE == 6) f[c] = __frsqrt_rn(f[c]) + 1.0f;              // v_rsq_f32 + add
            }
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c] + f[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE, int CHAINS>
void run(const char* name, double* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(1024), dim3(64), 0, 0, d, 10, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(1024), dim3(64), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16 * CHAINS;
    printf("%-28s chains=%d  %.2f ns/op/wave  (%.1f cycles @2.4GHz)\n", name, CHAINS, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}

int main() {
    double* d;
    hipMalloc(&d, 1024 * 64 * 8);
    run<0, 1>("fma_f64 dependent", d); run<0, 4>("fma_f64", d); run<0, 8>("fma_f64", d);
    run<1, 1>("mul_f64 dependent", d); run<1, 4>("mul_f64", d);
    run<5, 1>("add_f64 dependent", d); run<5, 4>("add_f64", d);
    run<2, 1>("max_f64 dependent", d); run<2, 4>("max_f64", d);
    run<3, 1>("fma_f32 dependent", d); run<3, 4>("fma_f32", d); run<3, 8>("fma_f32", d);
    run<4, 1>("select64 (cmp+2cndmask)", d); run<4, 4>("select64 (cmp+2cndmask)", d);
    run<6, 1>("rsq_f32+add dependent", d); run<6, 4>("rsq_f32+add", d);
    return 0;
}
