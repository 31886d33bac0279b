// How many 64-thread blocks with X bytes of LDS are co-resident on one MI355X CU?  2048 blocks that each spin for a
// fixed time finish in one round iff 8 fit per CU (256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (lds[threadIdx.x] == -1) *sink = 1;
}
int main() {
    int* sink; (void)hipMalloc(&sink, 4);
    for (int bytes = 18432; bytes <= 21504; bytes += 256) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(spin, dim3(2048), dim3(64), bytes, 0, 1000LL, sink);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(spin, dim3(2048), dim3(64), bytes, 0, 10000000LL, sink);   // 100 ms at 100 MHz wall clock
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("LDS %5d B/block: %.1f ms\n", bytes, ms);
    }
    return 0;
}
