// Micro-benchmark: two ways to overwrite one of N f64 registers chosen by a per-lane index, at 1 wave/SIMD.
//   (A) v_cmp_eq + 2 x v_cndmask per register (what the compiler emits for key[i] = (a==i) ? k : key[i])
//   (B) v_cmpx_eq (writes EXEC) + v_mov_b64 + s_mov_b64 exec restore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int N = 11;

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, int salt) {
    double key[N];
    for (int i = 0; i < N; ++i) key[i] = i + threadIdx.x;
    int a = (threadIdx.x * 7 + salt) % N;
    double v = salt;
    unsigned long long full = __builtin_amdgcn_read_exec();
    for (int it = 0; it < iters; ++it) {
        a = a + 5; a = a >= N ? a - N : a;        // cheap per-lane varying index
        v += 1.0;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) key[i] = (a == i) ? v : key[i];
        } else if (MODE == 2) {
            // pure-VALU blend: one-hot of the index, per key a sign-extended 1-bit field (0 / -1), two v_bfi_b32
            const int onehot = 1 << a;
            const int vlo = __double2loint(v), vhi = __double2hiint(v);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int m = __builtin_amdgcn_sbfe(onehot, i, 1);
                const int lo = (vlo & m) | (__double2loint(key[i]) & ~m);
                const int hi = (vhi & m) | (__double2hiint(key[i]) & ~m);
                key[i] = __hiloint2double(hi, lo);
            }
        } else {
#define SEL(I) asm volatile("v_cmpx_eq_u32_e32 " #I ", %1\n\tv_mov_b64 %0, %2\n\ts_mov_b64 exec, %3" \
                            : "+v"(key[I]) : "v"(a), "v"(v), "s"(full) : "vcc");
            SEL(0) SEL(1) SEL(2) SEL(3) SEL(4) SEL(5) SEL(6) SEL(7) SEL(8) SEL(9) SEL(10)
#undef SEL
        }
    }
    double s = 0;
    for (int i = 0; i < N; ++i) s += key[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
float run(double* d, int iters, int blocks = 1024) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(1024), dim3(64), 0, 0, d, 10, 1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* d;
    (void)hipMalloc(&d, 8192 * 64 * 8);
    double* h = (double*)malloc(1024 * 64 * 8);
    const int iters = 20000;
    float a = run<0>(d, iters);
    (void)hipMemcpy(h, d, 1024 * 64 * 8, hipMemcpyDeviceToHost);
    double ca = 0; for (int i = 0; i < 1024 * 64; ++i) ca += h[i];
    float c = run<2>(d, iters);
    (void)hipMemcpy(h, d, 1024 * 64 * 8, hipMemcpyDeviceToHost);
    double cc = 0; for (int i = 0; i < 1024 * 64; ++i) cc += h[i];
    printf("sbfe+bfi select          : %.1f ns/iter (%.0f cycles@2.4GHz)  checksum %.6e\n", c * 1e6 / iters, c * 1e6 / iters * 2.4, cc);
    float b = run<1>(d, iters);
    (void)hipMemcpy(h, d, 1024 * 64 * 8, hipMemcpyDeviceToHost);
    double cb = 0; for (int i = 0; i < 1024 * 64; ++i) cb += h[i];
    printf("cndmask select of %d keys: %.1f ns/iter (%.0f cycles@2.4GHz)  checksum %.6e\n", N, a * 1e6 / iters, a * 1e6 / iters * 2.4, ca);
    printf("cmpx+mov_b64 select      : %.1f ns/iter (%.0f cycles@2.4GHz)  checksum %.6e\n", b * 1e6 / iters, b * 1e6 / iters * 2.4, cb);
    for (int blocks : {1024, 2048, 4096, 8192})
        printf("cndmask select, %d waves (%d per SIMD): %.1f ns/iter/wave-slot\n", blocks, blocks / 1024, run<0>(d, iters, blocks) * 1e6 / iters);
    return 0;
}
