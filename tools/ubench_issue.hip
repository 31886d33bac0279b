// Micro-benchmark: raw issue cost (cycles per wave-instruction, 1 wave per SIMD) of the instruction kinds the
// trace kernel's select / arg-max code is built from.  Each body is 16 independent instructions, repeated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// 16 instructions per body, cycling over 8 independent destination registers (no RAW/WAW chains shorter than 8)
#define I8(pre, post) pre "%0" post pre "%1" post pre "%2" post pre "%3" post pre "%4" post pre "%5" post pre "%6" post pre "%7" post
#define BODY16(pre, post) I8(pre, post) I8(pre, post)
#define OUT8F "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
#define OUT8D "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3), "+v"(z4), "+v"(z5), "+v"(z6), "+v"(z7)

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    float a = threadIdx.x, b = 1.5f;
    float d0 = 0, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
    double x = threadIdx.x, y = 1.25;
    double z0 = 0, z1 = 1, z2 = 2, z3 = 3, z4 = 4, z5 = 5, z6 = 6, z7 = 7;
    unsigned long long m = 0x5555555555555555ull;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(BODY16("v_cndmask_b32_e64 ", ", %8, %9, %10\n\t") : OUT8F : "v"(a), "v"(b), "s"(m));
        if (MODE == 1) asm volatile(BODY16("v_cndmask_b32_e32 ", ", %8, %9, vcc\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 2) asm volatile(BODY16("v_cmp_eq_u32_e64 s[20:21], %8, ", "\n\t") : OUT8F : "v"(a) : "s20", "s21");
        if (MODE == 3) asm volatile(BODY16("v_bfi_b32 ", ", %8, %9, %9\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 4) asm volatile(BODY16("v_and_or_b32 ", ", %8, %9, %9\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 5) asm volatile(BODY16("v_mov_b64 ", ", %8\n\t") : OUT8D : "v"(x));
        if (MODE == 6) asm volatile(BODY16("v_max_f64 ", ", %8, %9\n\t") : OUT8D : "v"(x), "v"(y));
        if (MODE == 7) asm volatile(BODY16("v_add_u32_e32 ", ", %8, %9\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 8) asm volatile(BODY16("v_fma_f64 ", ", %8, %9, %8\n\t") : OUT8D : "v"(x), "v"(y));
        if (MODE == 9) asm volatile(BODY16("v_max3_f32 ", ", %8, %9, %8\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 10) asm volatile(BODY16("v_cmp_eq_u32_e32 vcc, %8, ", "\n\t") : OUT8F : "v"(a) : "vcc");
        if (MODE == 11) asm volatile(BODY16("v_mov_b32_e32 ", ", %8\n\t") : OUT8F : "v"(a));
        if (MODE == 12) asm volatile(BODY16("v_xor_b32_e32 ", ", %8, %9\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 13) asm volatile(BODY16("v_bfe_i32 ", ", %8, 3, 1\n\t") : OUT8F : "v"(a));
        if (MODE == 14) asm volatile(BODY16("s_mov_b64 s[20:21], %10 ; ", "\n\t") : OUT8F : "v"(a), "v"(b), "s"(m) : "s20", "s21");
        if (MODE == 15) asm volatile(BODY16("v_cmp_gt_f64_e64 s[20:21], %8, ", "\n\t") : OUT8D : "v"(x) : "s20", "s21");
        if (MODE == 16) asm volatile(BODY16("v_min_u32_e32 ", ", %8, %9\n\t") : OUT8F : "v"(a), "v"(b));
        if (MODE == 17) asm volatile(BODY16("v_mul_f64 ", ", %8, %9\n\t") : OUT8D : "v"(x), "v"(y));
        if (MODE == 18) asm volatile(BODY16("v_cvt_f64_f32_e32 ", ", %8\n\t") : OUT8D : "v"(a));
        if (MODE == 19) asm volatile(BODY16("v_rsq_f32_e32 ", ", %8\n\t") : OUT8F : "v"(a));
        // round 6: compare + select PAIRS as the compiler emits them (vcc) and with the predicate in an SGPR pair, and a vcc select between
        // two f64 operations (is the e32 form's cost its own, or an artefact of sixteen of them in a row?)
        if (MODE == 21) asm volatile(I8("v_cmp_eq_u32_e32 vcc, %8, %9\n\tv_cndmask_b32_e32 ", ", %8, %9, vcc\n\t") : OUT8F : "v"(a), "v"(b) : "vcc");
        if (MODE == 22) asm volatile(I8("v_cmp_eq_u32_e64 s[20:21], %8, %9\n\tv_cndmask_b32_e64 ", ", %8, %9, s[20:21]\n\t") : OUT8F : "v"(a), "v"(b) : "s20", "s21");
        if (MODE == 23) asm volatile(I8("v_fma_f64 %10, %11, %12, %11\n\tv_cndmask_b32_e32 ", ", %8, %9, vcc\n\t") : OUT8F : "v"(a), "v"(b), "v"(z0), "v"(x), "v"(y));
        if (MODE == 24) asm volatile(I8("v_fma_f64 %10, %11, %12, %11\n\tv_cndmask_b32_e64 ", ", %8, %9, %13\n\t") : OUT8F : "v"(a), "v"(b), "v"(z0), "v"(x), "v"(y), "s"(m));
        if (MODE == 20) asm volatile(BODY16("v_cndmask_b32_dpp ", ", %8, %9, vcc quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n\t") : OUT8F : "v"(a), "v"(b));
    }
    out[blockIdx.x * 64 + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (float)(z0 + z1 + z2 + z3 + z4 + z5 + z6 + z7);
}

static int g_waves = 1;      // wavefronts per SIMD (argv[1]): 1 = single-wave issue cadence, 4 (f32) / 3 (f64) = what the online kernel runs with
template <int MODE>
void run(const char* name, float* d) {
    const int iters = 4000;
    const int grid = 1024 * g_waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: g_waves waves each issue iters*16 instructions in ms -> SIMD time per wave-instruction
    printf("%-34s %.2f ns per wave-instruction and SIMD  (%.2f cycles @2.4GHz; %d waves per SIMD)\n", name,
           ms * 1e6 / (iters * 16.0 * g_waves), ms * 1e6 / (iters * 16.0 * g_waves) * 2.4, g_waves);
}

int main(int argc, char** argv) {
    if (argc > 1) g_waves = atoi(argv[1]) > 0 ? atoi(argv[1]) : 1;
    float* d;
    (void)hipMalloc(&d, 1024 * 8 * 64 * 4);
    run<0>("v_cndmask_b32_e64 (sgpr mask)", d);
    run<1>("v_cndmask_b32_e32 (vcc)", d);
    run<2>("v_cmp_eq_u32_e64 -> sgpr", d);
    run<10>("v_cmp_eq_u32_e32 -> vcc", d);
    run<3>("v_bfi_b32", d);
    run<4>("v_and_or_b32", d);
    run<5>("v_mov_b64", d);
    run<11>("v_mov_b32", d);
    run<6>("v_max_f64", d);
    run<8>("v_fma_f64", d);
    run<7>("v_add_u32_e32", d);
    run<12>("v_xor_b32_e32", d);
    run<16>("v_min_u32_e32", d);
    run<9>("v_max3_f32", d);
    run<13>("v_bfe_i32", d);
    run<14>("s_mov_b64", d);
    run<15>("v_cmp_gt_f64_e64", d);
    run<17>("v_mul_f64", d);
    run<18>("v_cvt_f64_f32", d);
    run<19>("v_rsq_f32", d);
    run<21>("pair: v_cmp_e32 vcc + v_cndmask_e32 (per PAIR)", d);
    run<22>("pair: v_cmp_e64 sgpr + v_cndmask_e64 (per PAIR)", d);
    run<23>("pair: v_fma_f64 + v_cndmask_e32 vcc (per PAIR)", d);
    run<24>("pair: v_fma_f64 + v_cndmask_e64 sgpr (per PAIR)", d);
    return 0;
}
