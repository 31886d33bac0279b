// Philox-4x32-10 counter RNG (Salmon et al., SC'11) + the Box-Muller pieces shared by the sampler kernels.
#pragma once
#include "common.h"

namespace dcarl {

struct U4 { uint32_t x0, x1, x2, x3; };

// a ^ b ^ k as ONE v_bitop3_b32 (truth table 0x96 = three-input xor); k is the round key: uniform, read from an SGPR.
// The compiler emits two v_xor_b32 for it: 40 of the ~100 full-rate instructions of a Philox block.
__device__ __forceinline__ uint32_t xor3_key(uint32_t a, uint32_t b, uint32_t k) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per constant (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the
        // multiplies are the quarter-rate instructions this kernel is bound by
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = xor3_key(hi1, c1, k0); c1 = lo1; c2 = xor3_key(hi0, c3, k1); c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// u = (x + 0.5) * 2^-32 in (0,1];  Box-Muller radius, and the angle in TURNS for v_cos_f32 / v_sin_f32.
__device__ __forceinline__ float unit_open(uint32_t x) {
    return fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
}
__device__ __forceinline__ float bm_radius(uint32_t x1) {
    // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u).  v_log_f32 is a 1-ulp instruction already, so the square root is the raw
    // v_sqrt_f32 too (the IEEE expansion costs 8 more instructions per draw for a bit the logarithm has lost); the
    // argument is 0 or a normal number.
    // Near u = 1 the f32 u is the problem, not the logarithm: u is on a 2^-24 grid there (the top 2^7 words round to 1.0f and gave
    // z = 0), so -ln u came out in steps of 6e-8 and the radius in steps of 3.4e-4 — |dz| up to 1.3e-4 against float64 on one draw in
    // 10^6 (profiles/r06_sampler_normal_error.txt).  The top 2^-10 of the words take t = 1 - u instead, which IS exact in f32 (the
    // complement of the word), and -ln(1 - t) = t + t^2/2 + t^3/3 (next term < 2^-32 relative): three instructions and a select.
    const float t = fmaf((float)~x1, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float near1 = 2.0f * t * fmaf(t, fmaf(t, 0.33333334f, 0.5f), 1.0f);
    const float far = -1.3862943611198906f * __log2f(unit_open(x1));
    return __builtin_amdgcn_sqrtf(x1 >= 0xffc00000u ? near1 : far);
}

// x / 6 correctly rounded in three instructions instead of the ten of the IEEE division expansion: q = RN(x * RN(1/6)),
// the exact residual r = x - 6q (one fma), q' = RN(q + r * RN(1/6)) (Markstein's final correction).  Checked against
// x / 6.0f for every float of magnitude 1e-30 .. 16 by tools/div6_check.c (0 differences in 1.74e9 values).
__device__ __forceinline__ float div6(float x) {
    constexpr float y = 1.0f / 6.0f;
    const float q = x * y;
    return fmaf(fmaf(-6.0f, q, x), y, q);
}

}  // namespace dcarl
