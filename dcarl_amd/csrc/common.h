// Shared device helpers for the DCARL gfx950 kernels.  CDNA4 only: wave64, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dcarl.h"

namespace dcarl {

constexpr int WAVE = 64;
constexpr int CODE_BITS = 5;                 // tie-break code in the 5 low mantissa bits (A <= 32)
constexpr long long CODE_MASK = (1LL << CODE_BITS) - 1;

// Kernel-side copy of dcarl_params_t with the derived Hoeffding constant.
struct DevParams {
    int rule_act;
    int n_thres;
    double hoeff;       // scale*sqrt(log(1/alpha)/2): scale*sqrt(log(1/alpha)/2/n) == hoeff/sqrt(n)  (S1:12,16,24)
    double cap;
    double init_rule;
    double init_other;
};

// ---- arg-max with the reference's first-max tie rule (np.argmax / max, S1:93-94) in ONE v_max_f64 per
// candidate: the 5 low mantissa bits of V are replaced by a code that orders equal values by action id
// (lower id wins).  For v >= 0 a larger mantissa is a larger value -> code = 31 - a; for v < 0 a larger
// mantissa is a smaller value -> code = a.  The perturbation is <= 31 ulp (7e-15 relative).
__device__ __forceinline__ double encode_key(double v, int a) {
    const int hi = __double2hiint(v);
    const int flip = ~(hi >> 31) & (int)CODE_MASK;        // 31 for v >= 0, 0 for v < 0
    const int lo = (__double2loint(v) & ~(int)CODE_MASK) | (a ^ flip);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int decode_action(double key) {
    const int flip = ~(__double2hiint(key) >> 31) & (int)CODE_MASK;
    return (__double2loint(key) & (int)CODE_MASK) ^ flip;
}
__device__ __forceinline__ double strip_code(double key) {
    return __hiloint2double(__double2hiint(key), __double2loint(key) & ~(int)CODE_MASK);
}

// 1/sqrt(x) for x >= 1 (a sample count) to ~1 ulp: v_rsq_f32 seed + two Newton steps in f64.
__device__ __forceinline__ double rsqrt_count(double x) {
    double y = (double)__frsqrt_rn((float)x);
    double h = 0.5 * x;
    y = y * fma(-h, y * y, 1.5);
    y = y * fma(-h, y * y, 1.5);
    return y;
}
// sqrt(x) for x >= 0 of moderate magnitude (a variance): v_rsq_f32 seed (clamped away from 0), one Newton step on
// y ~ 1/sqrt(x) (2e-14 relative), then s = x*y with one Heron correction (quadratic again: ~1e-16).  x = 0 -> 0.
__device__ __forceinline__ double sqrt_var(double x) {
    double y = (double)__frsqrt_rn(fmaxf((float)x, 1e-30f));
    y = y * fma(-0.5 * x, y * y, 1.5);
    double s = x * y;
    return fma(fma(-s, s, x), 0.5 * y, s);
}

// The reference's bound functions from a bucket's sufficient statistics, float64:
//   upper    = min(cap, mean + hoeff/sqrt(n))                                            S1:10-12
//   lower    = mean - hoeff/sqrt(n)                                                      S1:14-16
//   ci_lower = sum/n/(n+1) - 4*sigma/(n+1) + sum/(n+1) - hoeff/sqrt(n+1)                 S1:18-24
// sigma = population std (np.std, ddof=0).  The statistics are SHIFTED sums: sd = sum(x-K), qd = sum((x-K)^2)
// for a caller-chosen K near the data (first sample), so that var = qd/n - (sd/n)^2 does not cancel
// catastrophically when |mean| >> sigma; mean = K + sd/n, sum = n*K + sd.
struct Bounds { double upper, lower, ci_lower, mean; };
__device__ __forceinline__ Bounds bounds_from_sums(int n, double sd, double qd, double K, const DevParams& p) {
    double dn = (double)n;
    double r = rsqrt_count(dn), r1 = rsqrt_count(dn + 1.0);
    double inv_n = r * r, inv_n1 = r1 * r1;
    double md = sd * inv_n;
    double mean = K + md;
    double hw = p.hoeff * r;
    double var = fmax(fma(qd, inv_n, -md * md), 0.0);
    double sigma = sqrt_var(var);
    Bounds b;
    b.mean = mean;
    b.upper = fmin(p.cap, mean + hw);
    b.lower = mean - hw;
    // sum/n/(n+1) + sum/(n+1) == sum/n == mean exactly in real arithmetic (S1:24), so
    // ci_lower = mean - 4*sigma/(n+1) - hoeff/sqrt(n+1); the regrouping moves the result by O(1e-16*|mean|).
    b.ci_lower = fma(-4.0 * sigma, inv_n1, mean) - p.hoeff * r1;
    return b;
}
// V[s][a]: the rule action gets the optimistic bound (S1:88), every other candidate the pessimistic one (S1:90).
__device__ __forceinline__ double value_from_sums(int n, double sd, double qd, double K, bool is_rule,
                                                  const DevParams& p) {
    const Bounds b = bounds_from_sums(n, sd, qd, K, p);
    return is_rule ? b.upper : fmin(b.lower, b.ci_lower);
}

// max over N keys as a balanced tree (depth log2 N instead of an N-long dependent chain of v_max_f64)
template <int N>
__device__ __forceinline__ double tree_max(const double* k) {
    if constexpr (N == 1) return k[0];
    else return fmax(tree_max<N / 2>(k), tree_max<N - N / 2>(k + N / 2));
}

}  // namespace dcarl
