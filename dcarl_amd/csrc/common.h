// Shared device helpers for the DCARL gfx950 kernels.  CDNA4 only: wave64, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dcarl.h"

// The product library reads NO environment variable and carries no kernel instance that exists only for measurements.  The
// overrides of the launchers' own choices (run a kernel form at a size where the launcher would pick the other one; the two- /
// four-wave and unfenced instances of the online kernel) exist in the -DDCARL_AB_BUILD variant only: libdcarl_hip_ab.so
// (dcarl_amd/build.py variant "ab"), loaded by the tests that exercise every kernel instance and by tools/'s A/B scripts.
#ifdef DCARL_AB_BUILD
#include <stdlib.h>
#define DCARL_KNOB(name) getenv(name)
#else
#define DCARL_KNOB(name) (static_cast<const char*>(nullptr))
#endif

namespace dcarl {

// host side: remembers (thread-local) the name of the kernel a launcher chose; dcarl_last_kernel() hands it out (abi.hip)
void note_kernel(const char* fmt, ...);
// host side: something a launcher found wrong BEFORE its launch; the entry point's after_launch() reports it (thread-local, abi.hip)
void note_launch_problem(const char* fmt, ...);

// A kernel that asks for more than 64 KiB of dynamic LDS needs its limit raised once PER DEVICE (hipFuncSetAttribute).  Called before
// every launch: the first call on each device sets the attribute (a bit per device in the call site's own mask), later ones cost an
// atomic load; a failure — a device with less LDS than the kernel needs — is reported by name instead of surfacing as a bare
// "invalid argument" from the launch that follows.
inline bool raise_lds_limit(const void* kernel, int bytes, unsigned long long* done_mask, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_RELAXED) & bit) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        note_launch_problem("%s needs %d bytes of dynamic LDS and device %d refused the limit (%s)", what, bytes, dev, hipGetErrorString(e));
        return false;
    }
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELAXED);
    return true;
}
#define DCARL_RAISE_LDS_LIMIT(bytes, ...)                                                                             \
    do {                                                                                                              \
        static unsigned long long dcarl_lds_done_ = 0;                                                                \
        (void)::dcarl::raise_lds_limit(reinterpret_cast<const void*>(&__VA_ARGS__), (int)(bytes), &dcarl_lds_done_, #__VA_ARGS__); \
    } while (0)

constexpr int WAVE = 64;
// ceil(S / 64) and ceil(n / d) without the signed overflow `S + 63` has at S > 2^31 - 64 (found by the sanitized host build:
// tests/test_abi_host_sanitized.py)
__host__ __device__ inline int slices_of(int S) { return (int)(((int64_t)S + (WAVE - 1)) / WAVE); }
__host__ __device__ inline int64_t ceil_div64(int64_t n, int64_t d) { return (n + d - 1) / d; }
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
// (32-bit operations only; s = hi >> 31 arithmetic is 0 for v >= 0 and -1 for v < 0, so the code is a ^ (31 & ~s).)
// SignMask is how s is obtained: the plain shift here; the online kernels pass an asm variant (trace_common.h).
struct ShiftSign { __device__ __forceinline__ static int of(int hi) { return hi >> 31; } };
template <class SignMask = ShiftSign>
__device__ __forceinline__ double encode_key(double v, int a) {
    const int hi = __double2hiint(v);
    const int s = SignMask::of(hi);
    // (lo & ~31) | (a ^ (31 & ~s)), written so that it is two three-input bit operations: (lo | 31) ^ (a ^ (s & 31))
    const int lo = (__double2loint(v) | (int)CODE_MASK) ^ (a ^ (s & (int)CODE_MASK));
    return __hiloint2double(hi, lo);
}
template <class SignMask = ShiftSign>
__device__ __forceinline__ int decode_action(double key) {
    const int s = SignMask::of(__double2hiint(key));
    return (int)CODE_MASK & (__double2loint(key) ^ ~s);
}
__device__ __forceinline__ double strip_code(double key) {
    return __hiloint2double(__double2hiint(key), __double2loint(key) & ~(int)CODE_MASK);
}

// 1/sqrt(x) to ~1 ulp from the 1-ulp v_rsq_f32 seed with ONE cubically convergent (Halley) step in f64:
// e = 1 - x*y0^2 (|e| ~ 2e-7), y1 = y0*(1 + e/2 + 3e^2/8)  =>  relative error ~ e^3 ~ 1e-20.
// 5 f64 operations instead of the 7 of two Newton steps.  x must be in f32 range and > 0.
__device__ __forceinline__ double rsqrt_halley(double x, float xf) {
    const double y0 = (double)__frsqrt_rn(xf);
    const double e = fma(-x, y0 * y0, 1.0);
    const double c = e * fma(0.375, e, 0.5);
    return fma(y0, c, y0);
}
__device__ __forceinline__ double rsqrt_count(double x) { return rsqrt_halley(x, (float)x); }
// sqrt(x) for x >= 1e-30 of moderate magnitude (a variance, floored by the caller): one coupled Newton step on
// (g, h) = (x*y0, y0/2) from the v_rsq_f32 seed, g1 = g + g*(1/2 - g*h): relative error 3/8 e^2 <= 6e-15 for the seed's
// |e| <= 1.3e-7 (the term it feeds, 4*sigma/(n+1), is < 1/3 sigma, so V moves by < 1e-15 relative).  4 f64 operations.
__device__ __forceinline__ double sqrt_var(double x) {
    const double y0 = (double)__frsqrt_rn((float)x);
    const double g = x * y0, h = 0.5 * y0;
    return fma(g, fma(-g, h, 0.5), g);
}

// The reference's bound functions from a bucket's sufficient statistics, float64:
//   upper    = min(cap, mean + hoeff/sqrt(n))                                            S1:10-12
//   lower    = mean - hoeff/sqrt(n)                                                      S1:14-16
//   ci_lower = sum/n/(n+1) - 4*sigma/(n+1) + sum/(n+1) - hoeff/sqrt(n+1)                 S1:18-24
// sigma = population std (np.std, ddof=0).  The statistics are SHIFTED sums: sd = sum(x-K), qd = sum((x-K)^2)
// for a caller-chosen K near the data (first sample), so that var = qd/n - (sd/n)^2 does not cancel
// catastrophically when |mean| >> sigma; mean = K + sd/n, sum = n*K + sd.
struct Bounds { double upper, lower, ci_lower, mean; };
// the two functions of the count alone: r = 1/sqrt(n) and rho = 2/sqrt(n+1) = rsqrt((n+1)/4); rho^2 = 4/(n+1) is the
// factor of sigma in S1:24 and hoeff/sqrt(n+1) = (hoeff/2)*rho, so the "4*" costs nothing.  (The online kernels keep
// {r, rho} per count in an LDS table, one ds_read_b128 per record.)
struct CountRoots { double r, rho; };
__device__ __forceinline__ CountRoots count_roots(int n) {
    const double dn = (double)n;
    return CountRoots{rsqrt_count(dn), rsqrt_count(0.25 * (dn + 1.0))};
}
__device__ __forceinline__ Bounds bounds_from_roots(double r, double rho, double sd, double qd, double K,
                                                    const DevParams& p) {
    double inv_n = r * r;
    double md = sd * inv_n;
    double mean = K + md;
    // floored at 1e-30 instead of 0: the floor doubles as the guard of the f32 reciprocal-square-root seed (no separate
    // clamp); a constant bucket gets sigma = 1e-15 instead of 0, which moves V by < 1e-17
    double var = fmax(fma(qd, inv_n, -md * md), 1e-30);
    double sigma = sqrt_var(var);
    Bounds b;
    b.mean = mean;
    b.upper = fmin(p.cap, fma(p.hoeff, r, mean));
    b.lower = fma(-p.hoeff, r, mean);
    // sum/n/(n+1) + sum/(n+1) == sum/n == mean exactly in real arithmetic (S1:24), so
    // ci_lower = mean - 4*sigma/(n+1) - hoeff/sqrt(n+1) = mean - rho*(sigma*rho + hoeff/2): two fma (the three-term form
    // needed rho^2 as well: one instruction more per record); the regrouping moves the result by O(1e-16*|mean|).
    b.ci_lower = fma(-rho, fma(sigma, rho, 0.5 * p.hoeff), mean);
    return b;
}
__device__ __forceinline__ Bounds bounds_from_sums(int n, double sd, double qd, double K, const DevParams& p) {
    const CountRoots c = count_roots(n);
    return bounds_from_roots(c.r, c.rho, sd, qd, K, p);
}
// V[s][a]: the rule action gets the optimistic bound (S1:88), every other candidate the pessimistic one (S1:90).
__device__ __forceinline__ double value_from_roots(double r, double rho, double sd, double qd, double K, bool is_rule,
                                                   const DevParams& p) {
    // (folding the role into the sign of the half-width and into ONE min — min(fma(+-hoeff, r, mean), is_rule ? cap : ci)
    // — is bit-identical and one instruction less on paper; the compiler spends two more on building the signed constant)
    const Bounds b = bounds_from_roots(r, rho, sd, qd, K, p);
    return is_rule ? b.upper : fmin(b.lower, b.ci_lower);
}
__device__ __forceinline__ double value_from_sums(int n, double sd, double qd, double K, bool is_rule,
                                                  const DevParams& p) {
    const CountRoots c = count_roots(n);
    return value_from_roots(c.r, c.rho, sd, qd, K, is_rule, p);
}

// max over N keys as a balanced tree (depth log2 N instead of an N-long dependent chain of v_max_f64)
template <int N>
__device__ __forceinline__ double tree_max(const double* k) {
    if constexpr (N == 1) return k[0];
    else return fmax(tree_max<N / 2>(k), tree_max<N - N / 2>(k + N / 2));
}


// non-temporal 16- / 8- / 4-byte accesses (streams that are read or written once): the builtins want native vector types
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
    const nt_u4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<nt_u4*>(p));
}
__device__ __forceinline__ void nt_store4(void* p, unsigned a) { __builtin_nontemporal_store(a, reinterpret_cast<unsigned*>(p)); }
__device__ __forceinline__ uint4 nt_load16(const void* p) {
    const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 nt_load8(const void* p) {
    const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p));
    return make_uint2(v.x, v.y);
}

}  // namespace dcarl
