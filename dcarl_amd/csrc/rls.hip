// Field variant of the confidence test ("RLS"): neighbour statistics by box query + two-sample z-test.
// Replaces Field_testing/.../tools/DCARL/stable_baselines/deepq/RLS.py (RLS:120-181):
//   every visited (state, action) row s owns the box [s - d, s + d] in 21 dimensions (RLS:68,193-194); a query point q
//   "visits" the rows whose box contains it (RLS:161-163, closed intervals); the statistics of a query are the count,
//   mean and population variance of the rewards of those rows (RLS:165-181, (-1,-1,-1) if there are none); the test
//   policy takes the first candidate 1..7 whose mean beats the rule action's with norm.cdf(z) > threshold (RLS:120-157).
// The reference finds the rows with an R-tree (third-party `rtree`); here it is a brute-force scan, which is exact by
// construction: one lane = one query, the rows of a chunk stream through scalar registers (they are uniform across
// the wavefront), up to 42 f64 compares per (row, query), leaving the row at the first bound no lane satisfies.
// Sums are accumulated per (row chunk, query) and reduced in a fixed order, so results are run-to-run identical.
#include "common.h"

namespace dcarl {

constexpr int RLS_DIM = DCARL_RLS_DIM;
constexpr int RLS_ROWS_PER_BLOCK = 1024;
constexpr int RLS_THREADS = 256;

// box of row i: lo = s - d, hi = s + d, rounded exactly like the reference's float64 subtraction / addition (RLS:193)
__global__ __launch_bounds__(256) void rls_boxes_kernel(const double* __restrict__ states, const double* __restrict__ half,
                                                        int64_t N, double* __restrict__ lo, double* __restrict__ hi) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * RLS_DIM) return;
    const double s = states[e], d = half[e % RLS_DIM];
    lo[e] = s - d;
    hi[e] = s + d;
}

// partial statistics of every query over one chunk of rows: part[(chunk * Q + q) * 3 + {0,1,2}] = {count, sum, sum of squares}
__global__ __launch_bounds__(RLS_THREADS) void rls_partial_kernel(
    const double* __restrict__ lo, const double* __restrict__ hi, const double* __restrict__ reward, int64_t N,
    const double* __restrict__ queries, int Q, double* __restrict__ part) {
    const int qi = blockIdx.y * RLS_THREADS + threadIdx.x;
    const bool live = qi < Q;
    double q[RLS_DIM];
#pragma unroll
    for (int k = 0; k < RLS_DIM; ++k) q[k] = queries[(int64_t)min(qi, Q - 1) * RLS_DIM + k];
    const int64_t r0 = (int64_t)blockIdx.x * RLS_ROWS_PER_BLOCK, r1 = min(N, r0 + RLS_ROWS_PER_BLOCK);
    double cnt = 0.0, sum = 0.0, sq = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
        const double* L = lo + r * RLS_DIM;               // wave-uniform addresses: scalar loads
        const double* H = hi + r * RLS_DIM;
        // `&&` on purpose: the compiler turns every bound into "scalar load, compare, skip the rest if no lane is left",
        // so a row costs one 8-byte load when no query of the wavefront fits its first dimension (the ego position).
        // Measured alternatives, all slower on the field-shaped table (2.94 ms): one exit test per bound instead of the
        // three groups below 3.36 ms; eight rows' loads batched 3.64 ms; membership as lane masks with the bounds loaded
        // up front 3.45 ms -- they give up the cheap exit.
        bool in = live;
#pragma unroll
        for (int k = 0; k < 4; ++k) in = in && (L[k] <= q[k]) && (q[k] <= H[k]);
        if (!__any(in)) continue;                         // the ego vehicle's four dimensions
#pragma unroll
        for (int k = 4; k < 12; ++k) in = in && (L[k] <= q[k]) && (q[k] <= H[k]);
        if (!__any(in)) continue;
#pragma unroll
        for (int k = 12; k < RLS_DIM; ++k) in = in && (L[k] <= q[k]) && (q[k] <= H[k]);
        const double v = reward[r];
        cnt += in ? 1.0 : 0.0;
        sum += in ? v : 0.0;
        sq = in ? fma(v, v, sq) : sq;
    }
    if (live) {
        double* o = part + ((int64_t)blockIdx.x * Q + qi) * 3;
        o[0] = cnt; o[1] = sum; o[2] = sq;
    }
}

// RLS:165-181: count, np.mean, np.var (population) of the visited rewards; (-1, -1) when nothing was visited
__global__ __launch_bounds__(256) void rls_reduce_kernel(const double* __restrict__ part, int chunks, int Q,
                                                         int64_t* __restrict__ count, double* __restrict__ mean,
                                                         double* __restrict__ var) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= Q) return;
    double cnt = 0.0, sum = 0.0, sq = 0.0;
    for (int c = 0; c < chunks; ++c) {
        const double* o = part + ((int64_t)c * Q + qi) * 3;
        cnt += o[0]; sum += o[1]; sq += o[2];
    }
    count[qi] = (int64_t)cnt;
    if (cnt == 0.0) { mean[qi] = -1.0; var[qi] = -1.0; return; }
    const double m = sum / cnt;
    mean[qi] = m;
    var[qi] = fmax(sq / cnt - m * m, 0.0);
}

// RLS:120-157 act_test for B decisions: statistics laid out [B][1 + n_cand], column 0 = rule action, column c = candidate c
__global__ __launch_bounds__(256) void rls_decide_kernel(const int64_t* __restrict__ count, const double* __restrict__ mean,
                                                         const double* __restrict__ var, int B, int n_cand,
                                                         dcarl_rls_params_t p, int32_t* __restrict__ action) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t base = (int64_t)b * (1 + n_cand);
    const double n_rule = (double)count[base], mean_rule = mean[base], var_rule = var[base];
    int chosen = 0;
    for (int c = 1; c <= n_cand && chosen == 0; ++c) {
        const double n_rl = (double)count[base + c];
        // RLS:141 (the three skip conditions), then RLS:144-151
        if (n_rule < (double)p.visited_times_thres || n_rl < (double)p.min_rl_visits || mean_rule > p.rule_mean_gate) continue;
        const double diff = mean[base + c] - mean_rule, sd = sqrt(var_rule / n_rule + var[base + c] / n_rl);
        // norm.cdf(diff / sd); the degenerate sd == 0 cases spelled out (the library is built with -fno-honor-nans):
        // +-inf -> cdf 1 / 0, and 0/0 = NaN never exceeds the threshold in the reference
        double cdf;
        if (sd > 0.0) cdf = 0.5 * erfc(-(diff / sd) * 0.70710678118654752440);
        else cdf = diff > 0.0 ? 1.0 : (diff < 0.0 ? 0.0 : -1.0);
        if (cdf > p.confidence_thres) chosen = c;
    }
    action[b] = chosen;
}

// RLS:78-118 the TRAIN-time gate for B observations: should_use_rule = the rule action's neighbourhood is visited fewer than
// visited_times_thres times (RLS:107-108), or the exploration draw explore[b] ~ U(-1, 0) falls below its mean value
// (RLS:112-114: "rule performs good"); act_train returns 0 then, else the DQN's action (RLS:85-89).  explore is injected (the
// reference draws random.uniform(-1, 0) only for observations that pass the first test, in order: the caller's business).
__global__ __launch_bounds__(256) void rls_gate_train_kernel(const int64_t* __restrict__ count_rule, const double* __restrict__ mean_rule,
                                                             const double* __restrict__ explore, const int32_t* __restrict__ rl_action,
                                                             int32_t B, int32_t visited_times_thres, int32_t* __restrict__ action,
                                                             uint8_t* __restrict__ use_rule) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const bool rule = count_rule[b] < visited_times_thres || explore[b] < mean_rule[b];
    if (use_rule) use_rule[b] = rule ? 1 : 0;
    if (action) action[b] = rule ? 0 : rl_action[b];
}
int launch_rls_gate_train(const int64_t* count_rule, const double* mean_rule, const double* explore, const int32_t* rl_action, int32_t B,
                          int32_t thres, int32_t* action, uint8_t* use_rule, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(rls_gate_train_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, count_rule, mean_rule, explore,
                       rl_action, B, thres, action, use_rule);
    return 0;
}

int64_t rls_workspace_bytes(int64_t N, int32_t Q) {
    // (checked: a size that does not fit 63 bits is "invalid" = 0, not undefined behaviour — the sanitized host build found the product)
    if (N < 0 || Q < 0 || N > ((int64_t)1 << 40)) return 0;
    const int64_t chunks = (N + RLS_ROWS_PER_BLOCK - 1) / RLS_ROWS_PER_BLOCK;
    int64_t part, words;
    if (__builtin_mul_overflow(chunks, (int64_t)Q * 3, &part) || __builtin_add_overflow(2 * N * RLS_DIM, part, &words) ||
        __builtin_mul_overflow(words, (int64_t)sizeof(double), &words))
        return 0;
    return words;
}

int launch_rls_stats(const double* states, const double* reward, int64_t N, const double* half, const double* queries,
                     int32_t Q, void* ws, int64_t* count, double* mean, double* var, hipStream_t st) {
    if (Q == 0) return 0;
    double* lo = reinterpret_cast<double*>(ws);
    double* hi = lo + N * RLS_DIM;
    double* part = hi + N * RLS_DIM;
    const int chunks = (int)((N + RLS_ROWS_PER_BLOCK - 1) / RLS_ROWS_PER_BLOCK);
    if (N > 0) {
        hipLaunchKernelGGL(rls_boxes_kernel, dim3((unsigned)((N * RLS_DIM + 255) / 256)), dim3(256), 0, st, states, half, N,
                           lo, hi);
        hipLaunchKernelGGL(rls_partial_kernel, dim3((unsigned)chunks, (unsigned)((Q + RLS_THREADS - 1) / RLS_THREADS)),
                           dim3(RLS_THREADS), 0, st, lo, hi, reward, N, queries, Q, part);
    }
    hipLaunchKernelGGL(rls_reduce_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, part, chunks, Q, count, mean,
                       var);
    return 0;
}

int launch_rls_decide(const int64_t* count, const double* mean, const double* var, int32_t B, int32_t n_cand,
                      const dcarl_rls_params_t& p, int32_t* action, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(rls_decide_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, count, mean, var, B, n_cand, p,
                       action);
    return 0;
}

}  // namespace dcarl
