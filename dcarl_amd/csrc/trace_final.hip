// The FINAL table of an online record table, without the loop's per-record work.
//
// S1:73-99 re-evaluates V[s][a] after every record, but what the table holds when the loop ends is, per bucket, the evaluation
// of its LAST record — a function of the bucket's sufficient statistics (n, sum(x-K), sum((x-K)^2)) alone (S1:86-90) — or the
// prior when the bucket never passed the threshold (S1:50-53,86).  A caller that wants only TSRL_value / the final arg-max
// (dcarl_trace_* with step_val, step_act AND act_step all NULL: ConfidenceEstimator.bounds_from_table,
// bounds_from_reference_table) therefore needs the statistics stage of the loop and ONE evaluation per bucket: ~8 vector and 4
// LDS instructions per record instead of 55 and 13.5, no cross-wave hand-over, HBM-bound at 5 bytes per record.
//
// Same layout walk, same arithmetic in the same order as the online kernels (trace_common.h single_append: the additions happen
// in arrival order; common.h value_from_sums / encode_key), so V, n, max and arg-max equal the online kernel's bit for bit
// (tests/test_gpu_parity.py::test_final_table_kernel_equals_the_online_kernel).  What it cannot give is the activation latch
// (S1:98-99 needs the arg-max after every record): callers that pass act_step get the online kernel.
//
// One wavefront per 64-state slice, lane = state; the lane's A buckets live in LDS ([a][lane]: a per-lane action id is an LDS
// address, not a register index), 20 bytes each: 14 KiB per wavefront at 11 candidates = 11 wavefronts per CU.  Four quads
// (80 bytes per lane) are in flight while four are appended.
#include "trace_common.h"

namespace dcarl {

namespace {
constexpr int FT_PF = 8;                                            // quads per prefetch bank

template <typename T>
__global__ __launch_bounds__(WAVE) void final_table_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off, const int32_t* __restrict__ len,
    const int32_t* __restrict__ slot_state, int S, int A, DevParams p, double* __restrict__ V_out, int32_t* __restrict__ n_out,
    float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using Q4 = typename Quad<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    SumPair* lsum = reinterpret_cast<SumPair*>(smem);               // [A][WAVE]
    int* lcnt = reinterpret_cast<int*>(lsum + A * WAVE);            // [A][WAVE]
    for (int a = 0; a < A; ++a) {                                   // (a lane only ever touches its own column: no barrier)
        lsum[a * WAVE + lane] = SumPair{0.0, 0.0};
        lcnt[a * WAVE + lane] = 0;
    }
    const int s = w * WAVE + lane;
    const int64_t row0 = slice_row_off[w];
    const int rows = (int)(slice_row_off[w + 1] - row0);
    const int my_len = (s < S) ? min(len[s], rows) : 0;
    int max_len = my_len, min_len = my_len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_len = max(max_len, __shfl_xor(max_len, off));
        min_len = min(min_len, __shfl_xor(min_len, off));
    }
    max_len = __builtin_amdgcn_readfirstlane(max_len);
    min_len = __builtin_amdgcn_readfirstlane(min_len);
    const int nquads = (max_len + 3) >> 2, nfast = min_len >> 2;    // quads with every lane live come first
    const Q4* Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
    const unsigned* Aq = reinterpret_cast<const unsigned*>(act) + row0 / 4 * WAVE + lane;
    const double shift = (my_len > 0) ? (double)R[(row0 * WAVE) + lane * 4] : 0.0;    // K = the state's first reward
    const int amax_id = A - 1;

    auto append = [&](int a_raw, double xr) __attribute__((always_inline)) {
        const int a = min(a_raw, amax_id);
        const double x = xr - shift;
        const int e = a * WAVE + lane;
        const SumPair b = lsum[e];
        const int n = lcnt[e] + 1;
        lsum[e] = SumPair{b.s + x, fma(x, x, b.q)};
        lcnt[e] = n;
        asm volatile("" ::: "memory");                              // (the next record may hit the same bucket: LDS executes in order)
    };
    // Four records at once, every lane live: the four bucket reads go out together (ONE LDS round trip per quad instead of four
    // — a record-by-record wave spent most of its time waiting for the LDS), a record whose bucket an earlier record of the quad
    // already touched takes that record's result from registers (6 compares + 24 selects per quad; the LATEST earlier match
    // wins), and the four write-backs leave in record order (the LDS executes in order: the last write to a bucket is its newest
    // value).  The counts are not needed before the end: a fire-and-forget ds_add each.  Additions happen in arrival order
    // exactly as single_append (trace_common.h) does them.
    auto append_quad = [&](const Q4& r, unsigned av) __attribute__((always_inline)) {
        const int a0 = min((int)(av & 255u), amax_id), a1 = min((int)((av >> 8) & 255u), amax_id),
                  a2 = min((int)((av >> 16) & 255u), amax_id), a3 = min((int)(av >> 24), amax_id);
        const int e0 = a0 * WAVE + lane, e1 = a1 * WAVE + lane, e2 = a2 * WAVE + lane, e3 = a3 * WAVE + lane;
        const SumPair b0 = lsum[e0], b1 = lsum[e1], b2 = lsum[e2], b3 = lsum[e3];
        __hip_atomic_fetch_add(&lcnt[e0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&lcnt[e1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&lcnt[e2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(&lcnt[e3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const double x0 = (double)r.x - shift, x1 = (double)r.y - shift, x2 = (double)r.z - shift, x3 = (double)r.w - shift;
        const double s0 = b0.s + x0, q0 = fma(x0, x0, b0.q);
        double bs = b1.s, bq = b1.q;
        if (a1 == a0) { bs = s0; bq = q0; }
        const double s1 = bs + x1, q1 = fma(x1, x1, bq);
        bs = b2.s; bq = b2.q;
        if (a2 == a0) { bs = s0; bq = q0; }
        if (a2 == a1) { bs = s1; bq = q1; }
        const double s2 = bs + x2, q2 = fma(x2, x2, bq);
        bs = b3.s; bq = b3.q;
        if (a3 == a0) { bs = s0; bq = q0; }
        if (a3 == a1) { bs = s1; bq = q1; }
        if (a3 == a2) { bs = s2; bq = q2; }
        const double s3 = bs + x3, q3 = fma(x3, x3, bq);
        asm volatile("" ::: "memory");
        lsum[e0] = SumPair{s0, q0};
        asm volatile("" ::: "memory");                              // (the write-backs stay in record order)
        lsum[e1] = SumPair{s1, q1};
        asm volatile("" ::: "memory");
        lsum[e2] = SumPair{s2, q2};
        asm volatile("" ::: "memory");
        lsum[e3] = SumPair{s3, q3};
        asm volatile("" ::: "memory");
    };

    int q = 0;
    Q4 rb[FT_PF];
    unsigned ab[FT_PF];
    if (nfast >= FT_PF) {
#pragma unroll
        for (int i = 0; i < FT_PF; ++i) { rb[i] = Rq[(int64_t)i * WAVE]; ab[i] = Aq[(int64_t)i * WAVE]; }
        for (; q + FT_PF <= nfast; q += FT_PF) {
            Q4 rc[FT_PF];
            unsigned ac[FT_PF];
#pragma unroll
            for (int i = 0; i < FT_PF; ++i) { rc[i] = rb[i]; ac[i] = ab[i]; }
            if (q + 2 * FT_PF <= nfast) {                           // wave-uniform: the next bank, in flight under this one's appends
#pragma unroll
                for (int i = 0; i < FT_PF; ++i) { rb[i] = Rq[(int64_t)(q + FT_PF + i) * WAVE]; ab[i] = Aq[(int64_t)(q + FT_PF + i) * WAVE]; }
            }
#pragma unroll
            for (int i = 0; i < FT_PF; ++i) append_quad(rc[i], ac[i]);
        }
    }
    for (; q < nquads; ++q) {                                       // the last few whole quads and the ragged ends: per-lane guards
        if (q * 4 < my_len) {
            const Q4 r = Rq[(int64_t)q * WAVE];
            const unsigned av = Aq[(int64_t)q * WAVE];
            const double xr[4] = {(double)r.x, (double)r.y, (double)r.z, (double)r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q * 4 + j < my_len) append((int)((av >> (8 * j)) & 255u), xr[j]);
        }
    }
    if (s >= S) return;
    const int so = slot_state ? slot_state[s] : s;                  // per-state outputs go to the state's own row, not the slot's
    double best = 0.0;
    for (int a = 0; a < A; ++a) {
        const int n = lcnt[a * WAVE + lane];
        double v = (a == p.rule_act) ? p.init_rule : p.init_other;  // S1:50-53: the prior stands until the bucket passes the threshold
        if (n > p.n_thres) {                                        // S1:86
            const SumPair sp = lsum[a * WAVE + lane];
            v = value_from_sums(n, sp.s, sp.q, shift, a == p.rule_act, p);      // S1:87-90 on the whole bucket
        }
        const double key = encode_key(v, a);
        best = (a == 0) ? key : fmax(best, key);                    // S1:93-94: max / FIRST arg-max through the tie-break code
        if (V_out) V_out[(int64_t)so * A + a] = strip_code(key);
        if (n_out) n_out[(int64_t)so * A + a] = n;
    }
    if (vmax) vmax[so] = (float)best;
    if (amax) amax[so] = decode_action(best);
}
}  // namespace

template <typename T>
int launch_final_table(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int S, int A,
                       const DevParams& p, double* V_out, int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st) {
    const int W = slices_of(S);
    if (W == 0) return 0;
    const unsigned lds = (unsigned)A * WAVE * 20u;                  // <= 40 KiB at the ABI's 32 candidates
    hipLaunchKernelGGL((final_table_kernel<T>), dim3(W), dim3(WAVE), lds, st, R, act, slice_row_off, len, slot_state, S, A, p, V_out, n_out,
                       vmax, amax);
    note_kernel("final_table_kernel<%s>", sizeof(T) == 4 ? "float" : "double");
    return 0;
}
template int launch_final_table<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&,
                                       double*, int32_t*, float*, int32_t*, hipStream_t);
template int launch_final_table<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&,
                                        double*, int32_t*, float*, int32_t*, hipStream_t);

}  // namespace dcarl
