// Online confidence estimation + candidate arg-max over all states ("trace" mode).
// Replaces the reference hot loop S1:73-99 / S2:72-97 (Simulation_testing/Simulation_{1,2}/test_DCARL.py).
//
// Mapping to CDNA4: one wavefront (64 lanes) = one slice of 64 states, one lane = one state; the lane walks
// its state's records in arrival order (the loop is sequential per state by definition: every record's
// arg-max depends on all earlier records of that state).  Per 4 records a lane issues ONE 16-byte load of R,
// one 4-byte load of the action ids, and one 16-byte + one 4-byte store of the step traces; a wavefront's
// accesses are 1 KiB / 256 B contiguous ("sliced time-major, quad-packed" layout, include/dcarl.h).
// Per-bucket sufficient statistics (n, sum, sum of squares; f64) live in LDS, indexed [action][lane] so that
// the dynamic action index never causes a bank conflict; the A current values V[s][.] live in registers as
// tie-break-coded f64 keys so that the arg-max over candidates is a chain of v_max_f64.
// HBM-bound by design: 10 B per record/evaluation (f32 storage); the f64 evaluation is the co-limiter.
#include "common.h"

namespace dcarl {

template <typename T> struct Quad;
template <> struct Quad<float> { using type = float4; };
template <> struct Quad<double> { using type = double4; };

struct __attribute__((aligned(16))) SumPair { double s, q; };

template <typename T, int A_PAD, int PF>
__global__ __launch_bounds__(WAVE) void trace_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* __restrict__ act_step, double* __restrict__ V_out,
    int32_t* __restrict__ n_out, float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using Q4 = typename Quad<T>::type;
    __shared__ SumPair lds_sum[A_PAD][WAVE];
    __shared__ int lds_cnt[A_PAD][WAVE];

    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const int s = w * WAVE + lane;
    const int my_len = (s < S) ? len[s] : 0;
    const int64_t row0 = slice_row_off[w];
    const int rows = (int)(slice_row_off[w + 1] - row0);

    // longest stream in the slice bounds the loop (wave-uniform)
    int max_len = my_len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, __shfl_xor(max_len, off));
    max_len = min(max_len, rows);

#pragma unroll
    for (int a = 0; a < A_PAD; ++a) { lds_sum[a][lane] = SumPair{0.0, 0.0}; lds_cnt[a][lane] = 0; }

    double key[A_PAD];                                   // S1:50-53 initial table, tie-break coded
#pragma unroll
    for (int a = 0; a < A_PAD; ++a)
        key[a] = (a < A) ? encode_key(a == p.rule_act ? p.init_rule : p.init_other, a) : encode_key(-1e300, a);
    double best = key[0];
#pragma unroll
    for (int a = 1; a < A_PAD; ++a) best = fmax(best, key[a]);
    int latch = -1;

    const Q4* Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
    const uchar4* Aq = reinterpret_cast<const uchar4*>(act) + row0 / 4 * WAVE + lane;
    Q4* SVq = step_val ? reinterpret_cast<Q4*>(step_val) + row0 / 4 * WAVE + lane : nullptr;
    uchar4* SAq = step_act ? reinterpret_cast<uchar4*>(step_act) + row0 / 4 * WAVE + lane : nullptr;

    const int nquads = (max_len + 3) >> 2;
    // register prefetch ring: PF quads (= 4*PF records) ahead
    Q4 rbuf[PF];
    uchar4 abuf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        if (i < nquads && i * 4 < my_len) { rbuf[i] = Rq[(int64_t)i * WAVE]; abuf[i] = Aq[(int64_t)i * WAVE]; }
    }

    for (int qb = 0; qb < nquads; qb += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int qi = qb + i;
            if (qi >= nquads) break;
            const Q4 rv = rbuf[i];
            const uchar4 av = abuf[i];
            const int nxt = qi + PF;                      // refill this ring slot
            if (nxt < nquads && nxt * 4 < my_len) { rbuf[i] = Rq[(int64_t)nxt * WAVE]; abuf[i] = Aq[(int64_t)nxt * WAVE]; }

            const T rr[4] = {rv.x, rv.y, rv.z, rv.w};
            const int aa[4] = {av.x, av.y, av.z, av.w};
            T ov[4] = {T(0), T(0), T(0), T(0)};
            int oa[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = qi * 4 + j;
                if (t < my_len) {
                    const int a = aa[j] & (A_PAD - 1);
                    const double x = (double)rr[j];
                    SumPair sp = lds_sum[a][lane];              // S1:80 append == update sufficient statistics
                    const int n = lds_cnt[a][lane] + 1;
                    sp.s += x;
                    sp.q = fma(x, x, sp.q);
                    lds_sum[a][lane] = sp;
                    lds_cnt[a][lane] = n;
                    if (n > p.n_thres) {                        // S1:86
                        const double v = value_from_sums(n, sp.s, sp.q, a == p.rule_act, p);   // S1:87-90
                        const double k = encode_key(v, a);
#pragma unroll
                        for (int i2 = 0; i2 < A_PAD; ++i2) key[i2] = (a == i2) ? k : key[i2];
                    }
                    best = key[0];                              // S1:93-94: max + first arg-max
#pragma unroll
                    for (int i2 = 1; i2 < A_PAD; ++i2) best = fmax(best, key[i2]);
                    const int b = decode_action(best);
                    ov[j] = (T)best;
                    oa[j] = b;
                    if (latch < 0 && b != p.rule_act) latch = t + 1;   // S1:98-99
                }
            }
            if (qi * 4 < my_len) {
                if (SVq) { Q4 o; o.x = ov[0]; o.y = ov[1]; o.z = ov[2]; o.w = ov[3]; SVq[(int64_t)qi * WAVE] = o; }
                if (SAq) SAq[(int64_t)qi * WAVE] = make_uchar4(oa[0], oa[1], oa[2], oa[3]);
            }
        }
    }

    if (s < S) {
        if (act_step) act_step[s] = latch;
        if (vmax) vmax[s] = (float)best;
        if (amax) amax[s] = decode_action(best);
        if (V_out) {
#pragma unroll
            for (int a = 0; a < A_PAD; ++a) if (a < A) V_out[(int64_t)s * A + a] = strip_code(key[a]);
        }
        if (n_out) {
#pragma unroll
            for (int a = 0; a < A_PAD; ++a) if (a < A) n_out[(int64_t)s * A + a] = lds_cnt[a][lane];
        }
    }
}

template <typename T>
int launch_trace(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int S, int A,
                 const DevParams& p, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                 int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st) {
    const int W = (S + WAVE - 1) / WAVE;
    if (W == 0) return 0;
    dim3 grid(W), block(WAVE);
#define DCARL_LAUNCH(AP)                                                                                         \
    hipLaunchKernelGGL((trace_kernel<T, AP, 4>), grid, block, 0, st, R, act, slice_row_off, len, S, A, p, step_val, \
                       step_act, act_step, V_out, n_out, vmax, amax)
    if (A <= 8) DCARL_LAUNCH(8);
    else if (A <= 16) DCARL_LAUNCH(16);
    else DCARL_LAUNCH(32);
#undef DCARL_LAUNCH
    return 0;
}

template int launch_trace<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                 const DevParams&, float*, uint8_t*, int32_t*, double*, int32_t*, float*, int32_t*,
                                 hipStream_t);
template int launch_trace<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                  const DevParams&, double*, uint8_t*, int32_t*, double*, int32_t*, float*,
                                  int32_t*, hipStream_t);

}  // namespace dcarl
