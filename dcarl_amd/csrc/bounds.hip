// Final-state ("batch") confidence evaluation from samples sorted by (state, action).
// Replaces S1:10-24 (upper_bound / lower_bound / CI_lower_bound) + S1:86-95 evaluated once per bucket.
// HBM-bound: 4 B per sample read once (f32 storage) + 12*A+8 B per state written.  (tools/experiments/ubench_stream.hip: this
// request pattern streams at 6.3 TB/s read-only on the box and at 4.9-5.2 TB/s once 3 % of per-pass result writes are
// interleaved — the ceiling this kernel runs at.)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#ifndef DCARL_BQ_TAIL
#define DCARL_BQ_TAIL 0
#endif
#ifndef DCARL_BQ_RU
#define DCARL_BQ_RU 4
#endif
#ifndef DCARL_BQ_LINE
#define DCARL_BQ_LINE 128
#endif
#ifndef DCARL_BOUNDS_NT
#define DCARL_BOUNDS_NT 0       // non-temporal loads of the samples in bounds_quad_kernel: slower on every shape (configs[3] 0.92 -> 1.22 ms: neighbouring
                                // buckets share lines; configs[4] 0.39 -> 0.45, configs[1] 0.90 -> 0.92) — kept as a build flag only
#endif

namespace dcarl {

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

// accumulate the shifted sums sum(x-K), sum((x-K)^2) of one 16-byte vector
__device__ __forceinline__ void acc16(const float4& v, double K, double& s, double& q) {
    double a = v.x - K, b = v.y - K, c = v.z - K, d = v.w - K;
    s += (a + b) + (c + d);
    q = fma(a, a, q); q = fma(b, b, q); q = fma(c, c, q); q = fma(d, d, q);
}
__device__ __forceinline__ void acc16(const double2& v, double K, double& s, double& q) {
    double a = v.x - K, b = v.y - K;
    s += a + b;
    q = fma(a, a, q); q = fma(b, b, q);
}

// ---- the final-state kernel: 4 (or 8) lanes per bucket, 16 (8) buckets per pass, 16 states per wavefront -------------
// The flat (state, action) bucket list is cut into blocks of 16 states = 16*A buckets; a wavefront owns a block and walks it
// in passes of 64/G consecutive buckets (across state boundaries: every pass is full whatever A is).  A bucket belongs to a
// cluster of G lanes; ALL its 16-byte vectors (up to NV per lane; a remainder loop takes longer buckets) and its unaligned
// head / tail samples are requested before any is consumed, the f64 partial sums meet in log2(G) quad-permute steps, the
// bound formulas run once per pass with 64/G distinct buckets in flight, and the tie-coded key goes to LDS.  After the last
// pass lane k reads the A keys of state k and takes their maximum (S1:93-94).  Plain CSR: no alignment or padding contract.
// ~180 VALU instructions per 4 KB state (round 1's one-state-per-wavefront kernels: ~530), VALU 30 % busy: memory bound.
//
// Software-pipelined with D register buffers: the requests of pass p+1 are issued right after pass p's samples have been
// accumulated (their registers are free again) and BEFORE pass p's reduction / bound formulas / stores, so a wavefront
// always has a pass worth of loads in flight while it evaluates (the non-pipelined form has none during ~150 VALU
// instructions per pass; PMC: 71 % of its wave-cycles are waits at 4.7 waves per SIMD).
// vmcnt retires in order and the waitcnt pass can only leave younger requests outstanding if it can COUNT them, so
// every request of a pass is unconditional: a lane with fewer than NV vectors re-reads its last one (or the first
// element of `values` for an empty range; a cache hit either way) and the predicate moves to the accumulation.  The
// offsets of pass p+2 are requested after pass p+1's samples and used only once those have been consumed.
// CSR = seg_off given (a compile-time fact, for the same reason).
template <typename T, int G, int NV, bool CSR, int D>
__global__ __launch_bounds__(256) void bounds_quad_kernel(
    const T* __restrict__ values, const int64_t* __restrict__ seg_off, int64_t n_dense, int S, int A, int amul, DevParams p,
    double* __restrict__ V_out, int32_t* __restrict__ n_out, float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using V16 = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    constexpr int SB = 16;
    __shared__ double keys[256 / WAVE][SB * DCARL_MAX_ACTIONS];
    __shared__ int32_t counts[256 / WAVE][SB * DCARL_MAX_ACTIONS];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    constexpr int BP = WAVE / G;
    const int cl = lane / G, sub = lane % G;
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * (256 / WAVE) + wv);
    const int s0 = task * SB;
    if (s0 >= S) return;
    const int ns = min(SB, S - s0);
    const int nb = ns * A;
    const int64_t g0 = (int64_t)s0 * A;
    double* kw = keys[wv];
    int32_t* nw = counts[wv];

    // the loaded offsets are not touched before the pass that needs them is issued
    auto bucket_range = [&](int j, int64_t& b, int64_t& e) {
        const int64_t g = g0 + min(max(j, 0), nb - 1);
        if (CSR) { b = seg_off[g]; e = seg_off[g + 1]; }
        else { b = g * n_dense; e = b + n_dense; }
    };
    // D register buffers = D passes in flight (buffer d holds passes d, d+D, d+2D, ...)
    // Round 6: the window of a LONG bucket starts on the 128-byte line grid.  Until then it began at the bucket's first aligned
    // vector, so it ended — and the remainder loop began — in the middle of a line: that line was requested at issue time and again
    // by the remainder loop a whole pass later, and with ~10 MB streaming through an XCD's 4-MiB L2 in between it had left the L2
    // and came from HBM twice.  Likewise a bucket's last line, which the neighbouring cluster requests at issue time as the first
    // line of ITS window.  Two lines per long bucket = configs[3]'s 1.19x read traffic (uniform tables, FETCH_SIZE over samples +
    // offsets: 1.007 at 91 samples per bucket — they fit the window —, 1.156 at 364, 1.082 at 729, 1.034 at 1 818:
    // tools/experiments/exp_bounds_traffic.py, profiles/r06_ab_bounds_line_grid.txt).  A bucket that does not fit the window now
    // starts it at the line that holds its first whole vector (the vectors before that are the previous bucket's: requested, not
    // used) and the remainder loop runs on whole lines up to the last one: 1.086 / 1.045 / 1.018, configs[3] 0.960 -> 0.910 ms,
    // configs[1] 0.902 -> 0.890, configs[4] unchanged (its 64-sample buckets fit).  -DDCARL_BQ_TAIL=1 also requests the bucket's last
    // line at issue time, in TV slots of its own, in the same instruction as the neighbour's window: 1.004 ... 1.010 on every table,
    // i.e. the whole excess is explained — but the 8 / 16 more registers cost the fourth wave per SIMD and every shape is slower
    // (configs[1] 0.861 -> 0.926, configs[3] 0.89 -> 0.92); -DDCARL_BQ_TAIL=2 (tail slots for <4,6,...> only, where the occupancy
    // is 3 either way): configs[3] 0.905 against 0.892, no gain.  So the tail line stays with the remainder loop.
    constexpr int RU = DCARL_BQ_RU;                               // vectors per lane and turn of the remainder loop
    constexpr int LV = (G * NV) % (DCARL_BQ_LINE / 16) == 0 ? DCARL_BQ_LINE / 16 : 128 / 16;   // 16-byte vectors per line
    constexpr bool TAIL = DCARL_BQ_TAIL == 1 || (DCARL_BQ_TAIL == 2 && G == 4 && NV >= 6);
    constexpr int TV = TAIL ? (LV + G - 1) / G : 0;               // slots per lane for the bucket's last line
    constexpr int TVA = TV > 0 ? TV : 1;
    static_assert((G * NV) % LV == 0, "the pipelined window ends on a line boundary");
    // vectors are counted from the window's first line: the bucket's whole vectors are [rvb, rve), its last (partial) line starts
    // at rtl (== rve rounded down to a line; the remainder loop runs over [G*NV, rtl)); vp = this lane's first vector of the window
    struct Meta { int n, nh, nt, rvb, rve, rtl; const V16* vp; };
    Meta m[D];
    T kraw[D], xh[D], xt[D];
    V16 x[D][NV], xl[D][TVA];
    int64_t bo[D], eo[D];                                         // offsets of the NEXT pass of each buffer, on their way
    auto issue = [&](int d, int jj, int64_t b, int64_t e) {       // every request of a pass, unconditionally
        if (jj >= nb) e = b;                                      // no such bucket: empty
        int64_t hb = (b + VN - 1) & ~(int64_t)(VN - 1);
        if (hb > e) hb = e;
        int64_t eb = e & ~(int64_t)(VN - 1);
        if (eb < hb) eb = hb;
        m[d].n = (int)(e - b); m[d].nh = (int)(hb - b); m[d].nt = (int)(e - eb);
        const int64_t vb = (b + VN - 1) / VN;                     // whole vectors [vb, ve) in 16-byte units from `values`
        int64_t ve = eb / VN;
        if (ve < vb) ve = vb;
        // a bucket that fits the window keeps the window at its first vector (nothing is left for the remainder loop); a longer one
        // starts it on the line grid, so that the remainder loop does
        const int64_t base = ve - vb <= G * NV ? vb : vb & ~(int64_t)(LV - 1);
        m[d].rvb = (int)(vb - base);
        m[d].rve = (int)(ve - base);                              // (a bucket is < 2^31 samples: n above)
        m[d].rtl = TAIL ? (int)((ve & ~(int64_t)(LV - 1)) - base) : m[d].rve;
        const bool has = ve > vb, mine = has && sub < m[d].rve;
        // a lane without a vector of its own requests the bucket's last one, a bucket without whole vectors the first element of
        // `values` (cache hits either way, and never past the end of the array)
        m[d].vp = mine ? reinterpret_cast<const V16*>(values) + base + sub
                       : has ? reinterpret_cast<const V16*>(values) + (ve - 1) : reinterpret_cast<const V16*>(values);
        const int last = mine ? (m[d].rve - 1 - sub) / G * G : 0;  // this lane's last vector of the bucket, as an index from vp
        // request order = consumption order, pinned (the scheduler would otherwise cluster the vector loads first and
        // the wait for K, the first value needed, would drain the whole pass)
        kraw[d] = values[m[d].n > 0 ? b : 0];
        xh[d] = values[sub < m[d].nh ? b + sub : 0];
        xt[d] = values[sub < m[d].nt ? eb + sub : 0];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#if DCARL_BOUNDS_NT
            { const uint4 w = nt_load16(m[d].vp + min(G * i, last)); x[d][i] = *reinterpret_cast<const V16*>(&w); }
#else
            x[d][i] = m[d].vp[min(G * i, last)];
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        const int t0 = max(m[d].rtl, G * NV);
#pragma unroll
        for (int k = 0; k < TV; ++k) {
            xl[d][k] = m[d].vp[t0 + G * k + sub < m[d].rve ? t0 + G * k : 0];   // (0: the window's own first request, not a line of the remainder loop)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // One code instance of every stage: the first D turns only issue (nothing to consume yet), so the loop header sees
    // the same request sequence from its preheader and from its back edge and the waits stay exact counts.
    // (The first offsets are waited for HERE, through a use the waitcnt pass can see: it merges the preheader's and
    // the back edge's pending requests conservatively, and two offsets loads pending one after the other at loop entry
    // would make it wait for "all but one" request where the steady state allows ten younger ones to stay in flight.)
#pragma unroll
    for (int d = 0; d < D; ++d) {
        bucket_range(d * BP + cl, bo[d], eo[d]);
        m[d] = Meta{0, 0, 0, 0, 0, 0, reinterpret_cast<const V16*>(values)};
        kraw[d] = T(0); xh[d] = T(0); xt[d] = T(0);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) asm volatile("" : "+v"(bo[d]), "+v"(eo[d]));    // a use: the compiler waits for the loads here
    for (int j0 = -D * BP; j0 < nb; j0 += D * BP) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int jp = j0 + d * BP;                           // first bucket of this stage's pass
            const int j = jp + cl;
            const bool live = jp >= 0 && jp < nb;                 // wave-uniform
            // ---- consume pass jp
            const double K = m[d].n > 0 ? (double)kraw[d] : 0.0;  // shift of the sums: the bucket's first sample
            const int n = m[d].n;
            double sm = 0.0, sq = 0.0;
            if (live) {
                if (sub < m[d].nh) { const double dd = (double)xh[d] - K; sm += dd; sq = fma(dd, dd, sq); }
                if (sub < m[d].nt) { const double dd = (double)xt[d] - K; sm += dd; sq = fma(dd, dd, sq); }
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (G * i + sub >= m[d].rvb && G * i + sub < m[d].rve) acc16(x[d][i], K, sm, sq);
                for (int v = G * NV; v + sub < m[d].rtl; v += RU * G) {  // long buckets, whole lines: RU more vectors in flight per turn
                    V16 y[RU];
                    bool h[RU];
#pragma unroll
                    for (int k = 0; k < RU; ++k) {
                        h[k] = v + k * G + sub < m[d].rtl;
                        if (h[k]) y[k] = m[d].vp[v + k * G];
                    }
#pragma unroll
                    for (int k = 0; k < RU; ++k)
                        if (h[k]) acc16(y[k], K, sm, sq);
                }
                {
                    const int t0 = max(m[d].rtl, G * NV);         // the last line, requested with the pass
#pragma unroll
                    for (int k = 0; k < TV; ++k)
                        if (t0 + G * k + sub < m[d].rve) acc16(xl[d][k], K, sm, sq);
                }
            }
            // ---- issue pass jp + D*BP into the registers just freed, then request the offsets of the pass after it
            issue(d, j + D * BP, bo[d], eo[d]);                   // (an empty pass beyond the block: cache hits on values[0])
            bucket_range(j + 2 * D * BP, bo[d], eo[d]);
            __builtin_amdgcn_sched_barrier(0);
            // ---- finish pass jp under those loads
            if (live) {
#pragma unroll
                for (int o = 1; o < G; o <<= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
                if (j < nb) {
                    const int st = (j * amul) >> 16;              // j / A for j < 512 (amul = 65536/A + 1)
                    const int a = j - st * A;
                    const bool is_rule = (a == p.rule_act);
                    double val = is_rule ? p.init_rule : p.init_other;                          // S1:50-53
                    const double vv = value_from_sums(max(n, 1), sm, sq, K, is_rule, p);        // S1:86-90
                    val = (n > p.n_thres) ? vv : val;
                    const double key = encode_key(val, a);
                    if (sub == 0) { kw[j] = key; nw[j] = n; }     // results stay in LDS until the block of states is done
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the keys were written by other lanes of this wavefront
    // the block's table leaves in ONE burst of full-width coalesced stores (16*A*12 bytes) instead of 128 + 64 bytes per
    // pass: result writes interleaved with the sample stream cost read bandwidth out of proportion to their size
    // (tools/experiments/ubench_stream.hip: 3 % of writes per pass -> -20 % read bandwidth; once per block -> -17 %)
    if (V_out)
        for (int i = lane; i < nb; i += WAVE) V_out[g0 + i] = strip_code(kw[i]);
    if (n_out)
        for (int i = lane; i < nb; i += WAVE) n_out[g0 + i] = nw[i];
    if (lane < ns) {
        double best = kw[lane * A];
        for (int a = 1; a < A; ++a) best = fmax(best, kw[lane * A + a]);                // S1:93-94
        if (vmax) vmax[s0 + lane] = (float)best;
        if (amax) amax[s0 + lane] = decode_action(best);
    }
}

// ---- few states, huge buckets: one BLOCK per state ---------------------------------------------------------------------------
// The kernel above gives a wavefront 16 states: a table of 20 states x 3e5 samples per bucket (the reference's own Simulation_2
// shape with a long log) would be streamed by two wavefronts (measured: 2^26 samples over 20 states x 11 actions, 38 ms).  Here
// a block of 256 threads owns a state, walks its A buckets one after the other with coalesced 16-byte loads (four in flight per
// thread), reduces the f64 partial sums through shuffles and LDS, and thread 0 evaluates the bucket; 0.4 ms for that table.
// launch_bounds_csr picks it when there are fewer states than the chip has wavefront slots to fill and the buckets are long.
template <typename T, bool CSR>
__global__ __launch_bounds__(256) void bounds_wide_kernel(const T* __restrict__ values, const int64_t* __restrict__ seg_off, int64_t n_dense,
                                                          int S, int A, DevParams p, double* __restrict__ V_out, int32_t* __restrict__ n_out,
                                                          float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using V16 = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    __shared__ double red[2][256 / WAVE];
    __shared__ double keys[DCARL_MAX_ACTIONS];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid >> 6;
    for (int a = 0; a < A; ++a) {
        const int64_t g = (int64_t)s * A + a;
        int64_t b, e;
        if (CSR) { b = seg_off[g]; e = seg_off[g + 1]; } else { b = g * n_dense; e = b + n_dense; }
        const int64_t n64 = e - b;
        const double K = n64 > 0 ? (double)values[b] : 0.0;       // shift of the sums: the bucket's first sample
        int64_t hb = (b + VN - 1) & ~(int64_t)(VN - 1);
        if (hb > e) hb = e;
        int64_t eb = e & ~(int64_t)(VN - 1);
        if (eb < hb) eb = hb;
        double sm = 0.0, sq = 0.0;
        if (b + tid < hb) { const double d = (double)values[b + tid] - K; sm += d; sq = fma(d, d, sq); }      // head (< VN samples)
        if (eb + tid < e) { const double d = (double)values[eb + tid] - K; sm += d; sq = fma(d, d, sq); }      // tail
        const V16* vp = reinterpret_cast<const V16*>(values + hb);
        const int64_t nvec = (eb - hb) / VN;
        int64_t v = tid;
        for (; v + 3 * 256 < nvec; v += 4 * 256) {
            const V16 y0 = vp[v], y1 = vp[v + 256], y2 = vp[v + 512], y3 = vp[v + 768];
            acc16(y0, K, sm, sq); acc16(y1, K, sm, sq); acc16(y2, K, sm, sq); acc16(y3, K, sm, sq);
        }
        for (; v < nvec; v += 256) acc16(vp[v], K, sm, sq);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
        __syncthreads();                                          // (the previous bucket's partials have been read)
        if (lane == 0) { red[0][wv] = sm; red[1][wv] = sq; }
        __syncthreads();
        if (tid == 0) {
            double ts = 0.0, tq = 0.0;
            for (int w = 0; w < 256 / WAVE; ++w) { ts += red[0][w]; tq += red[1][w]; }
            const int n = (int)(n64 > 0x7fffffff ? 0x7fffffff : n64);
            const bool is_rule = (a == p.rule_act);
            double val = is_rule ? p.init_rule : p.init_other;                              // S1:50-53
            const double vv = value_from_sums(n > 1 ? n : 1, ts, tq, K, is_rule, p);       // S1:86-90
            val = (n > p.n_thres) ? vv : val;
            const double key = encode_key(val, a);
            keys[a] = key;
            if (V_out) V_out[g] = strip_code(key);
            if (n_out) n_out[g] = n;
        }
    }
    if (tid == 0) {
        double best = keys[0];
        for (int a = 1; a < A; ++a) best = fmax(best, keys[a]);                             // S1:93-94
        if (vmax) vmax[s] = (float)best;
        if (amax) amax[s] = decode_action(best);
    }
}

// The four bound functions themselves (S1:10-28), one wavefront per bucket: out[b] = {upper_bound,
// lower_bound, CI_lower_bound, mean_value} of values[off[b] .. off[b+1]).  Backs the drop-in Python functions.
template <typename T>
__global__ __launch_bounds__(256) void bucket_bounds_kernel(const T* __restrict__ values,
                                                            const int64_t* __restrict__ off, int64_t B,
                                                            DevParams p, double* __restrict__ out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t bkt = (int64_t)blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    if (bkt >= B) return;
    const int64_t b = off[bkt], e = off[bkt + 1];
    double sm = 0.0, sq = 0.0;
    const double K = (e > b) ? (double)values[b] : 0.0;
    for (int64_t i = b + lane; i < e; i += WAVE) { double x = (double)values[i] - K; sm += x; sq = fma(x, x, sq); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
    if (lane == 0 && e > b) {
        const Bounds r = bounds_from_sums((int)(e - b), sm, sq, K, p);
        reinterpret_cast<double4*>(out)[bkt] = make_double4(r.upper, r.lower, r.ci_lower, fmin(p.cap, r.mean));
    }
}

template <typename T>
int launch_bucket_bounds(const T* values, const int64_t* off, int64_t B, const DevParams& p, double* out,
                         hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL((bucket_bounds_kernel<T>), dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, values, off, B, p,
                       out);
    return 0;
}
template int launch_bucket_bounds<float>(const float*, const int64_t*, int64_t, const DevParams&, double*, hipStream_t);
template int launch_bucket_bounds<double>(const double*, const int64_t*, int64_t, const DevParams&, double*,
                                          hipStream_t);

// One kernel family.  Default instance by the expected bucket size: 4 lanes per bucket up to ~256 samples (more buckets
// in flight per wavefront), 8 lanes beyond (longer contiguous runs per request); 2 register buffers; 4 vector slots per
// lane, 6 when a 4-lane cluster's bucket is expected to hold more than 16 vectors (configs[3], 91 samples per bucket: the
// non-pipelined remainder loop then rarely runs, 0.907 -> 0.886 ms; configs[4], 64 per bucket, keeps 4: 0.387 vs 0.404 ms).
// DCARL_QUAD=G,NV,D picks another compiled instance (same-box A/B measurements, tests of every instance).
template <typename T>
int launch_bounds_csr(const T* values, const int64_t* seg_off, int64_t n_dense, int64_t n_mean, int S, int A,
                      const DevParams& p, double* V_out, int32_t* n_out, float* vmax, int32_t* amax,
                      hipStream_t st) {
    if (S == 0) return 0;
    dim3 grid((unsigned)slices_of(S)), block(256);                         // a wavefront = 16 states, a block = 64
    constexpr int VN = Vec16<T>::N;
    int g = n_mean >= 64 * VN ? 8 : 4, dd = 2;
    int nv = (g == 4 && n_mean > 16 * VN) ? 6 : 4;               // all of a bucket's vectors in the pipelined slots: 16 or 24 per 4-lane cluster
    // fewer states than wavefront slots (a wavefront of the kernel above takes 16 states) and long buckets: a block per state
    bool wide = S < 16 * 1024 && n_mean >= 256 * VN;
    if (const char* e = DCARL_KNOB("DCARL_QUAD")) { sscanf(e, "%d,%d,%d", &g, &nv, &dd); wide = g == 0; }
    if (wide) {
        if (seg_off) hipLaunchKernelGGL((bounds_wide_kernel<T, true>), dim3(S), dim3(256), 0, st, values, seg_off, n_dense, S, A, p, V_out, n_out, vmax, amax);
        else hipLaunchKernelGGL((bounds_wide_kernel<T, false>), dim3(S), dim3(256), 0, st, values, seg_off, n_dense, S, A, p, V_out, n_out, vmax, amax);
        note_kernel("bounds_wide_kernel<%s,%s>", sizeof(T) == 4 ? "float" : "double", seg_off ? "csr" : "dense");
        return 0;
    }
#define DCARL_QUAD_CASE(GG, NN, DD)                                                                                   \
    if (g == GG && nv == NN && dd == DD) {                                                                            \
        if (seg_off)                                                                                                  \
            hipLaunchKernelGGL((bounds_quad_kernel<T, GG, NN, true, DD>), grid, block, 0, st, values, seg_off,        \
                               n_dense, S, A, 65536 / A + 1, p, V_out, n_out, vmax, amax);                            \
        else                                                                                                          \
            hipLaunchKernelGGL((bounds_quad_kernel<T, GG, NN, false, DD>), grid, block, 0, st, values, seg_off,       \
                               n_dense, S, A, 65536 / A + 1, p, V_out, n_out, vmax, amax);                            \
        note_kernel("bounds_quad_kernel<%s,%d,%d,%s,%d>", sizeof(T) == 4 ? "float" : "double", GG, NN,                \
                    seg_off ? "csr" : "dense", DD);                                                                   \
        return 0;                                                                                                     \
    }
    DCARL_QUAD_CASE(4, 4, 2) DCARL_QUAD_CASE(8, 4, 2) DCARL_QUAD_CASE(4, 4, 1) DCARL_QUAD_CASE(4, 6, 3) DCARL_QUAD_CASE(16, 4, 2)
    DCARL_QUAD_CASE(4, 6, 2)
    g = n_mean >= 64 * VN ? 8 : 4; dd = 2;                       // unknown instance requested: the default
    nv = (g == 4 && n_mean > 16 * VN) ? 6 : 4;
    DCARL_QUAD_CASE(4, 4, 2) DCARL_QUAD_CASE(8, 4, 2) DCARL_QUAD_CASE(4, 6, 2)
#undef DCARL_QUAD_CASE
    return 0;
}

template int launch_bounds_csr<float>(const float*, const int64_t*, int64_t, int64_t, int, int, const DevParams&,
                                      double*, int32_t*, float*, int32_t*, hipStream_t);
template int launch_bounds_csr<double>(const double*, const int64_t*, int64_t, int64_t, int, int, const DevParams&,
                                       double*, int32_t*, float*, int32_t*, hipStream_t);

}  // namespace dcarl
