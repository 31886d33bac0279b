// Online confidence estimation ("trace" mode) as a PRODUCER / CONSUMER pair of wavefronts per 64-state slice.
// Same results as trace.hip (S1:73-99 / S2:72-97), different mapping to the CU:
//
//   wave 0 (producer)  loads the records, runs the statistics stage (S1:80) and the f64 evaluation (S1:87-90) and
//                      posts, per record, the tie-break-coded value and the LDS address of the key it replaces
//   wave 1 (consumer)  overwrites the key (S1:86), reloads the candidate keys, arg-max tree, latch (S1:93-99) and
//                      stores the step traces
//
// The per-state loop is sequential, so a slice cannot be split over time; splitting it by STAGE instead doubles the
// resident wavefronts for a given number of states.  That matters because the headline table (65 536 states = 1024
// slices) gives the single-wave kernel exactly one wavefront per SIMD: a lone wavefront issues a VALU instruction
// every ~5 cycles instead of 4 and has nothing to run while it waits on LDS.  The two waves hand over through a
// two-entry LDS queue with one s_barrier per quad (4 records); the producer's HBM prefetch ring stays in flight
// across barriers (the barrier waits on lgkmcnt only).
#include <cstdlib>
#include <type_traits>

#include "trace_common.h"

namespace dcarl {

// lgkmcnt(0): the queue writes (producer) / reads (consumer) of this quad are complete; vmcnt is left alone.
#ifdef DCARL_PAIR_PROFILE
#define PAIR_BARRIER()                                                           \
    do {                                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        const unsigned long long t1 = __builtin_readcyclecounter();              \
        asm volatile("s_barrier" ::: "memory");                                  \
        const unsigned long long t2 = __builtin_readcyclecounter();              \
        prof_lgkm += t1 - t0;                                                    \
        prof_bar += t2 - t1;                                                     \
    } while (0)
#else
#define PAIR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

template <typename T, int NA>
__global__ __launch_bounds__(2 * WAVE) void trace_pair_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* __restrict__ act_step, double* __restrict__ V_out,
    int32_t* __restrict__ n_out, float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using Q4 = typename Quad<T>::type;
    constexpr int PF = 8;                                // prefetch ring depth in quads (32 records ahead)
    constexpr int NK = NA + 1;                           // candidate keys + one trash slot (bucket below threshold)
    __shared__ SumPair lds_sum[NA][WAVE];                // producer-private
    __shared__ int lds_cnt[NA][WAVE];                    // producer-private
    __shared__ double lds_key[NK][WAVE];                 // consumer-private: V[s][.] as tie-break-coded keys
    __shared__ double q_key[2][4][WAVE];                 // queue: coded value of each record of a quad
    __shared__ int q_off[2][4][WAVE];                    //        byte offset into lds_key of the key it replaces

    const int lane = threadIdx.x & (WAVE - 1);
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x;
    const int s = w * WAVE + lane;
    const int64_t row0 = slice_row_off[w];
    const int rows = (int)(slice_row_off[w + 1] - row0);
    const int my_len = (s < S) ? min(len[s], rows) : 0;

    int max_len = my_len, min_len = my_len;              // wave-uniform loop bounds (kept in SGPRs)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        max_len = max(max_len, __shfl_xor(max_len, off));
        min_len = min(min_len, __shfl_xor(min_len, off));
    }
    max_len = __builtin_amdgcn_readfirstlane(max_len);
    min_len = __builtin_amdgcn_readfirstlane(min_len);
    const int nquads = (max_len + 3) >> 2;
    const int lane_off = lane * (int)sizeof(double);
    const int trash_off = NA * WAVE * (int)sizeof(double) + lane_off;

#ifdef DCARL_PAIR_PROFILE
    unsigned long long prof_lgkm = 0, prof_bar = 0;
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
    if (role == 0) {
        // ================================ producer ================================================================
#pragma unroll
        for (int a = 0; a < NA; ++a) { lds_sum[a][lane] = SumPair{0.0, 0.0}; lds_cnt[a][lane] = 0; }
        const Q4* Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
        const uchar4* Aq = reinterpret_cast<const uchar4*>(act) + row0 / 4 * WAVE + lane;
        const double shift = (my_len > 0) ? (double)R[(row0 * WAVE) + lane * 4] : 0.0;
        const int nfast = (min_len >> 2) / PF * PF;      // quads (whole ring turns) in which every lane is live

        Q4 rbuf[PF];
        uchar4 abuf[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if (i < nfast) { rbuf[i] = Rq[(int64_t)i * WAVE]; abuf[i] = Aq[(int64_t)i * WAVE]; }
        PairRaw pa, pb;
        QuadStat cur, nxt;
        // post one evaluated record into queue entry qs
        auto post = [&](int qs, int j, int a, int n, double v) {
            q_key[qs][j][lane] = encode_key(v, a);
            q_off[qs][j][lane] = (n > p.n_thres) ? a * (WAVE * (int)sizeof(double)) + lane_off : trash_off;
        };
        //   Aa1(q+1) | B(q) records 0,1 | Aa2(q+1), Ab1(q+1) | B(q) records 2,3 | Ab2(q+1) | post(q) | barrier
        auto step = [&](int qi, auto slot, auto refill_c, auto more_c) {
            constexpr int i = decltype(slot)::value;
            constexpr bool REFILL = decltype(refill_c)::value, MORE = decltype(more_c)::value;
            constexpr int in = (i + 1) % PF;
            if (REFILL) { rbuf[i] = Rq[(int64_t)(qi + PF) * WAVE]; abuf[i] = Aq[(int64_t)(qi + PF) * WAVE]; }
            if (MORE) pair_read<NA>(pa, shift, lds_sum, lds_cnt, lane, abuf[in].x, abuf[in].y, (double)rbuf[in].x,
                                    (double)rbuf[in].y);
            double v[4];
            v[0] = value_from_sums(cur.n[0], cur.s[0], cur.q[0], shift, cur.a[0] == p.rule_act, p);
            v[1] = value_from_sums(cur.n[1], cur.s[1], cur.q[1], shift, cur.a[1] == p.rule_act, p);
            if (MORE) {
                pair_update(nxt, 0, pa, lds_sum, lds_cnt, lane);
                pair_read<NA>(pb, shift, lds_sum, lds_cnt, lane, abuf[in].z, abuf[in].w, (double)rbuf[in].z,
                              (double)rbuf[in].w);
            }
            v[2] = value_from_sums(cur.n[2], cur.s[2], cur.q[2], shift, cur.a[2] == p.rule_act, p);
            v[3] = value_from_sums(cur.n[3], cur.s[3], cur.q[3], shift, cur.a[3] == p.rule_act, p);
            if (MORE) pair_update(nxt, 2, pb, lds_sum, lds_cnt, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) post(i & 1, j, cur.a[j], cur.n[j], v[j]);
            if (MORE) cur = nxt;
            PAIR_BARRIER();
        };
        using std::integral_constant;
        using T_ = integral_constant<bool, true>;
        using F_ = integral_constant<bool, false>;
        int qb = 0;
        if (nfast > 0) {
            pair_read<NA>(pa, shift, lds_sum, lds_cnt, lane, abuf[0].x, abuf[0].y, (double)rbuf[0].x, (double)rbuf[0].y);
            pair_update(cur, 0, pa, lds_sum, lds_cnt, lane);
            pair_read<NA>(pb, shift, lds_sum, lds_cnt, lane, abuf[0].z, abuf[0].w, (double)rbuf[0].z, (double)rbuf[0].w);
            pair_update(cur, 2, pb, lds_sum, lds_cnt, lane);
            for (; qb < nfast - PF; qb += PF) {
                step(qb + 0, integral_constant<int, 0>{}, T_{}, T_{});
                step(qb + 1, integral_constant<int, 1>{}, T_{}, T_{});
                step(qb + 2, integral_constant<int, 2>{}, T_{}, T_{});
                step(qb + 3, integral_constant<int, 3>{}, T_{}, T_{});
                step(qb + 4, integral_constant<int, 4>{}, T_{}, T_{});
                step(qb + 5, integral_constant<int, 5>{}, T_{}, T_{});
                step(qb + 6, integral_constant<int, 6>{}, T_{}, T_{});
                step(qb + 7, integral_constant<int, 7>{}, T_{}, T_{});
            }
            step(qb + 0, integral_constant<int, 0>{}, F_{}, T_{});
            step(qb + 1, integral_constant<int, 1>{}, F_{}, T_{});
            step(qb + 2, integral_constant<int, 2>{}, F_{}, T_{});
            step(qb + 3, integral_constant<int, 3>{}, F_{}, T_{});
            step(qb + 4, integral_constant<int, 4>{}, F_{}, T_{});
            step(qb + 5, integral_constant<int, 5>{}, F_{}, T_{});
            step(qb + 6, integral_constant<int, 6>{}, F_{}, T_{});
            step(qb + 7, integral_constant<int, 7>{}, F_{}, F_{});
            qb += PF;
        }
        // tail: ragged ends of the slice, per-lane guards; records past a lane's end are posted to the trash slot
        for (int qi = qb; qi < nquads; ++qi) {
            const int qs = qi & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { q_key[qs][j][lane] = 0.0; q_off[qs][j][lane] = trash_off; }
            if (qi * 4 < my_len) {
                const Q4 rv = Rq[(int64_t)qi * WAVE];
                const uchar4 av = Aq[(int64_t)qi * WAVE];
                const double xr[4] = {(double)rv.x, (double)rv.y, (double)rv.z, (double)rv.w};
                const int aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (qi * 4 + j < my_len) {
                        const int a = min(aa[j], NA - 1);
                        const double x = xr[j] - shift;
                        SumPair sp = lds_sum[a][lane];
                        const int n = lds_cnt[a][lane] + 1;
                        sp.s += x;
                        sp.q = fma(x, x, sp.q);
                        lds_sum[a][lane] = sp;
                        lds_cnt[a][lane] = n;
                        post(qs, j, a, n, value_from_sums(n, sp.s, sp.q, shift, a == p.rule_act, p));
                    }
            }
            PAIR_BARRIER();
        }
        if (s < S && n_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) n_out[(int64_t)s * A + a] = lds_cnt[a][lane];
        }
#ifdef DCARL_PAIR_PROFILE
        if (lane == 0 && V_out && A >= 8) {               // producer: slots 0..3 of the slice's first state
            double* o = V_out + (int64_t)s * A;
            o[0] = (double)(__builtin_readcyclecounter() - prof_t0); o[1] = (double)prof_lgkm; o[2] = (double)prof_bar;
            o[3] = (double)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        }
#endif
    } else {
        // ================================ consumer ================================================================
        double best;
        {                                                 // S1:50-53 initial table, tie-break coded
            double key[NK];
#pragma unroll
            for (int a = 0; a < NK; ++a) {
                key[a] = (a < A) ? encode_key(a == p.rule_act ? p.init_rule : p.init_other, a) : encode_key(-1e300, a & 31);
                lds_key[a][lane] = key[a];
            }
            best = tree_max<NA>(key);
        }
        int latch = 0x7fffffff;
        Q4* SVq = step_val ? reinterpret_cast<Q4*>(step_val) + row0 / 4 * WAVE + lane : nullptr;
        uchar4* SAq = step_act ? reinterpret_cast<uchar4*>(step_act) + row0 / 4 * WAVE + lane : nullptr;
        char* key_base = reinterpret_cast<char*>(&lds_key[0][0]);

        auto consume = [&](int qi, int qs) {
            PAIR_BARRIER();                               // entry qs holds quad qi
            double k[4];
            int off[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { k[j] = q_key[qs][j][lane]; off[j] = q_off[qs][j][lane]; }
            double keys[4][NA];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // LDS executes in order: reload j sees the writes 0..j
                *reinterpret_cast<double*>(key_base + off[j]) = k[j];
#pragma unroll
                for (int a = 0; a < NA; ++a) keys[j][a] = lds_key[a][lane];
            }
            double ov[4];
            int oa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                best = tree_max<NA>(keys[j]);
                const int b = decode_action(best);
                ov[j] = best;
                oa[j] = b;
                latch = min(latch, (b != p.rule_act) ? qi * 4 + j + 1 : 0x7fffffff);
            }
            if (SVq) { Q4 o; o.x = (T)ov[0]; o.y = (T)ov[1]; o.z = (T)ov[2]; o.w = (T)ov[3]; SVq[(int64_t)qi * WAVE] = o; }
            if (SAq) SAq[(int64_t)qi * WAVE] = make_uchar4(oa[0], oa[1], oa[2], oa[3]);
        };
        int qi = 0;
        for (; qi + 1 < nquads; qi += 2) { consume(qi, 0); consume(qi + 1, 1); }
        if (qi < nquads) consume(qi, 0);

        if (s < S) {
            if (act_step) act_step[s] = latch == 0x7fffffff ? -1 : latch;
            if (vmax) vmax[s] = (float)best;
            if (amax) amax[s] = decode_action(best);
#ifndef DCARL_PAIR_PROFILE
            if (V_out) {
#pragma unroll
                for (int a = 0; a < NA; ++a) if (a < A) V_out[(int64_t)s * A + a] = strip_code(lds_key[a][lane]);
            }
#endif
        }
#ifdef DCARL_PAIR_PROFILE
        if (lane == 0 && V_out && A >= 8) {
            double* o = V_out + (int64_t)s * A + 4;
            o[0] = (double)(__builtin_readcyclecounter() - prof_t0); o[1] = (double)prof_lgkm; o[2] = (double)prof_bar;
            o[3] = (double)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        }
#endif
    }
}

template <typename T>
int launch_trace_pair(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int S, int A,
                      const DevParams& p, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                      int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st) {
    const int W = (S + WAVE - 1) / WAVE;
    if (W == 0) return 0;
    dim3 grid(W), block(2 * WAVE);
    const unsigned pad = getenv("DCARL_LDS_PAD") ? (unsigned)atoi(getenv("DCARL_LDS_PAD")) : 0u;
#define DCARL_CASE(NA)                                                                                           \
    case NA:                                                                                                     \
        hipLaunchKernelGGL((trace_pair_kernel<T, NA>), grid, block, pad, st, R, act, slice_row_off, len, S, A, p,  \
                           step_val, step_act, act_step, V_out, n_out, vmax, amax);                              \
        break
    const int na = A <= 16 ? A : (A <= 24 ? 24 : 32);
    switch (na) {
        DCARL_CASE(1); DCARL_CASE(2); DCARL_CASE(3); DCARL_CASE(4); DCARL_CASE(5); DCARL_CASE(6); DCARL_CASE(7);
        DCARL_CASE(8); DCARL_CASE(9); DCARL_CASE(10); DCARL_CASE(11); DCARL_CASE(12); DCARL_CASE(13);
        DCARL_CASE(14); DCARL_CASE(15); DCARL_CASE(16); DCARL_CASE(24); DCARL_CASE(32);
    }
#undef DCARL_CASE
    return 0;
}

template int launch_trace_pair<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                      const DevParams&, float*, uint8_t*, int32_t*, double*, int32_t*, float*,
                                      int32_t*, hipStream_t);
template int launch_trace_pair<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                       const DevParams&, double*, uint8_t*, int32_t*, double*, int32_t*, float*,
                                       int32_t*, hipStream_t);

}  // namespace dcarl
