// Online confidence estimation + candidate arg-max over all states ("trace" mode).
// Replaces the reference hot loop S1:73-99 / S2:72-97 (Simulation_testing/Simulation_{1,2}/test_DCARL.py).
//
// Mapping to CDNA4: one wavefront (64 lanes) = one slice of 64 states, one lane = one state; the lane walks
// its state's records in arrival order (the loop is sequential per state by definition: every record's
// arg-max depends on all earlier records of that state).  Per 4 records a lane issues ONE 16-byte load of R,
// one 4-byte load of the action ids, and one 16-byte + one 4-byte store of the step traces; a wavefront's
// accesses are 1 KiB / 256 B contiguous ("sliced time-major, quad-packed" layout, include/dcarl.h).
// Per-bucket sufficient statistics (n, shifted sum, shifted sum of squares; f64) live in LDS, indexed
// [action][lane] so that the per-lane dynamic action index never causes a bank conflict; the NA current values
// V[s][.] live in LDS too, as tie-break-coded f64 keys, so that the arg-max over candidates is a balanced tree of
// v_max_f64 over a reload.  While every lane of the wavefront still has records left the loop body is branch-free
// straight-line code, software-pipelined over quads (trace_common.h), so that LDS round trips and the HBM prefetch
// complete behind the f64 evaluation chains.  10 B of HBM traffic per record/evaluation (f32 storage); measured VALU-
// and LDS-bound.  This file is the COMPUTE kernel (both count roots evaluated per record): it
// serves 17..32 candidates and DCARL_TRACE_KERNEL=single; up to 16 candidates launch_trace prefers the multi-wave
// count-root table kernel (trace_nwave_impl.h: four wavefronts per slice for f32 storage, three for f64).
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "trace_common.h"

namespace dcarl {

template <typename T, int NA>
__global__ __launch_bounds__(WAVE) void trace_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* act_step, double* V_out,          // (a resumed launch reads these three through cy)
    int32_t* n_out, float* __restrict__ vmax, int32_t* __restrict__ amax, const TraceCarry cy) {
    using Q4 = typename Quad<T>::type;
    constexpr int PF = 8;                                // prefetch ring depth in quads (32 records ahead)
    constexpr int NP = key_cells<NA>();
    __shared__ __attribute__((aligned(16))) double lds_rows[2 * NA + 2 * NP][WAVE];   // sums, sums of squares, keys (trace_common.h)
    __shared__ int lds_cnt[NA][WAVE];
    LdsRow* lrows = lds_rows;
    KeyRow* lds_key = key_rows_of<NA>(lrows);

    const int lane = threadIdx.x;
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

    // a resumed loop (dcarl_trace_resume_*) starts from the caller's state instead of S1:41-59's priors
    const bool resumed = cy.n != nullptr && !cy.fresh;   // launch-uniform
    const int so = (s < S && slot_state) ? slot_state[s] : s;   // per-state rows are the STATE's, not the slot's
    const CarryIn cin = carry_in(cy, s < S, so, A);
    const bool from_state = resumed && s < S;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        SumPair sp{0.0, 0.0};
        int cn = 0;
        if (from_state && a < A) {
            sp = SumPair{cy.sum[(int64_t)so * A + a], cy.sumsq[(int64_t)so * A + a]};
            cn = cy.n[(int64_t)so * A + a];
        }
        lrows[a][lane] = sp.s;
        lrows[NA + a][lane] = sp.q;
        lds_cnt[a][lane] = cn;
    }

    const Q4* Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
    const uchar4* Aq = reinterpret_cast<const uchar4*>(act) + row0 / 4 * WAVE + lane;
    Q4* SVq = step_val ? reinterpret_cast<Q4*>(step_val) + row0 / 4 * WAVE + lane : nullptr;
    uchar4* SAq = step_act ? reinterpret_cast<uchar4*>(step_act) + row0 / 4 * WAVE + lane : nullptr;

    LaneState<NA> st;                                    // S1:50-53 initial table, tie-break coded
    {
        double key[2 * NP];
#pragma unroll
        for (int a = 0; a < 2 * NP; ++a) {
            double v0 = a == p.rule_act ? p.init_rule : p.init_other;
            if (from_state && a < A) v0 = cy.V[(int64_t)so * A + a];        // encode_key(strip_code(key)) == key: the same keys
            key[a] = (a < A) ? encode_key(v0, a) : encode_key(-1e300, a & 31);
        }
#pragma unroll
        for (int c = 0; c < 2 * NP; ++c) lds_key[c][lane] = key[c];
        st.best = tree_max<NA>(key);
    }
    st.latch = 0x7fffffff;
    const unsigned rule4 = (unsigned)p.rule_act * 0x01010101u;
    st.shift = (my_len > 0) ? (double)R[(row0 * WAVE) + lane * 4] : 0.0;
    if (resumed && cin.t_base > 0) st.shift = cy.shift[so];               // K stays the state's first reward of ALL launches

    const int nquads = (max_len + 3) >> 2;
    const int nfast = (min_len >> 2) / PF * PF;          // quads (whole ring turns) in which every lane is live

    // ---- main loop: every lane live; PF quads (16 records) per iteration, software-pipelined (see above) -------
    Q4 rbuf[PF];
    uchar4 abuf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < nfast) { rbuf[i] = Rq[(int64_t)i * WAVE]; abuf[i] = Aq[(int64_t)i * WAVE]; }
    PairRaw pa, pb;
    QuadStat cur, nxt;
    // one pipeline step for quad qi living in ring slot i.  REFILL / MORE are compile-time so that the steady-state
    // loop body is branch-free: the waitcnt pass can then count outstanding loads exactly (a conditional load makes
    // it fall back to vmcnt(0), which would expose the full HBM latency of the prefetch issued a moment earlier).
    //   Aa1(q+1) | B(q) | C1(q) | Aa2(q+1), Ab1(q+1) | C2(q) | Ab2(q+1)      (A at pair granularity, trace_common.h)
    auto step = [&](int qi, auto slot, auto refill_c, auto more_c) {
        constexpr int i = decltype(slot)::value;
        constexpr bool REFILL = decltype(refill_c)::value, MORE = decltype(more_c)::value;
        constexpr int in = (i + 1) % PF;                  // ring slot of quad qi+1 (refilled PF-1 quads ago)
        if (REFILL) { rbuf[i] = Rq[(int64_t)(qi + PF) * WAVE]; abuf[i] = Aq[(int64_t)(qi + PF) * WAVE]; }
        if (MORE) pair_read<NA>(pa, st.shift, lrows, lds_cnt, lane, abuf[in].x, abuf[in].y, (double)rbuf[in].x,
                                (double)rbuf[in].y);
        double v[4];                                      // B(qi)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = value_from_sums(cur.n[j], cur.s[j], cur.q[j], st.shift, cur.a[j] == p.rule_act, p);
        double k0[NA], k1[NA], k2[NA], k3[NA];            // C1(qi)
        commit_issue<NA>(k0, lds_key, lane, cur.a[0], cur.n[0], v[0], p);
        commit_issue<NA>(k1, lds_key, lane, cur.a[1], cur.n[1], v[1], p);
        commit_issue<NA>(k2, lds_key, lane, cur.a[2], cur.n[2], v[2], p);
        commit_issue<NA>(k3, lds_key, lane, cur.a[3], cur.n[3], v[3], p);
        if (MORE) {
            pair_update<NA>(nxt, 0, pa, lrows, lds_cnt, lane);
            pair_read<NA>(pb, st.shift, lrows, lds_cnt, lane, abuf[in].z, abuf[in].w, (double)rbuf[in].z,
                          (double)rbuf[in].w);
        }
        double ov[4];                                     // C2(qi)
        int oa[4];
        commit_finish<NA>(st, k0, ov[0], oa[0]);
        commit_finish<NA>(st, k1, ov[1], oa[1]);
        commit_finish<NA>(st, k2, ov[2], oa[2]);
        commit_finish<NA>(st, k3, ov[3], oa[3]);
        if (SVq) { Q4 o; o.x = step_out<T>(ov[0]); o.y = step_out<T>(ov[1]); o.z = step_out<T>(ov[2]); o.w = step_out<T>(ov[3]); SVq[(int64_t)qi * WAVE] = o; }
        const unsigned packed = (unsigned)oa[0] | ((unsigned)oa[1] << 8) | ((unsigned)oa[2] << 16) | ((unsigned)oa[3] << 24);
        latch_quad(st.latch, packed, rule4, qi * 4);
        if (SAq) *reinterpret_cast<unsigned*>(&SAq[(int64_t)qi * WAVE]) = packed;
        if (MORE) { pair_update<NA>(nxt, 2, pb, lrows, lds_cnt, lane); cur = nxt; }
    };
    using std::integral_constant;
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    int qb = 0;
    if (nfast > 0) {                                      // pipeline prologue: stage A of quad 0
        pair_read<NA>(pa, st.shift, lrows, lds_cnt, lane, abuf[0].x, abuf[0].y, (double)rbuf[0].x, (double)rbuf[0].y);
        pair_update<NA>(cur, 0, pa, lrows, lds_cnt, lane);
        pair_read<NA>(pb, st.shift, lrows, lds_cnt, lane, abuf[0].z, abuf[0].w, (double)rbuf[0].z, (double)rbuf[0].w);
        pair_update<NA>(cur, 2, pb, lrows, lds_cnt, lane);
        for (; qb < nfast - PF; qb += PF) {               // steady state: every refill and every next quad exists
            step(qb + 0, integral_constant<int, 0>{}, T_{}, T_{});
            step(qb + 1, integral_constant<int, 1>{}, T_{}, T_{});
            step(qb + 2, integral_constant<int, 2>{}, T_{}, T_{});
            step(qb + 3, integral_constant<int, 3>{}, T_{}, T_{});
            step(qb + 4, integral_constant<int, 4>{}, T_{}, T_{});
            step(qb + 5, integral_constant<int, 5>{}, T_{}, T_{});
            step(qb + 6, integral_constant<int, 6>{}, T_{}, T_{});
            step(qb + 7, integral_constant<int, 7>{}, T_{}, T_{});
        }
        step(qb + 0, integral_constant<int, 0>{}, F_{}, T_{});               // last ring turn: nothing left to prefetch
        step(qb + 1, integral_constant<int, 1>{}, F_{}, T_{});
        step(qb + 2, integral_constant<int, 2>{}, F_{}, T_{});
        step(qb + 3, integral_constant<int, 3>{}, F_{}, T_{});
        step(qb + 4, integral_constant<int, 4>{}, F_{}, T_{});
        step(qb + 5, integral_constant<int, 5>{}, F_{}, T_{});
        step(qb + 6, integral_constant<int, 6>{}, F_{}, T_{});
        step(qb + 7, integral_constant<int, 7>{}, F_{}, F_{});
        qb += PF;
    }
    // ---- tail: ragged ends of the slice, per-lane guards ------------------------------------------------------
    for (int qi = qb; qi < nquads; ++qi) {
        if (qi * 4 < my_len) {
            const Q4 rv = Rq[(int64_t)qi * WAVE];
            const uchar4 av = Aq[(int64_t)qi * WAVE];
            const double xr[4] = {(double)rv.x, (double)rv.y, (double)rv.z, (double)rv.w};
            const int aa[4] = {av.x, av.y, av.z, av.w};
            double ov[4] = {0.0, 0.0, 0.0, 0.0};
            int oa[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (qi * 4 + j < my_len)
                    guarded_record<NA>(st, lrows, lds_cnt, lane, aa[j], xr[j], qi * 4 + j, p, ov[j], oa[j]);
            if (SVq) { Q4 o; o.x = step_out<T>(ov[0]); o.y = step_out<T>(ov[1]); o.z = step_out<T>(ov[2]); o.w = step_out<T>(ov[3]); SVq[(int64_t)qi * WAVE] = o; }
            if (SAq) SAq[(int64_t)qi * WAVE] = make_uchar4(oa[0], oa[1], oa[2], oa[3]);
        }
    }

    if (s < S) {
        if (act_step) act_step[so] = carry_latch(cin, st.latch, LATCH_NEVER);
        if (cy.n != nullptr) {                            // the advanced sufficient statistic (V, n, latch: the outputs below)
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (a < A) { cy.sum[(int64_t)so * A + a] = lrows[a][lane]; cy.sumsq[(int64_t)so * A + a] = lrows[NA + a][lane]; }
            cy.shift[so] = st.shift;
        }
        if (vmax) vmax[so] = (float)st.best;
        if (amax) amax[so] = decode_action(st.best);
        if (V_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (a < A) V_out[(int64_t)so * A + a] = strip_code(lds_key[a][lane]);
        }
        if (n_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) n_out[(int64_t)so * A + a] = lds_cnt[a][lane];
        }
    }
}

// The one word of device memory the library owns (static, not allocated): a hand-over of the multi-wave kernel that never
// arrives sets it before the waiting wave ends (trace_nwave_impl.h, wait_for).  dcarl_trace_status() reads and clears it.
__device__ int g_trace_fault = 0;
int* trace_fault_word() {
    static int* addr = [] {
        void* p = nullptr;
        return hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace_fault)) == hipSuccess ? static_cast<int*>(p) : nullptr;
    }();
    return addr;
}
int trace_raise_fault() {
    int* w = trace_fault_word();
    const int one = 1;
    return (w && hipMemcpy(w, &one, sizeof(int), hipMemcpyHostToDevice) == hipSuccess) ? 0 : -1;
}
int trace_status(hipStream_t st) {
    int v = 0;
    int* w = trace_fault_word();
    if (!w) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (hipMemcpy(&v, w, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v) { const int zero = 0; (void)hipMemcpy(w, &zero, sizeof(int), hipMemcpyHostToDevice); }
    return v;
}

template <typename T>
int launch_final_table(const T*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&, double*, int32_t*,
                       float*, int32_t*, hipStream_t);
template <typename T>
bool launch_trace_nwave(const T*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const DevParams&, T*, uint8_t*,
                        int32_t*, double*, int32_t*, float*, int32_t*, hipStream_t, int waves_per_slice, const TraceCarry&);

// DCARL_TRACE_KERNEL=single|duo|trio|quad overrides the choice (A/B measurements, tests of every kernel; duo / trio / quad = two /
// three / four waves per slice); read per launch
static int trace_kernel_override() {
    const char* e = DCARL_KNOB("DCARL_TRACE_KERNEL");
    if (!e) return 0;
    return !strcmp(e, "single") ? 1 : !strcmp(e, "duo") ? 4 : !strcmp(e, "trio") ? 5 : !strcmp(e, "quad") ? 6 : 0;
}

template <typename T>
int launch_trace(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int S, int A,
                 const DevParams& p, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                 int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st, const TraceCarry& cy) {
    const int W = slices_of(S);
    if (W == 0) return 0;
    const int which = trace_kernel_override();
    // nobody asked for anything per record (no step traces, no activation latch) and nothing is carried: the final table needs
    // the statistics stage and one evaluation per bucket only (trace_final.hip).  DCARL_FINAL_TABLE=0: the online kernel anyway
    // (the equivalence test, A/B runs)
    if (!step_val && !step_act && !act_step && cy.n == nullptr && which == 0) {
        const char* e = DCARL_KNOB("DCARL_FINAL_TABLE");
        if (!(e && e[0] == '0')) return launch_final_table<T>(R, act, slice_row_off, len, slot_state, S, A, p, V_out, n_out, vmax, amax, st);
    }
    // default: four (f32) / three (f64) waves per slice on round-robin quads sharing the count-root table (A <= 16),
    // else the one-wave compute kernel below
    if (which != 1 && launch_trace_nwave<T>(R, act, slice_row_off, len, slot_state, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax,
                                            amax, st, which == 4 ? 2 : which == 5 ? 3 : which == 6 ? 4 : 0, cy))
        return 0;
    dim3 grid(W), block(WAVE);
#define DCARL_CASE(NA)                                                                                           \
    case NA:                                                                                                     \
        hipLaunchKernelGGL((trace_kernel<T, NA>), grid, block, 0, st, R, act, slice_row_off, len, slot_state, S, A, p, step_val, \
                           step_act, act_step, V_out, n_out, vmax, amax, cy);                                    \
        note_kernel("trace_kernel<%s,%d>", sizeof(T) == 4 ? "float" : "double", NA);                            \
        break
    // the number of key registers / LDS rows is the exact candidate count up to 16, then 24 / 32
    const int na = A <= 16 ? A : (A <= 24 ? 24 : 32);
    switch (na) {
#ifdef DCARL_AB_BUILD                                    // (A <= 16 reaches this kernel only through DCARL_TRACE_KERNEL=single)
        DCARL_CASE(1); DCARL_CASE(2); DCARL_CASE(3); DCARL_CASE(4); DCARL_CASE(5); DCARL_CASE(6); DCARL_CASE(7);
        DCARL_CASE(8); DCARL_CASE(9); DCARL_CASE(10); DCARL_CASE(11); DCARL_CASE(12); DCARL_CASE(13);
        DCARL_CASE(14); DCARL_CASE(15); DCARL_CASE(16);
#endif
        DCARL_CASE(24); DCARL_CASE(32);
    }
#undef DCARL_CASE
    return 0;
}

template int launch_trace<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int,
                                 const DevParams&, float*, uint8_t*, int32_t*, double*, int32_t*, float*, int32_t*,
                                 hipStream_t, const TraceCarry&);
template int launch_trace<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int,
                                  const DevParams&, double*, uint8_t*, int32_t*, double*, int32_t*, float*,
                                  int32_t*, hipStream_t, const TraceCarry&);

}  // namespace dcarl
