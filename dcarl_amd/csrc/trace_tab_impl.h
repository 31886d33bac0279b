// Online confidence estimation ("trace" mode) with the two functions of the bucket COUNT, r = 1/sqrt(n) and
// rho = 2/sqrt(n+1), read from a table in LDS instead of being recomputed for every record.
// Same arithmetic and results as trace.hip (S1:73-99 / S2:72-97): the table is filled with the very function the
// compute path evaluates (count_roots, common.h), so both paths produce bit-identical values and a wavefront may
// switch between them at any ring turn.
//
// Why: the kernel is VALU-throughput bound (trace.hip) and the two reciprocal square roots of
// the count are 2 x (v_cvt, quarter-rate v_rsq_f32, v_cvt, 5 f64 operations) + 3 = ~25 issue slots of the ~95 per
// record, yet they depend on an integer that is < 4096 for every table this path is quoted on.  One ds_read_b128
// issued a pipeline stage ahead replaces them.  A workgroup is WPB independent wavefronts (one 64-state slice each,
// no barrier after the fill) that share one 64 KiB table; LDS = 64 KiB + WPB x (statistics + keys), one workgroup per
// CU.  Every four ring turns (128 records) a wavefront checks max_a count[a] + 128 < TAB_N over its lanes and otherwise
// runs those turns on the compute path, so tables with longer buckets stay exact.
//
// Included by trace_tab_f32.hip / trace_tab_f64.hip, which define DCARL_TAB_T (one translation unit per storage type:
// 32 kernel instances each).
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "trace_common.h"

namespace dcarl {

constexpr int TAB_N = 4096;                                  // counts 0 .. TAB_N-1
struct __attribute__((aligned(16))) RootPair { double r, rho; };
struct QuadRoots { double r[4], rho[4]; };

template <class F, int... I>
__device__ __forceinline__ void for_each_slot(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

__device__ __forceinline__ uchar4 as_uchar4(unsigned v) { return make_uchar4(v & 255u, (v >> 8) & 255u, (v >> 16) & 255u, v >> 24); }

template <int NA> constexpr int tab_wave_bytes() { return NA * WAVE * 20 + key_cells<NA>() * WAVE * 16; }
template <int NA> constexpr int tab_waves_per_block() { return NA <= 13 ? 4 : 3; }

template <typename T, int NA, bool STEPS>
__global__ __launch_bounds__(tab_waves_per_block<NA>() * WAVE) void trace_tab_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* __restrict__ act_step, double* __restrict__ V_out,
    int32_t* __restrict__ n_out, float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using Q4 = typename Quad<T>::type;
    constexpr int PF = 8;                                // prefetch ring depth in quads (32 records ahead); even
    constexpr int NP = key_cells<NA>();
    constexpr int WPB = tab_waves_per_block<NA>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    RootPair* tab = reinterpret_cast<RootPair*>(smem);

    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = (S + WAVE - 1) / WAVE;
    const int w = blockIdx.x * WPB + wid;

    {   // fill the table as far as this workgroup's longest slice can count
        int64_t need = 0;
        for (int i = 0; i < WPB; ++i) {
            const int wi = min(blockIdx.x * WPB + i, W - 1);
            need = max(need, slice_row_off[wi + 1] - slice_row_off[wi]);
        }
        const int fill = (int)min((int64_t)TAB_N, need + 2);
        for (int i = threadIdx.x; i < fill; i += WPB * WAVE) {
            const CountRoots c = count_roots(max(i, 1));
            tab[i] = RootPair{c.r, c.rho};
        }
    }
    __syncthreads();
    if (w >= W) return;                                  // the wavefronts are independent from here on

    unsigned char* mine = smem + TAB_N * sizeof(RootPair) + wid * tab_wave_bytes<NA>();
    SumPair (*lds_sum)[WAVE] = reinterpret_cast<SumPair (*)[WAVE]>(mine);
    KeyPair (*lds_key)[WAVE] = reinterpret_cast<KeyPair (*)[WAVE]>(mine + NA * WAVE * 16);
    int (*lds_cnt)[WAVE] = reinterpret_cast<int (*)[WAVE]>(mine + (NA + NP) * WAVE * 16);

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

#pragma unroll
    for (int a = 0; a < NA; ++a) { lds_sum[a][lane] = SumPair{0.0, 0.0}; lds_cnt[a][lane] = 0; }

    // wave-uniform bases (SGPR pairs) + a 32-bit per-lane byte offset: the global accesses of the fast path then use
    // the scalar-base addressing mode and need no 64-bit VALU address arithmetic
    const Q4* Rw = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE;
    const unsigned* Aw = reinterpret_cast<const unsigned*>(act) + row0 / 4 * WAVE;
    Q4* SVw = reinterpret_cast<Q4*>(step_val) + row0 / 4 * WAVE;
    unsigned* SAw = reinterpret_cast<unsigned*>(step_act) + row0 / 4 * WAVE;
    auto at_lane = [lane](auto* base) -> decltype(*base)& {
        using E = std::remove_reference_t<decltype(*base)>;
        using B = std::conditional_t<std::is_const<E>::value, const unsigned char, unsigned char>;
        return *reinterpret_cast<E*>(reinterpret_cast<B*>(base) + (unsigned)(lane * (int)sizeof(E)));
    };
    const Q4* Rq = Rw + lane;                            // per-lane pointers for the guarded tail
    const uchar4* Aq = reinterpret_cast<const uchar4*>(Aw) + lane;
    Q4* SVq = SVw + lane;
    uchar4* SAq = reinterpret_cast<uchar4*>(SAw) + lane;
    const bool has_sv = STEPS || step_val != nullptr, has_sa = STEPS || step_act != nullptr;   // wave-uniform

    LaneState<NA> st;                                    // S1:50-53 initial table, tie-break coded
    {
        double key[2 * NP];
#pragma unroll
        for (int a = 0; a < 2 * NP; ++a)
            key[a] = (a < A) ? encode_key(a == p.rule_act ? p.init_rule : p.init_other, a) : encode_key(-1e300, a & 31);
#pragma unroll
        for (int c = 0; c < NP; ++c) lds_key[c][lane] = KeyPair{key[2 * c], key[2 * c + 1]};
        st.best = tree_max<NA>(key);
    }
    st.latch = 0x7fffffff;
    const unsigned rule4 = (unsigned)p.rule_act * 0x01010101u;
    st.shift = (my_len > 0) ? (double)R[(row0 * WAVE) + lane * 4] : 0.0;

    const int nquads = (max_len + 3) >> 2;
    const int nfast = (min_len >> 2) / PF * PF;          // quads (whole ring turns) in which every lane is live

    Q4 rbuf[PF];
    uchar4 abuf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (i < nfast) { rbuf[i] = at_lane(Rw + (int64_t)i * WAVE); abuf[i] = as_uchar4(at_lane(Aw + (int64_t)i * WAVE)); }
    PairRaw pa, pb;
    QuadStat qstat[2];                                   // quad in ring slot i uses entry i & 1: no copies between steps
    QuadRoots qroot[2];
    auto roots_read = [&](QuadRoots& o, const QuadStat& q, int j0) {     // counts are < TAB_N here (see table_safe)
        const RootPair t0 = tab[q.n[j0]], t1 = tab[q.n[j0 + 1]];
        o.r[j0] = t0.r; o.rho[j0] = t0.rho; o.r[j0 + 1] = t1.r; o.rho[j0 + 1] = t1.rho;
    };
    // every bucket of every lane stays below TAB_N for the next CHECK_TURNS ring turns (4*PF records per lane each)
    constexpr int CHECK_TURNS = 4;
    auto table_safe = [&]() {
        int m = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a) m = max(m, lds_cnt[a][lane]);
        return __all(m + CHECK_TURNS * 4 * PF < TAB_N) != 0;
    };
    // one pipeline step for quad qi living in ring slot i (compile-time flags keep the steady state branch-free):
    //   Aa1(q+1) | B(q) | C1(q) | Aa2(q+1), roots(q+1), Ab1(q+1) | C2(q) | Ab2(q+1), roots(q+1)
    auto step = [&](int qi, auto slot, auto refill_c, auto more_c, auto tab_c) {
        constexpr int i = decltype(slot)::value;
        constexpr bool REFILL = decltype(refill_c)::value, MORE = decltype(more_c)::value, TAB = decltype(tab_c)::value;
        constexpr int in = (i + 1) % PF;                  // ring slot of quad qi+1 (refilled PF-1 quads ago)
        QuadStat& cur = qstat[i & 1];
        QuadStat& nxt = qstat[in & 1];
        QuadRoots& crt = qroot[i & 1];
        QuadRoots& nrt = qroot[in & 1];
        if (REFILL) {
            rbuf[i] = at_lane(Rw + (int64_t)(qi + PF) * WAVE);
            abuf[i] = as_uchar4(at_lane(Aw + (int64_t)(qi + PF) * WAVE));
        }
        if (MORE) pair_read<NA>(pa, st.shift, lds_sum, lds_cnt, lane, abuf[in].x, abuf[in].y, (double)rbuf[in].x,
                                (double)rbuf[in].y);
        double v[4];                                      // B(qi)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = TAB ? value_from_roots(crt.r[j], crt.rho[j], cur.s[j], cur.q[j], st.shift, cur.a[j] == p.rule_act, p)
                       : value_from_sums(cur.n[j], cur.s[j], cur.q[j], st.shift, cur.a[j] == p.rule_act, p);
        double k0[NA], k1[NA], k2[NA], k3[NA];            // C1(qi)
        commit_issue<NA>(k0, lds_key, lane, cur.a[0], cur.n[0], v[0], p);
        commit_issue<NA>(k1, lds_key, lane, cur.a[1], cur.n[1], v[1], p);
        commit_issue<NA>(k2, lds_key, lane, cur.a[2], cur.n[2], v[2], p);
        commit_issue<NA>(k3, lds_key, lane, cur.a[3], cur.n[3], v[3], p);
        if (MORE) {
            pair_update(nxt, 0, pa, lds_sum, lds_cnt, lane);
            if (TAB) roots_read(nrt, nxt, 0);
            pair_read<NA>(pb, st.shift, lds_sum, lds_cnt, lane, abuf[in].z, abuf[in].w, (double)rbuf[in].z,
                          (double)rbuf[in].w);
        }
        double ov[4];                                     // C2(qi)
        int oa[4];
        commit_finish<NA>(st, k0, ov[0], oa[0]);
        commit_finish<NA>(st, k1, ov[1], oa[1]);
        commit_finish<NA>(st, k2, ov[2], oa[2]);
        commit_finish<NA>(st, k3, ov[3], oa[3]);
        const unsigned packed = (unsigned)oa[0] | ((unsigned)oa[1] << 8) | ((unsigned)oa[2] << 16) | ((unsigned)oa[3] << 24);
        latch_quad(st.latch, packed, rule4, qi * 4);
        if (has_sv) { Q4 o; o.x = (T)ov[0]; o.y = (T)ov[1]; o.z = (T)ov[2]; o.w = (T)ov[3]; at_lane(SVw + (int64_t)qi * WAVE) = o; }
        if (has_sa) at_lane(SAw + (int64_t)qi * WAVE) = packed;
        if (MORE) {
            pair_update(nxt, 2, pb, lds_sum, lds_cnt, lane);
            if (TAB) roots_read(nrt, nxt, 2);
        }
    };
    using std::integral_constant;
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    int qb = 0;
    if (nfast > 0) {                                      // pipeline prologue: stage A of quad 0
        pair_read<NA>(pa, st.shift, lds_sum, lds_cnt, lane, abuf[0].x, abuf[0].y, (double)rbuf[0].x, (double)rbuf[0].y);
        pair_update(qstat[0], 0, pa, lds_sum, lds_cnt, lane);
        pair_read<NA>(pb, st.shift, lds_sum, lds_cnt, lane, abuf[0].z, abuf[0].w, (double)rbuf[0].z, (double)rbuf[0].w);
        pair_update(qstat[0], 2, pb, lds_sum, lds_cnt, lane);
        roots_read(qroot[0], qstat[0], 0);
        roots_read(qroot[0], qstat[0], 2);
        auto turn = [&](auto refill_c, auto tab_c) {      // one ring turn = PF pipeline steps
            for_each_slot([&](auto slot) { step(qb + decltype(slot)::value, slot, refill_c, T_{}, tab_c); },
                          std::make_integer_sequence<int, PF>{});
        };
        while (qb < nfast - PF) {                         // steady state: every refill and every next quad exists
            const int group_end = min(qb + CHECK_TURNS * PF, nfast - PF);
            if (table_safe()) for (; qb < group_end; qb += PF) turn(T_{}, T_{});
            else for (; qb < group_end; qb += PF) turn(T_{}, F_{});
        }
        // last ring turn: nothing left to prefetch, and the very last step has no next quad
        for_each_slot([&](auto slot) { step(qb + decltype(slot)::value, slot, F_{}, T_{}, F_{}); },
                      std::make_integer_sequence<int, PF - 1>{});
        step(qb + PF - 1, integral_constant<int, PF - 1>{}, F_{}, F_{}, F_{});
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
                    guarded_record<NA>(st, lds_sum, lds_cnt, lds_key, lane, aa[j], xr[j], qi * 4 + j, p, ov[j], oa[j]);
            if (has_sv) { Q4 o; o.x = (T)ov[0]; o.y = (T)ov[1]; o.z = (T)ov[2]; o.w = (T)ov[3]; SVq[(int64_t)qi * WAVE] = o; }
            if (has_sa) SAq[(int64_t)qi * WAVE] = make_uchar4(oa[0], oa[1], oa[2], oa[3]);
        }
    }

    if (s < S) {
        if (act_step) act_step[s] = st.latch >= LATCH_NEVER ? -1 : st.latch;
        if (vmax) vmax[s] = (float)st.best;
        if (amax) amax[s] = decode_action(st.best);
        if (V_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (a < A) V_out[(int64_t)s * A + a] = strip_code(reinterpret_cast<const double*>(&lds_key[a >> 1][lane])[a & 1]);
        }
        if (n_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) n_out[(int64_t)s * A + a] = lds_cnt[a][lane];
        }
    }
}

template <typename T, int NA, bool STEPS>
static void launch_tab_instance(int W, hipStream_t st, const T* R, const uint8_t* act, const int64_t* slice_row_off,
                                const int32_t* len, int S, int A, const DevParams& p, T* step_val, uint8_t* step_act,
                                int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax) {
    constexpr int WPB = tab_waves_per_block<NA>();
    constexpr unsigned bytes = TAB_N * sizeof(RootPair) + WPB * tab_wave_bytes<NA>();
    static_assert(bytes <= 160 * 1024, "LDS budget of a gfx950 CU");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_tab_kernel<T, NA, STEPS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)attr;
    hipLaunchKernelGGL((trace_tab_kernel<T, NA, STEPS>), dim3((W + WPB - 1) / WPB), dim3(WPB * WAVE), bytes, st, R, act,
                       slice_row_off, len, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax, amax);
    note_kernel("trace_tab_kernel<%s,%d,%s>", sizeof(T) == 4 ? "float" : "double", NA, STEPS ? "true" : "false");
}

// candidate counts 1..16 (exact instance each); returns false if A is not covered here
template <typename T>
bool launch_trace_tab(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int S, int A,
                      const DevParams& p, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                      int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st) {
    const int W = (S + WAVE - 1) / WAVE;
    if (A > 16) return false;
    if (W == 0) return true;
    const bool steps = step_val && step_act;
#define DCARL_CASE(NA)                                                                                              \
    case NA:                                                                                                        \
        if (steps) launch_tab_instance<T, NA, true>(W, st, R, act, slice_row_off, len, S, A, p, step_val, step_act, \
                                                    act_step, V_out, n_out, vmax, amax);                            \
        else launch_tab_instance<T, NA, false>(W, st, R, act, slice_row_off, len, S, A, p, step_val, step_act,      \
                                               act_step, V_out, n_out, vmax, amax);                                 \
        break
    switch (A) {
        DCARL_CASE(1); DCARL_CASE(2); DCARL_CASE(3); DCARL_CASE(4); DCARL_CASE(5); DCARL_CASE(6); DCARL_CASE(7);
        DCARL_CASE(8); DCARL_CASE(9); DCARL_CASE(10); DCARL_CASE(11); DCARL_CASE(12); DCARL_CASE(13);
        DCARL_CASE(14); DCARL_CASE(15); DCARL_CASE(16);
    }
#undef DCARL_CASE
    return true;
}

template bool launch_trace_tab<DCARL_TAB_T>(const DCARL_TAB_T*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                            const DevParams&, DCARL_TAB_T*, uint8_t*, int32_t*, double*, int32_t*,
                                            float*, int32_t*, hipStream_t);

}  // namespace dcarl
