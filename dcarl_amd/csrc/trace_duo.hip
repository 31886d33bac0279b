// Online confidence estimation ("trace" mode), TWO wavefronts per 64-state slice taking ALTERNATE quads.
// Same arithmetic and results as trace_tab_impl.h (S1:73-99 / S2:72-97); different schedule.
//
// The per-state loop is sequential, but only through two short stages: the statistics stage A (S1:80: append to the
// bucket) must see every earlier append, and the commit stage C (S1:86,93-99: overwrite the key, arg-max) must see every
// earlier overwrite.  The evaluation B (S1:87-90) of a record depends on its own bucket's statistics only.  So wave X
// takes quads 0,2,4,... and wave Y quads 1,3,5,... of the same slice; statistics and keys are in LDS anyway and are
// simply shared; two monotone per-slice counters in LDS order the stages across the two waves:
//
//     a_done = number of quads whose statistics are appended     (A(q) may start when a_done >= q)
//     c_done = number of quads whose key overwrites are issued   (C(q) may start when c_done >= q)
//
//     X:  A(q)   B(q)......  C(q)          A(q+2)  B(q+2)......  C(q+2)
//     Y:         A(q+1)  B(q+1)......  C(q+1)          A(q+3) ...
//
// No data moves between the waves (unlike a producer/consumer split), the work is balanced by
// construction, and 65 536 states become 2048 wavefronts = two per SIMD, so one wave's VALU work runs under the other's
// LDS instructions and waits.  The LDS executes each wavefront's operations in order, so "issue the stage's LDS writes,
// then write the counter" needs no s_waitcnt; the counters are plain LDS words (one copy per lane) between asm memory
// clobbers -- volatile accesses would make the backend drain vmcnt/lgkmcnt after each one.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "trace_common.h"

namespace dcarl {

constexpr int DUO_TAB_N = 4096;                          // counts 0 .. DUO_TAB_N-1 in the shared count-root table
constexpr int DUO_SLICES = 4;                            // slices per workgroup (8 wavefronts)
struct __attribute__((aligned(16))) DuoRoots { double r, rho; };

template <class F, int... I>
__device__ __forceinline__ void duo_for_each(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
__device__ __forceinline__ uchar4 duo_uchar4(unsigned v) { return make_uchar4(v & 255u, (v >> 8) & 255u, (v >> 16) & 255u, v >> 24); }

// LDS per slice: statistics NA x 64 x (16 + 4), keys (NA/2 + 1) x 64 x 16, counters + latch exchange 3 x 64 x 4
template <int NA> constexpr int duo_slice_bytes() { return NA * WAVE * 20 + key_cells<NA>() * WAVE * 16 + 3 * WAVE * 4; }
template <int NA> constexpr int duo_lds_bytes() { return DUO_TAB_N * 16 + DUO_SLICES * duo_slice_bytes<NA>(); }

#define DUO_ORDER() asm volatile("" ::: "memory")


template <typename T, int NA, bool STEPS>
__global__ __launch_bounds__(2 * DUO_SLICES * WAVE) void trace_duo_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* __restrict__ act_step, double* __restrict__ V_out,
    int32_t* __restrict__ n_out, float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using Q4 = typename Quad<T>::type;
    constexpr int PF = 4;                                // own quads per turn (= 8 quads of the slice = 32 records per lane)
    constexpr int NP = key_cells<NA>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DuoRoots* tab = reinterpret_cast<DuoRoots*>(smem);

    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sl = wid & (DUO_SLICES - 1);               // waves i and i + 4 share a SIMD and a slice (the same pair on
    const int parity = wid >> 2;                         // different SIMDs measured 6 % slower)
    const int W = (S + WAVE - 1) / WAVE;
    const int w = blockIdx.x * DUO_SLICES + sl;

    {   // fill the table as far as this workgroup's longest slice can count
        int64_t need = 0;
        for (int i = 0; i < DUO_SLICES; ++i) {
            const int wi = min(blockIdx.x * DUO_SLICES + i, W - 1);
            need = max(need, slice_row_off[wi + 1] - slice_row_off[wi]);
        }
        const int fill = (int)min((int64_t)DUO_TAB_N, need + 2);
        for (int i = threadIdx.x; i < fill; i += 2 * DUO_SLICES * WAVE) {
            const CountRoots c = count_roots(max(i, 1));
            tab[i] = DuoRoots{c.r, c.rho};
        }
    }
    unsigned char* mine = smem + DUO_TAB_N * 16 + sl * duo_slice_bytes<NA>();
    SumPair (*lds_sum)[WAVE] = reinterpret_cast<SumPair (*)[WAVE]>(mine);
    KeyPair (*lds_key)[WAVE] = reinterpret_cast<KeyPair (*)[WAVE]>(mine + NA * WAVE * 16);
    int (*lds_cnt)[WAVE] = reinterpret_cast<int (*)[WAVE]>(mine + (NA + NP) * WAVE * 16);
    int* a_done = reinterpret_cast<int*>(mine + NA * WAVE * 20 + NP * WAVE * 16);
    int* c_done = a_done + WAVE;
    int* latch_x = c_done + WAVE;                        // wave Y's latch, handed to wave X at the end

    // S1:50-53 initial table (tie-break coded) and empty buckets: wave X sets the slice up before the barrier
    if (parity == 0) {
#pragma unroll
        for (int a = 0; a < NA; ++a) { lds_sum[a][lane] = SumPair{0.0, 0.0}; lds_cnt[a][lane] = 0; }
        double key[2 * NP];
#pragma unroll
        for (int a = 0; a < 2 * NP; ++a)
            key[a] = (a < A) ? encode_key(a == p.rule_act ? p.init_rule : p.init_other, a) : encode_key(-1e300, a & 31);
#pragma unroll
        for (int c = 0; c < NP; ++c) lds_key[c][lane] = KeyPair{key[2 * c], key[2 * c + 1]};
        a_done[lane] = 0;
        c_done[lane] = 0;
    }
    __syncthreads();                                     // the only barrier: table, buckets, keys, counters are set
    if (w >= W) return;

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
    const int nfast = (min_len >> 2) / (4 * PF) * (4 * PF);   // quads (whole pairs of turns of both waves) with every lane live

    const Q4* Rw = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE;
    const unsigned* Aw = reinterpret_cast<const unsigned*>(act) + row0 / 4 * WAVE;
    Q4* SVw = reinterpret_cast<Q4*>(step_val) + row0 / 4 * WAVE;
    unsigned* SAw = reinterpret_cast<unsigned*>(step_act) + row0 / 4 * WAVE;
    auto at_lane = [lane](auto* base) __attribute__((always_inline)) -> decltype(*base)& {
        using E = std::remove_reference_t<decltype(*base)>;
        using B = std::conditional_t<std::is_const<E>::value, const unsigned char, unsigned char>;
        return *reinterpret_cast<E*>(reinterpret_cast<B*>(base) + (unsigned)(lane * (int)sizeof(E)));
    };
    const bool has_sv = STEPS || step_val != nullptr, has_sa = STEPS || step_act != nullptr;   // wave-uniform

    LaneState<NA> st;
    st.best = 0.0;
    st.latch = 0x7fffffff;
    st.shift = (my_len > 0) ? (double)R[(row0 * WAVE) + lane * 4] : 0.0;
    const unsigned rule4 = (unsigned)p.rule_act * 0x01010101u;

    // ---- hand-over helpers --------------------------------------------------------------------------------------
    auto peek = [&](const int* counter) __attribute__((always_inline)) { DUO_ORDER(); const int c = counter[lane]; DUO_ORDER(); return c; };
    // The whole wait is ONE asm statement (check of the value read earlier, then the spin): C++ control flow in the middle
    // of the pipeline step makes the waitcnt pass give up on counting the HBM prefetch ring across it.
    auto wait_for = [&](const int* counter, int seen, int need) __attribute__((always_inline)) {     // `seen` was read earlier; spin only if stale
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const int*)(counter + lane);
        int budget = 1 << 24;                              // a hand-over that never arrives traps instead of hanging
        asm volatile(
            "v_cmp_gt_i32 vcc, %3, %0\n\t"        // lanes whose copy is still below `need`
            "s_cbranch_vccz 2f\n\t"
            "1:\n\t"
            "s_sleep 1\n\t"
            "s_sub_u32 %1, %1, 1\n\t"
            "s_cbranch_scc1 3f\n\t"
            "ds_read_b32 %0, %2\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_gt_i32 vcc, %3, %0\n\t"
            "s_cbranch_vccnz 1b\n\t"
            "s_branch 2f\n\t"
            "3:\n\t"
            "s_trap 2\n\t"
            "2:"
            : "+v"(seen), "+s"(budget) : "v"(addr), "s"(need) : "vcc", "scc", "memory");
    };
    auto publish = [&](int* counter, int value) __attribute__((always_inline)) { DUO_ORDER(); counter[lane] = value; DUO_ORDER(); };

    // ---- fast path: this wave's quads are q = parity, parity + 2, ... < nfast ------------------------------------
    // Two banks of PF own quads: while one bank is consumed (one "turn" = PF own quads = 2*PF quads of the slice) the
    // other is in flight, loaded in one go at the start of the turn.  (A slot-by-slot ring would do with half the
    // registers, but with the hand-over asm in the loop body the waitcnt pass settles for ONE vmcnt(0) per loop
    // iteration; with whole banks that wait is for loads issued a full turn earlier.)
    Q4 rbuf[2][PF];
    uchar4 abuf[2][PF];
    auto load_bank = [&](auto bank, int q0) __attribute__((always_inline)) {             // own quads of the turn starting at slice quad q0
        constexpr int b = decltype(bank)::value;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            rbuf[b][i] = at_lane(Rw + (int64_t)(q0 + parity + 2 * i) * WAVE);
            abuf[b][i] = duo_uchar4(at_lane(Aw + (int64_t)(q0 + parity + 2 * i) * WAVE));
        }
    };
    constexpr int CHECK_TURNS = 2;                        // table-range check every 2 turns = 16 quads = 64 records/lane
    auto table_safe = [&]() __attribute__((always_inline)) {
        int m = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a) m = max(m, lds_cnt[a][lane]);
        return __all(m + CHECK_TURNS * 2 * PF * 4 + 16 < DUO_TAB_N) != 0;   // + the pipelined A stages of both waves
    };
    // Per own quad q a wave runs  B(q)  and then  C(q) interleaved with A(q+2)  (its next own quad).  B is pure VALU; the
    // C/A block is where the LDS round trips are (key reloads, two statistics pairs) and it is the block that the
    // counters serialise between the two waves -- so in steady state one wave's B runs under the other's C/A block.
    //   B(q) | wait c | C1(q) issue, publish c | wait a | Aa1(q+2) | trees 0,1 | Aa2, roots, Ab1 | trees 2,3, outputs |
    //   Ab2, publish a, roots
    QuadStat qstat[2];                                    // own quad in slot i uses entry i & 1 (PF is even)
    DuoRoots qroot[2][4];
    auto evaluate = [&](double (&v)[4], const QuadStat& q, const DuoRoots (&rt)[4], auto tab_c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = decltype(tab_c)::value
                       ? value_from_roots(rt[j].r, rt[j].rho, q.s[j], q.q[j], st.shift, q.a[j] == p.rule_act, p)
                       : value_from_sums(q.n[j], q.s[j], q.q[j], st.shift, q.a[j] == p.rule_act, p);
    };
    auto stage_a_first = [&](int qi) __attribute__((always_inline)) {     // prologue: A of this wave's first quad
        PairRaw pa, pb;
        const Q4 rq = rbuf[0][0];
        const uchar4 aq = abuf[0][0];
        wait_for(a_done, peek(a_done), qi);
        pair_read<NA>(pa, st.shift, lds_sum, lds_cnt, lane, aq.x, aq.y, (double)rq.x, (double)rq.y);
        pair_update(qstat[0], 0, pa, lds_sum, lds_cnt, lane);
        pair_read<NA>(pb, st.shift, lds_sum, lds_cnt, lane, aq.z, aq.w, (double)rq.z, (double)rq.w);
        pair_update(qstat[0], 2, pb, lds_sum, lds_cnt, lane);
        publish(a_done, qi + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) qroot[0][j] = tab[qstat[0].n[j]];
    };
    //   B(q) | wait c | C1(q) issue, publish c | wait a | Aa1(q+2) | trees 0,1 | Aa2, roots, Ab1 | trees 2,3, outputs |
    //   Ab2, publish a, roots          (A(q+2) runs at raised wave priority: it is the stage the other wave waits for)
    auto step = [&](int qi, auto bank, auto slot, auto more_c, auto tab_c) __attribute__((always_inline)) {
        constexpr int i = decltype(slot)::value, b = decltype(bank)::value;
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int in = (i + 1) % PF, bn = (i + 1 < PF) ? b : 1 - b;      // slot / bank of this wave's next quad
        QuadStat& cur = qstat[i & 1];
        QuadStat& nxt = qstat[in & 1];
        DuoRoots (&nrt)[4] = qroot[in & 1];
        const int seen_c = peek(c_done);                  // read now, checked after the evaluation
        const int seen_a = peek(a_done);
        double v[4];
        evaluate(v, cur, qroot[i & 1], tab_c);            // B(qi)
        wait_for(c_done, seen_c, qi);                     // C1(qi): needs the key overwrites of every earlier quad
        double k0[NA], k1[NA], k2[NA], k3[NA];
        commit_issue<NA>(k0, lds_key, lane, cur.a[0], cur.n[0], v[0], p);
        commit_issue<NA>(k1, lds_key, lane, cur.a[1], cur.n[1], v[1], p);
        commit_issue<NA>(k2, lds_key, lane, cur.a[2], cur.n[2], v[2], p);
        commit_issue<NA>(k3, lds_key, lane, cur.a[3], cur.n[3], v[3], p);
        publish(c_done, qi + 1);
        PairRaw pa, pb;
        if (MORE) {                                       // Aa1(qi+2): needs the statistics of every earlier quad
            wait_for(a_done, seen_a, qi + 2);
            __builtin_amdgcn_s_setprio(3);
            pair_read<NA>(pa, st.shift, lds_sum, lds_cnt, lane, abuf[bn][in].x, abuf[bn][in].y, (double)rbuf[bn][in].x,
                          (double)rbuf[bn][in].y);
        }
        double ov[4];
        int oa[4];
        commit_finish<NA>(st, k0, ov[0], oa[0]);
        commit_finish<NA>(st, k1, ov[1], oa[1]);
        if (MORE) {
            pair_update(nxt, 0, pa, lds_sum, lds_cnt, lane);
            if (decltype(tab_c)::value) { nrt[0] = tab[nxt.n[0]]; nrt[1] = tab[nxt.n[1]]; }
            pair_read<NA>(pb, st.shift, lds_sum, lds_cnt, lane, abuf[bn][in].z, abuf[bn][in].w, (double)rbuf[bn][in].z,
                          (double)rbuf[bn][in].w);
        }
        commit_finish<NA>(st, k2, ov[2], oa[2]);
        commit_finish<NA>(st, k3, ov[3], oa[3]);
        const unsigned packed = (unsigned)oa[0] | ((unsigned)oa[1] << 8) | ((unsigned)oa[2] << 16) | ((unsigned)oa[3] << 24);
        latch_quad(st.latch, packed, rule4, qi * 4);
        if (has_sv) { Q4 o; o.x = (T)ov[0]; o.y = (T)ov[1]; o.z = (T)ov[2]; o.w = (T)ov[3]; at_lane(SVw + (int64_t)qi * WAVE) = o; }
        if (has_sa) at_lane(SAw + (int64_t)qi * WAVE) = packed;
        if (MORE) {
            pair_update(nxt, 2, pb, lds_sum, lds_cnt, lane);
            publish(a_done, qi + 3);
            __builtin_amdgcn_s_setprio(0);
            if (decltype(tab_c)::value) { nrt[2] = tab[nxt.n[2]]; nrt[3] = tab[nxt.n[3]]; }
        }
    };
    using std::integral_constant;
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    int qb = 0;                                           // first slice quad of the current pair of turns
    if (nfast > 0) {
        auto turn = [&](int q0, auto bank, auto tab_c) __attribute__((always_inline)) {
            duo_for_each([&](auto slot) __attribute__((always_inline)) { step(q0 + parity + 2 * decltype(slot)::value, bank, slot, T_{}, tab_c); },
                         std::make_integer_sequence<int, PF>{});
        };
        // a pair of turns: bank 0 is consumed while bank 1 flies and vice versa
        auto turn_pair = [&](auto tab_c) __attribute__((always_inline)) {
            turn(qb, I0{}, tab_c);                        // (its last step starts on bank 1, loaded a turn ago)
            load_bank(I0{}, qb + 4 * PF);
            turn(qb + 2 * PF, I1{}, tab_c);               // (its last step starts on the bank 0 just requested)
            load_bank(I1{}, qb + 6 * PF);
        };
        load_bank(I0{}, 0);
        if (nfast > 2 * PF) load_bank(I1{}, 2 * PF);
        stage_a_first(parity);
        // steady state: two further pairs of turns exist, so every prefetch and every next quad is in range
        for (; qb < nfast - 8 * PF; qb += 4 * PF) {
            // the counts read here include exactly the quads before this wave's next A stage (the other wave cannot
            // pass it); the pair of turns appends 16 quads = 64 records per lane
            if (table_safe()) turn_pair(T_{});
            else turn_pair(F_{});
        }
        // the last one or two pairs of turns: guarded prefetch, compute path, no next quad after the very last step
        for (; qb < nfast; qb += 4 * PF) {
            const bool more_pairs = qb + 4 * PF < nfast;
            turn(qb, I0{}, F_{});
            if (more_pairs) load_bank(I0{}, qb + 4 * PF);
            duo_for_each([&](auto slot) __attribute__((always_inline)) { step(qb + 2 * PF + parity + 2 * decltype(slot)::value, I1{}, slot, T_{}, F_{}); },
                         std::make_integer_sequence<int, PF - 1>{});
            if (more_pairs) {
                step(qb + 2 * PF + parity + 2 * (PF - 1), I1{}, integral_constant<int, PF - 1>{}, T_{}, F_{});
                if (qb + 6 * PF < nfast) load_bank(I1{}, qb + 6 * PF);
            } else {
                step(qb + 2 * PF + parity + 2 * (PF - 1), I1{}, integral_constant<int, PF - 1>{}, F_{}, F_{});
            }
        }
    }
    // ---- tail: ragged ends of the slice, per-lane guards; wave X alone, after every fast quad is committed -------
    if (parity == 1) {
        latch_x[lane] = st.latch;
        publish(c_done, nfast + 1);                       // "wave Y is done" (c_done == nfast means: last fast commit issued)
        return;
    }
    if (nfast > 0) wait_for(c_done, peek(c_done), nfast + 1);
    st.latch = min(st.latch, (nfast > 0) ? peek(latch_x) : 0x7fffffff);
    {
        const Q4* Rq = Rw + lane;
        const uchar4* Aq = reinterpret_cast<const uchar4*>(Aw) + lane;
        Q4* SVq = SVw + lane;
        uchar4* SAq = reinterpret_cast<uchar4*>(SAw) + lane;
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
    }
    if (s < S) {
        double key[NA];                                   // final table = the keys as they stand
#pragma unroll
        for (int a = 0; a < NA; ++a) key[a] = reinterpret_cast<const double*>(&lds_key[a >> 1][lane])[a & 1];
        const double best = tree_max<NA>(key);
        if (act_step) act_step[s] = st.latch >= LATCH_NEVER ? -1 : st.latch;
        if (vmax) vmax[s] = (float)best;
        if (amax) amax[s] = decode_action(best);
        if (V_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) V_out[(int64_t)s * A + a] = strip_code(key[a]);
        }
        if (n_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) n_out[(int64_t)s * A + a] = lds_cnt[a][lane];
        }
    }
}

template <typename T, int NA, bool STEPS>
static void launch_duo_instance(int W, hipStream_t st, const T* R, const uint8_t* act, const int64_t* slice_row_off,
                                const int32_t* len, int S, int A, const DevParams& p, T* step_val, uint8_t* step_act,
                                int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax) {
    constexpr unsigned bytes = duo_lds_bytes<NA>();
    static_assert(bytes <= 160 * 1024, "LDS budget of a gfx950 CU");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_duo_kernel<T, NA, STEPS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)attr;
    hipLaunchKernelGGL((trace_duo_kernel<T, NA, STEPS>), dim3((W + DUO_SLICES - 1) / DUO_SLICES), dim3(2 * DUO_SLICES * WAVE), bytes,
                       st, R, act, slice_row_off, len, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax, amax);
}

// fp32 record storage with up to 12 candidates (256 VGPRs per wave at two waves per SIMD: the 13-candidate instance
// would spill; LDS: 64 KiB table + 4 slices <= 160 KiB up to 13); returns false otherwise.  f64 storage stays on the
// one-wave kernels (its prefetch banks do not fit the register budget either).
bool launch_trace_duo(const float* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int S, int A,
                      const DevParams& p, float* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                      int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st) {
    const int W = (S + WAVE - 1) / WAVE;
    if (A > 12) return false;
    if (W == 0) return true;
    const bool steps = step_val && step_act;
#define DCARL_CASE(NA)                                                                                                  \
    case NA:                                                                                                            \
        if (steps) launch_duo_instance<float, NA, true>(W, st, R, act, slice_row_off, len, S, A, p, step_val, step_act, \
                                                        act_step, V_out, n_out, vmax, amax);                            \
        else launch_duo_instance<float, NA, false>(W, st, R, act, slice_row_off, len, S, A, p, step_val, step_act,      \
                                                   act_step, V_out, n_out, vmax, amax);                                 \
        break
    switch (A) {
        DCARL_CASE(1); DCARL_CASE(2); DCARL_CASE(3); DCARL_CASE(4); DCARL_CASE(5); DCARL_CASE(6); DCARL_CASE(7);
        DCARL_CASE(8); DCARL_CASE(9); DCARL_CASE(10); DCARL_CASE(11); DCARL_CASE(12);
    }
#undef DCARL_CASE
    return true;
}

}  // namespace dcarl
