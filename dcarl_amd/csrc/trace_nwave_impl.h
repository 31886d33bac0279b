// Online confidence estimation ("trace" mode), NW wavefronts per 64-state slice taking its quads ROUND-ROBIN.
// Same arithmetic and results as the one-wave kernel of trace.hip (S1:73-99 / S2:72-97); different schedule.
//
// The per-state loop is sequential, but only through two short stages: the statistics stage A (S1:80: append to the
// bucket) must see every earlier append, and the commit stage C (S1:86,93-99: overwrite the key, arg-max) must see every
// earlier overwrite.  The evaluation B (S1:87-90) of a record depends on its own bucket's statistics only.  So wave w of
// a slice takes quads w, w + NW, w + 2 NW, ...; statistics and keys are in LDS anyway and are simply shared; two
// monotone per-slice counters in LDS order the stages across the waves:
//
//     a_done = number of quads whose statistics are appended     (A(q) may start when a_done >= q)
//     c_done = number of quads whose key overwrites are issued   (C(q) may start when c_done >= q)
//
//     wave 0:  A(0) B(0)..... C(0)                 A(3) B(3)..... C(3)
//     wave 1:       A(1) B(1)..... C(1)                 A(4) ...
//     wave 2:            A(2) B(2)..... C(2)                 A(5) ...
//
// (NW = 3 drawn; f32 storage ships NW = 4 since round 6, f64 NW = 3: launch_trace_nwave below.)
// No data moves between the waves (unlike a producer/consumer split), the work is balanced by
// construction, and 65 536 states become 3072 / 4096 wavefronts = three / four per SIMD, so one wave's VALU work runs under the
// others' LDS instructions and waits (VALU-active 59 % of SIMD time with one wave, 69 % with two, 78 % with three).
// The hand-over is "issue the stage's LDS writes, release fence, write the counter" on one side and "read the counter,
// acquire fence, read the data" on the other (the fences are lgkmcnt(0) drains; see FENCED below for the bare form that
// relies on the LDS executing in issue order).  The counters are LDS words (one copy per lane) accessed with relaxed
// workgroup-scope atomics between asm memory clobbers -- volatile accesses
// would make the backend drain vmcnt/lgkmcnt after each one.  Three waves per SIMD leave 168 VGPRs, four 128: a quad's stages run
// one after the other (no software pipeline inside a wave -- the other waves are the pipeline) and the four arg-max
// trees of a quad are done two at a time.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "trace_common.h"

namespace dcarl {

// counts 0 .. N-1 in the workgroup's count-root table of 16-byte entries {1/sqrt(n), 2/sqrt(n+1)} (ONE ds_read_b128 per
// record): 4096 entries = 64 KiB up to 12 candidates, 2048 = 32 KiB for 13..16 so that four slices still fit
// the CU's 160 KiB.  (Measured and rejected: 8-byte entries r[n] fetched as r[n], r[n+1] with ds_read2_b64 halve the table
// but the instruction is two 8-byte accesses banked mod 32 — SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE went from 19.5 % to
// 25.9 %, profiles/r02_pmc_trace_nwave3_SQ_LDS_8byte_table.csv.)
#ifndef DCARL_NWV_TAB12
#define DCARL_NWV_TAB12 4096
#endif
#ifndef DCARL_NWV_SLICES
#define DCARL_NWV_SLICES 4
#endif
#ifndef DCARL_NWV_TAB16
#define DCARL_NWV_TAB16 2048
#endif
template <int NA> constexpr int nwv_tab_n() { return NA <= 12 ? DCARL_NWV_TAB12 : DCARL_NWV_TAB16; }
constexpr int NWV_SLICES = DCARL_NWV_SLICES;             // slices per workgroup at most (the launch chooses 1..: nwv_slices_for)
struct __attribute__((aligned(16))) NwvRoots { double r, rho; };
typedef double nwv_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) nwv_d2 LdsRoots;       // (one ds_read_b128)

template <class F, int... I>
__device__ __forceinline__ void nwv_for_each(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
__device__ __forceinline__ uchar4 nwv_uchar4(unsigned v) { return make_uchar4(v & 255u, (v >> 8) & 255u, (v >> 16) & 255u, v >> 24); }

// The arg-max of a quad: quad_commit_* (trace_common.h) — the (<= 4) touched key slots exchanged against a knocked key, ONE re-load
// of the keys, the four new keys written, and the four maxima from the untouched slots' maximum + the touched slots' values as they
// stand after each record.  Round 6; it replaced both forms of rounds 1-5 for every candidate count: the per-record re-load + tree
// (up to 12 candidates: 7 LDS operations and 10 v_max_f64 per record) and the carried {best, bound} pair with re-scans (13..16).
// Same-box A/B on configs[1] / [3] / [4] (A = 11 / 11 / 16): 3.313 -> 3.137, 2.785 -> 2.637, 1.271 -> 1.200 ms (profiles/r06_ab_online.txt).
// LDS per slice: statistics NA x 64 x (16 + 4); keys key_rows<NA>() x 64 x 8; two counters + (latch, done flag) per extra wave
template <int NA> constexpr int nwv_cells() { return key_cells<NA>(); }
template <int NA, int NW> constexpr int nwv_slice_bytes() {
    return NA * WAVE * 20 + nwv_cells<NA>() * WAVE * 16 + (2 + 2 * (NW - 1)) * WAVE * 4;
}
template <int NA, int NW> constexpr int nwv_lds_bytes(int ns) { return nwv_tab_n<NA>() * 16 + ns * nwv_slice_bytes<NA, NW>(); }
// the most slices a workgroup of this instance can hold: the CU's 160 KiB of LDS, 1 024 threads
template <int NA, int NW> constexpr int nwv_max_slices() {
    int ns = NWV_SLICES;
    while (ns > 1 && (nwv_lds_bytes<NA, NW>(ns) > 160 * 1024 || NW * ns * WAVE > 1024)) --ns;
    return ns;
}

#define NWV_ORDER() asm volatile("" ::: "memory")
#ifndef DCARL_TRACE_NT
// Every record is read once and every trace element written once: loads and stores of the fast path are marked non-temporal (they neither
// find anything in the L2 nor leave anything worth keeping).  Same-box A/B of four builds (tools/experiments/ab_trace_nt.sh): headline 3.363 ->
// 3.345 (stores) / 3.310 (loads) / 3.281 ms (both); configs[3] online 2.826 -> 2.745, configs[4] 1.285 -> 1.253.  bit 0: stores, bit 1: loads.
#define DCARL_TRACE_NT 3
#endif


// FENCED (the DEFAULT since round 4): the hand-over with workgroup-scope release / acquire fences around relaxed atomic accesses
// of the counters (what the C++ memory model asks for).  The bare form, which relies on the hardware-ordering assumption
// documented at publish() below, is 0.9 % faster on the headline and is kept for the shapes of the A/B measurement only
// (DCARL_TRACE_FENCED=0; tools/experiments/ab_fenced.py): no ISA-manual sentence in reach of this build guarantees cross-wave DS issue
// order, so it is not what ships.  tests/test_hardening.py checks that both forms give bit-identical outputs.
int* trace_fault_word();     // trace.hip: device word a hand-over that never arrives sets before its wave ends (dcarl_trace_status)

template <typename T, int NA, int NW, bool STEPS, bool FENCED = true>
__global__ __launch_bounds__((NW * nwv_max_slices<NA, NW>() * WAVE)) void trace_nwave_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* act_step, double* V_out,          // (a resumed launch reads these three through cy)
    int32_t* n_out, float* __restrict__ vmax, int32_t* __restrict__ amax, int ns, int* __restrict__ fault,
    const TraceCarry cy) {
    using Q4 = typename Quad<T>::type;
    // Own quads per turn (two banks of PF quads are the prefetch registers; a pair of turns is 2 * NW * PF quads, and what does not fill one goes
    // quad by quad).  Round 6, same-box A/B of four builds after the quad-commit rewrite (profiles/r06_ab_online_prefetch.txt): PF = 2 / 3 / 4 / 6 =
    // 3.154 / 3.124 / 3.136 / 3.148 ms on the headline, 2.630 / 2.609 / 2.630 / 2.657 on configs[3], 1.198 / 1.188 / 1.202 / 1.267 on configs[4]'s
    // 1 000-record streams: three.  (Round 4, before it: 3.315 / 3.29 / 3.284 / 3.252 and 1.239 / 1.245 / 1.247 / 1.354: four.)
#ifndef DCARL_TRACE_PF
#define DCARL_TRACE_PF 3
#endif
#ifndef DCARL_TRACE_PF4
#define DCARL_TRACE_PF4 3
#endif
    // (four waves, no step traces: the third quad does not fit 128 registers there — 36 bytes of scratch for 2..9 candidates — so two)
    constexpr int PF = sizeof(T) == 8 ? 2 : NW >= 4 ? (STEPS ? DCARL_TRACE_PF4 : 2) : DCARL_TRACE_PF;   // (the macros: A/B builds)
    constexpr int NP = nwv_cells<NA>();                  // 16-byte units of keys per lane (two rows each)
    constexpr int KR = key_rows<NA>();                   // key rows: candidates, the trash row, (padding)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TAB_N = nwv_tab_n<NA>();
    NwvRoots* tab = reinterpret_cast<NwvRoots*>(smem);

    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ns slices per workgroup (a launch argument, 1..4): wave wid serves slice wid % ns as its wave wid / ns, so with
    // ns = 4 waves i, i + 4, i + 8 share a SIMD and a slice.  (wid < NW * ns <= 16: the quotient by comparisons.)
    const int wv = (wid >= ns) + (wid >= 2 * ns) + (NW > 3 ? (wid >= 3 * ns) : 0);   // this wave takes quads wv, wv + NW, ...
    const int sl = wid - wv * ns;
    const int W = slices_of(S);
    const int w = blockIdx.x * ns + sl;

    {   // fill the table as far as this workgroup's longest slice can count
        int64_t need = 0;
        for (int i = 0; i < ns; ++i) {
            const int wi = min(blockIdx.x * ns + i, W - 1);
            need = max(need, slice_row_off[wi + 1] - slice_row_off[wi]);
        }
        // (a resumed state's buckets start at their earlier counts: the whole table then)
        const int fill = (cy.n != nullptr && !cy.fresh) ? TAB_N : (int)min((int64_t)TAB_N, need + 2);
        for (int i = threadIdx.x; i < fill; i += NW * ns * WAVE) {
            const CountRoots c = count_roots(max(i, 1));
            tab[i] = NwvRoots{c.r, c.rho};
        }
    }
    unsigned char* mine = smem + TAB_N * 16 + sl * nwv_slice_bytes<NA, NW>();
    LdsRow* lrows = reinterpret_cast<LdsRow*>(mine);      // sums, sums of squares, keys: 2 NA + KR rows (trace_common.h)
    KeyRow* lds_key = key_rows_of<NA>(lrows);
    int (*lds_cnt)[WAVE] = reinterpret_cast<int (*)[WAVE]>(mine + (NA + NP) * WAVE * 16);   // bucket sizes x NS (below)
    const unsigned row_base = (unsigned)(size_t)(LdsF64*)(&lrows[0][lane]);     // LDS address of this lane's element of row 0 ...
    const unsigned cnt_base = (unsigned)(size_t)(LdsU32*)(&lds_cnt[0][lane]);   // ... and of counter row 0
    // The counters hold the bucket sizes in units of NS = 16: a counter IS the byte offset of its bucket's entry in the count-root
    // table (16-byte entries at LDS offset 0), so a record's table look-up needs no address arithmetic, and the size after the
    // append exceeds n_thres exactly when the counter before it was >= 16 n_thres.  Sizes therefore stay below 2^27: checked once
    // per launch below (a bucket of 134 million samples of ONE state is refused through the fault word, not mis-counted).
    constexpr int NS = 16;
    const bool tab_at_zero = (unsigned)(size_t)(LdsF64*)tab == 0u;
    static_assert(sizeof(NwvRoots) == NS, "a counter step = one table entry");
    const int thr_raw = p.n_thres * NS;
    int* a_done = reinterpret_cast<int*>(mine + NA * WAVE * 20 + NP * WAVE * 16);
    int* c_done = a_done + WAVE;
    int* latch_x = c_done + WAVE;                        // latches of waves 1..NW-1, handed to wave 0 at the end
    int* fin = latch_x + (NW - 1) * WAVE;                // "wave w is done" flags

    // the state this lane serves and, for a resumed loop (dcarl_trace_resume_*), what it has seen so far
    const bool resumed = cy.n != nullptr && !cy.fresh;                    // launch-uniform
    const bool live_state = w < W && w * WAVE + lane < S;
    const int so_pre = (live_state && slot_state) ? slot_state[w * WAVE + lane] : w * WAVE + lane;
    const CarryIn cin = carry_in(cy, live_state, so_pre, A);
    // S1:50-53 initial table (tie-break coded) and empty buckets -- or the resumed state's statistics and values: wave 0 of the
    // slice sets it up before the barrier
    if (wv == 0) {
        const bool from_state = resumed && live_state;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            SumPair sp{0.0, 0.0};
            int cn = 0;
            if (from_state && a < A) {
                sp = SumPair{cy.sum[(int64_t)so_pre * A + a], cy.sumsq[(int64_t)so_pre * A + a]};
                cn = cy.n[(int64_t)so_pre * A + a];
            }
            lrows[a][lane] = sp.s;
            lrows[NA + a][lane] = sp.q;
            lds_cnt[a][lane] = (cn < 0 || cn >= (1 << 27)) ? 0x7fffffff : cn * NS;      // (beyond the counters' range: caught below)
        }
#pragma unroll
        for (int a = 0; a < KR; ++a) {
            double v0 = a == p.rule_act ? p.init_rule : p.init_other;
            if (from_state && a < A) v0 = cy.V[(int64_t)so_pre * A + a];    // encode_key(strip_code(key)) == key: the same keys
            lds_key[a][lane] = (a < A) ? encode_key(v0, a) : encode_key(-1e300, a & 31);    // (the trash row too: below every real key)
        }
        a_done[lane] = 0;
        c_done[lane] = 0;
#pragma unroll
        for (int o = 0; o < NW - 1; ++o) fin[o * WAVE + lane] = 0;
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
    const int nfast = (min_len >> 2) / (2 * NW * PF) * (2 * NW * PF);   // quads (whole pairs of turns of all waves) with every lane live
    {   // the counters' range (see NS above): what the buckets hold + what this launch can add must stay below 2^27 samples
        int m = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a) m = max(m, lds_cnt[a][lane]);
        if (__any((unsigned)m / NS + (unsigned)max_len >= (1u << 27))) {
            if (lane == 0) atomicMax(fault, 2);          // dcarl_trace_status: 2 = a bucket beyond the online kernel's count range
            return;                                      // (every wave of the slice sees the same counters and leaves the same way)
        }
    }

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
    if (resumed && cin.t_base > 0) st.shift = cy.shift[so_pre];             // K stays the state's first reward of ALL launches
    const unsigned rule4 = (unsigned)p.rule_act * 0x01010101u;

    // ---- hand-over helpers --------------------------------------------------------------------------------------
    // FENCED (default): publish() = release fence + relaxed workgroup-scope atomic store of the counter, wait_for() = relaxed
    // atomic loads (peek / the spin) + acquire fence: the stage's LDS writes happen-before every read behind the wait.
    // !FENCED (A/B only) relies on a HARDWARE-ORDERING ASSUMPTION instead: a CU's LDS executes the DS operations of one
    // wavefront in issue order and a DS write is visible to every later-issued DS read of any wavefront of the workgroup, so
    // that publish() needs no s_waitcnt between the stage's data writes and the counter write and peek() none before the
    // data reads that follow it -- a formal data race in the C++ memory model, gfx9-specific, 0.9 % faster (the fence pair
    // costs an lgkmcnt(0) drain per stage).  The compiler-level NWV_ORDER() barriers keep the accesses in program order.
    auto peek = [&](const int* counter) __attribute__((always_inline)) {
        NWV_ORDER();
        int c;
        if constexpr (FENCED) c = __hip_atomic_load(counter + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else c = counter[lane];
        NWV_ORDER();
        return c;
    };
    // The whole wait is ONE asm statement (check of the value read earlier, then the spin): C++ control flow in the middle
    // of the pipeline step makes the waitcnt pass give up on counting the HBM prefetch ring across it.
    auto wait_for = [&](const int* counter, int seen, int need) __attribute__((always_inline)) {     // `seen` was read earlier; spin only if stale
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const int*)(counter + lane);
        int budget = 1 << 28;                              // ~3e10 cycles: a hand-over that NEVER arrives does not hang the GPU and
                                                           // does not kill the HIP context either: the wave raises the library's fault
                                                           // word (dcarl_trace_status reports it) and ends; its partners, waiting
                                                           // for ITS hand-overs, follow the same way.  The launch's outputs are void then.
        const int one = 1;
#ifndef DCARL_SPIN_SLEEP
#define DCARL_SPIN_SLEEP 1                                 // s_sleep argument between two polls (64 clocks each); -1: no sleep at all
#endif
#define DCARL_STR2(x) #x
#define DCARL_STR(x) DCARL_STR2(x)
        asm volatile(
            "v_cmp_gt_i32 vcc, %3, %0\n\t"        // lanes whose copy is still below `need`
            "s_cbranch_vccz 2f\n\t"
            "1:\n\t"                              // stale: poll FIRST, sleep only between polls (round 6: the sleep used to come first)
            "ds_read_b32 %0, %2\n\t"
            "s_sub_u32 %1, %1, 1\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_gt_i32 vcc, %3, %0\n\t"
            "s_cbranch_vccz 2f\n\t"
#if DCARL_SPIN_SLEEP >= 0
            "s_sleep " DCARL_STR(DCARL_SPIN_SLEEP) "\n\t"
#endif
            "s_branch 1b\n\t"
            "3:\n\t"
            "s_store_dword %5, %4, 0x0 glc\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_dcache_wb\n\t"
            "s_endpgm\n\t"
            "2:"
            : "+v"(seen), "+s"(budget) : "v"(addr), "s"(need), "s"(fault), "s"(one) : "vcc", "scc", "memory");
        if constexpr (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto publish = [&](int* counter, int value) __attribute__((always_inline)) {
        if constexpr (FENCED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        NWV_ORDER();
        if constexpr (FENCED) __hip_atomic_store(counter + lane, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else counter[lane] = value;
        NWV_ORDER();
    };

    // ---- fast path: this wave's quads are q = wv, wv + NW, ... < nfast ---------------------------------------------
    // Two banks of PF own quads: while one bank is consumed (a "turn" = PF own quads = NW*PF quads of the slice) the other
    // is in flight, loaded in one go.  (A slot-by-slot ring would do with half the registers, but with the hand-over asm
    // in the loop body the waitcnt pass settles for ONE vmcnt(0) per loop iteration; with whole banks that wait is for
    // loads issued a full turn earlier.)
    Q4 rbuf[2][PF];
    uchar4 abuf[2][PF];
    auto load_bank = [&](auto bank, int q0) __attribute__((always_inline)) {
        constexpr int b = decltype(bank)::value;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if constexpr ((DCARL_TRACE_NT & 2) != 0) {
                rbuf[b][i] = nt_load_quad<T>(&at_lane(Rw + (int64_t)(q0 + wv + NW * i) * WAVE));
                abuf[b][i] = nwv_uchar4(__builtin_nontemporal_load(&at_lane(Aw + (int64_t)(q0 + wv + NW * i) * WAVE)));
            } else {
            rbuf[b][i] = at_lane(Rw + (int64_t)(q0 + wv + NW * i) * WAVE);
            abuf[b][i] = nwv_uchar4(at_lane(Aw + (int64_t)(q0 + wv + NW * i) * WAVE));
            }
        }
    };
    // every bucket of every lane stays inside the count-root table for the next pair of turns (2*NW*PF quads of the slice
    // = that many * 4 records per lane, + the quads the other waves may be ahead); otherwise the pair runs on the compute
    // path -- bit-identical values, so the waves of a slice may even decide differently
    auto table_safe = [&]() __attribute__((always_inline)) {
        int m = 0;
#pragma unroll
        for (int a = 0; a < NA; ++a) m = max(m, lds_cnt[a][lane]);
        // (the look-up uses a counter as the entry's LDS ADDRESS: the table is the first thing in the kernel's only LDS allocation, i.e. at
        //  address 0 — were it ever not, every record takes the compute path: the same values, bit for bit)
        return tab_at_zero && __all(m < (TAB_N - 2 * NW * PF * 4 - 16) * NS) != 0;
    };
    auto step = [&](int qi, auto bank, auto slot, auto tab_c) __attribute__((always_inline)) {
        constexpr int i = decltype(slot)::value, b = decltype(bank)::value;
        constexpr bool TAB = decltype(tab_c)::value;
        QuadStat cur;
        NwvRoots rt[4];
        QuadIn in;
        {                                                 // A(qi)
            {
                const int aa[4] = {abuf[b][i].x, abuf[b][i].y, abuf[b][i].z, abuf[b][i].w};
                const double xx[4] = {(double)rbuf[b][i].x, (double)rbuf[b][i].y, (double)rbuf[b][i].z, (double)rbuf[b][i].w};
                quad_in<NA>(in, st.shift, row_base, cnt_base, aa, xx);
                // (consumed by a volatile statement: computed before the wait below, not inside the handed-over stage)
                asm volatile("" :: "v"(in.ra[0]), "v"(in.ra[1]), "v"(in.ra[2]), "v"(in.ra[3]), "v"(in.rc[0]), "v"(in.rc[1]), "v"(in.rc[2]), "v"(in.rc[3]),
                             "v"(in.x[0]), "v"(in.x[1]), "v"(in.x[2]), "v"(in.x[3]));
            }
            // (the counter is read right here, not earlier under the VALU work above: an early read is usually stale and sends the wave
            //  into the polling loop — measured +3 %, profiles/r06_ab_online_handover.txt)
            wait_for(a_done, peek(a_done), qi);
            __builtin_amdgcn_s_setprio(3);
            count_quad<NS>(cur, in);                      // the four counters: atomics, back to back, outside the chain below
            prepared_append<NA>(cur, 0, in);
            prepared_append<NA>(cur, 1, in);
            prepared_append<NA>(cur, 2, in);
            prepared_append<NA>(cur, 3, in);
            publish(a_done, qi + 1);
            __builtin_amdgcn_s_setprio(0);
            if (TAB) {                                    // the entry of the size AFTER the append: one past the counter's own
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const nwv_d2 e = *(const LdsRoots*)(size_t)((unsigned)cur.n[j] + (unsigned)sizeof(NwvRoots));   // (the table sits at LDS address 0: table_safe)
                    rt[j] = NwvRoots{e.x, e.y};
                }
            }
        }
        double v[4];                                      // B(qi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = TAB ? value_from_roots(rt[j].r, rt[j].rho, cur.s[j], cur.q[j], st.shift, cur.a[j] == p.rule_act, p)
                       : value_from_sums((int)((unsigned)cur.n[j] / NS) + 1, cur.s[j], cur.q[j], st.shift, cur.a[j] == p.rule_act, p);
        }
        double ov[4];
        int oa[4];
        {                                                 // C(qi), the quad at once (trace_common.h, quad_commit_*)
            QuadCommit qc;
            quad_commit_prepare<NA>(qc, row_base, in, cur, v, thr_raw);
            // (consumed by a volatile statement: keys and addresses are made BEFORE the wait, outside the handed-over stage)
            asm volatile("" :: "v"(qc.addr[0]), "v"(qc.addr[1]), "v"(qc.addr[2]), "v"(qc.addr[3]), "v"(qc.kk[0]), "v"(qc.kk[1]), "v"(qc.kk[2]), "v"(qc.kk[3]));
            wait_for(c_done, peek(c_done), qi);
            __builtin_amdgcn_s_setprio(2);                // the commit chain is the other stage the waves wait on (-2 %)
            double key[NA];
            quad_commit_issue<NA>(qc, key, lds_key, lane);
            publish(c_done, qi + 1);
            __builtin_amdgcn_s_setprio(0);
            quad_commit_finish<NA>(qc, key, ov, oa);
        }
        const unsigned packed = (unsigned)oa[0] | ((unsigned)oa[1] << 8) | ((unsigned)oa[2] << 16) | ((unsigned)oa[3] << 24);
        latch_quad(st.latch, packed, rule4, qi * 4);
        if (has_sv) {
            Q4 o; o.x = step_out<T>(ov[0]); o.y = step_out<T>(ov[1]); o.z = step_out<T>(ov[2]); o.w = step_out<T>(ov[3]);
            if constexpr ((DCARL_TRACE_NT & 1) != 0) nt_store_quad<T>(&at_lane(SVw + (int64_t)qi * WAVE), o);
            else at_lane(SVw + (int64_t)qi * WAVE) = o;
        }
        if (has_sa) {
            if constexpr ((DCARL_TRACE_NT & 1) != 0) __builtin_nontemporal_store(packed, &at_lane(SAw + (int64_t)qi * WAVE));
            else at_lane(SAw + (int64_t)qi * WAVE) = packed;
        }
    };
    using std::integral_constant;
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    int qb = 0;
    if (nfast > 0) {
        auto turn = [&](int q0, auto bank, auto tab_c) __attribute__((always_inline)) {
            nwv_for_each([&](auto slot) __attribute__((always_inline)) { step(q0 + wv + NW * decltype(slot)::value, bank, slot, tab_c); },
                         std::make_integer_sequence<int, PF>{});
        };
        load_bank(I0{}, 0);
        for (; qb < nfast; qb += 2 * NW * PF) {
            const bool more = qb + 2 * NW * PF < nfast;
            load_bank(I1{}, qb + NW * PF);
            if (table_safe()) {
                turn(qb, I0{}, T_{});
                if (more) load_bank(I0{}, qb + 2 * NW * PF);
                turn(qb + NW * PF, I1{}, T_{});
            } else {
                turn(qb, I0{}, F_{});
                if (more) load_bank(I0{}, qb + 2 * NW * PF);
                turn(qb + NW * PF, I1{}, F_{});
            }
        }
    }
    // ---- the quads between the last whole pair of turns and the first stream end: still every lane live, still round-
    // robin over the waves, one quad at a time without the prefetch banks.  Without this stretch up to 2*NW*PF - 1 = 23 quads fell to the one-wave guarded tail below, which
    // for streams of ~1 000 records (configs[4] in its online form) was a fifth of the run time.
    {
        const int nfq = min_len >> 2;
        const bool tab_ok = qb + wv < nfq && table_safe();       // (the stretch is shorter than the pair of turns the check covers)
        for (int q = qb + wv; q < nfq; q += NW) {
            rbuf[0][0] = at_lane(Rw + (int64_t)q * WAVE);
            abuf[0][0] = nwv_uchar4(at_lane(Aw + (int64_t)q * WAVE));
            if (tab_ok) step(q, I0{}, integral_constant<int, 0>{}, T_{});
            else step(q, I0{}, integral_constant<int, 0>{}, F_{});
        }
        qb = max(qb, nfq);
    }
    // ---- tail: ragged ends of the slice, per-lane guards; wave 0 alone, after every fast quad is committed ---------
    if (wv != 0) {
        latch_x[(wv - 1) * WAVE + lane] = st.latch;
        publish(fin + (wv - 1) * WAVE, 1);
        return;
    }
#pragma unroll
    for (int o = 0; o < NW - 1; ++o) {
        wait_for(fin + o * WAVE, peek(fin + o * WAVE), 1);
        st.latch = min(st.latch, peek(latch_x + o * WAVE));
    }
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
                        guarded_record<NA, NS>(st, lrows, lds_cnt, lane, aa[j], xr[j], qi * 4 + j, p, ov[j], oa[j]);
                if (has_sv) { Q4 o; o.x = step_out<T>(ov[0]); o.y = step_out<T>(ov[1]); o.z = step_out<T>(ov[2]); o.w = step_out<T>(ov[3]); SVq[(int64_t)qi * WAVE] = o; }
                if (has_sa) SAq[(int64_t)qi * WAVE] = make_uchar4(oa[0], oa[1], oa[2], oa[3]);
            }
        }
    }
    if (s < S) {
        double key[NA];                                   // final table = the keys as they stand
#pragma unroll
        for (int a = 0; a < NA; ++a) key[a] = lds_key[a][lane];
        const double best = tree_max<NA>(key);
        const int so = so_pre;                            // per-state outputs go to the state's own row, not the slot's
        if (act_step) act_step[so] = carry_latch(cin, st.latch, LATCH_NEVER);
        if (cy.n != nullptr) {                            // the advanced sufficient statistic (V, n, latch: the outputs below)
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (a < A) { cy.sum[(int64_t)so * A + a] = lrows[a][lane]; cy.sumsq[(int64_t)so * A + a] = lrows[NA + a][lane]; }
            cy.shift[so] = st.shift;
        }
        if (vmax) vmax[so] = (float)best;
        if (amax) amax[so] = decode_action(best);
        if (V_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) V_out[(int64_t)so * A + a] = strip_code(key[a]);
        }
        if (n_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) n_out[(int64_t)so * A + a] = (int)((unsigned)lds_cnt[a][lane] / NS);
        }
    }
}

// Slices per workgroup.  A workgroup carries its own count-root table (32 / 64 KiB), so only ONE workgroup is resident per
// CU whatever its slice count: four slices (12 waves) per workgroup fill a CU, but a table of fewer than 4 x CUs slices
// would then leave CUs empty (32 768 states on 128 of 256 CUs: 4.1 instead of 2.7 ps per record).  So the launcher picks the
// slice count per table size.  DCARL_TRACE_SLICES=1..4 overrides (A/B runs).
static int nwv_slices_for(int W, int cap) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    if (const char* e = DCARL_KNOB("DCARL_TRACE_SLICES")) {
        const int v = atoi(e);
        if (v >= 1 && v <= cap) return v;
    }
    // one round of workgroups: as few slices each as still give every CU one (16 384 states, 4 096 records each: 0.42 vs
    // 0.55 ms with four; 32 768: 0.52 vs 0.60).  Beyond one round four per workgroup is as good as anything: a cost model
    // over rounds x round-time chose three for some sizes and measured within 2 % (tools/experiments/bench_states.py).
    const int ns = (W + cus - 1) / cus;
    return ns < 1 ? 1 : ns > cap ? cap : ns;
}

template <typename T, int NA, int NW, bool STEPS, bool FENCED = true>
static void launch_nwv_instance(int W, hipStream_t st, const T* R, const uint8_t* act, const int64_t* slice_row_off,
                                const int32_t* len, const int32_t* slot_state, int S, int A, const DevParams& p, T* step_val, uint8_t* step_act,
                                int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax, const TraceCarry& cy) {
    constexpr unsigned max_bytes = nwv_lds_bytes<NA, NW>(nwv_max_slices<NA, NW>());
    static_assert(max_bytes <= 160 * 1024, "LDS budget of a gfx950 CU");
    DCARL_RAISE_LDS_LIMIT(((int)max_bytes), trace_nwave_kernel<T, NA, NW, STEPS, FENCED>);
    const int ns = nwv_slices_for(W, nwv_max_slices<NA, NW>());
    const unsigned bytes = (unsigned)nwv_lds_bytes<NA, NW>(ns);
    hipLaunchKernelGGL((trace_nwave_kernel<T, NA, NW, STEPS, FENCED>), dim3((W + ns - 1) / ns), dim3(NW * ns * WAVE), bytes,
                       st, R, act, slice_row_off, len, slot_state, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax, amax, ns,
                       trace_fault_word(), cy);
    note_kernel("trace_nwave_kernel<%s,%d,%d,%s>%s", sizeof(T) == 4 ? "float" : "double", NA, NW, STEPS ? "true" : "false",
                FENCED ? "" : " unfenced");
}

// Every candidate count up to 16 and both storage types; returns false for A > 16 (the one-wave kernel of trace.hip takes those).
// f32 storage: FOUR waves per slice (16 per CU, 124 VGPRs each) with three own quads per turn — round 6, after the quad commit had
// made room for the third quad in a 128-register budget: same box, three waves / four waves with two quads / four with three =
// 3.235 / 3.209 / 3.177 ms on the headline; every f32 instance on four: configs[1] 3.116 -> 3.056, configs[3] 2.607 -> 2.571,
// configs[4] 1.191 -> 1.184 (profiles/r06_ab_online_four_waves.txt).  (Round 2, before any of it: 3.523 vs 3.538 — "not short of
// waves".)  f64 storage stays on three (its quads are twice the registers).  LDS: 64 KiB table + 4 slices of 24 576 B = 163 840 B for
// 12 candidates, 32 KiB + 4 x 31 744 B for 16.
// waves_per_slice: 0 = the shipped choice; 2 / 3 / 4 (DCARL_TRACE_KERNEL=duo / trio / quad) run the instances compiled for A/B runs.
template <typename T>
bool launch_trace_nwave(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int S, int A,
                        const DevParams& p, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                        int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st, int waves_per_slice, const TraceCarry& cy) {
    const int W = slices_of(S);
    if (A > 16) return false;
    if (W == 0) return true;
    const bool steps = step_val && step_act;
#define DCARL_ARGS W, st, R, act, slice_row_off, len, slot_state, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax, amax, cy
#ifndef DCARL_NWV_F32_WAVES
#define DCARL_NWV_F32_WAVES 4
#endif
#ifndef DCARL_NWV_F64_WAVES
#define DCARL_NWV_F64_WAVES 3
#endif
    constexpr int NWD = sizeof(T) == 4 ? DCARL_NWV_F32_WAVES : DCARL_NWV_F64_WAVES;       // waves per slice of the shipped instances
#define DCARL_CASE3(NA)                                                                       \
    case NA:                                                                                  \
        if (steps) launch_nwv_instance<T, NA, NWD, true>(DCARL_ARGS);                         \
        else launch_nwv_instance<T, NA, NWD, false>(DCARL_ARGS);                              \
        break
#ifdef DCARL_AB_BUILD
    const int wps = waves_per_slice == 0 ? NWD : waves_per_slice;
    // the bare hand-over (DCARL_TRACE_FENCED=0), compiled for the shapes the equivalence test and the cost measurement use
    if (const char* e = DCARL_KNOB("DCARL_TRACE_FENCED"); e && e[0] == '0' && wps == NWD && steps) {
        if constexpr (sizeof(T) == 4) {
            if (A == 11) { launch_nwv_instance<T, 11, NWD, true, false>(DCARL_ARGS); return true; }
            if (A == 16) { launch_nwv_instance<T, 16, NWD, true, false>(DCARL_ARGS); return true; }
            if (A == 5) { launch_nwv_instance<T, 5, NWD, true, false>(DCARL_ARGS); return true; }
        } else {
            if (A == 12) { launch_nwv_instance<T, 12, NWD, true, false>(DCARL_ARGS); return true; }
        }
    }
    if constexpr (sizeof(T) == 4 && NWD != 3) if (wps == 3 && (A == 11 || A == 16)) {
        if (A == 11) { if (steps) launch_nwv_instance<T, 11, 3, true>(DCARL_ARGS); else launch_nwv_instance<T, 11, 3, false>(DCARL_ARGS); }
        else { if (steps) launch_nwv_instance<T, 16, 3, true>(DCARL_ARGS); else launch_nwv_instance<T, 16, 3, false>(DCARL_ARGS); }
        return true;
    }
    if constexpr (sizeof(T) == 4 && NWD != 4) if (wps == 4 && A == 11) {
        if (steps) launch_nwv_instance<T, 11, 4, true>(DCARL_ARGS); else launch_nwv_instance<T, 11, 4, false>(DCARL_ARGS);
        return true;
    }
    if constexpr (sizeof(T) == 4) if (wps == 2 && (A == 11 || A == 16)) {
        if (A == 11) { if (steps) launch_nwv_instance<T, 11, 2, true>(DCARL_ARGS); else launch_nwv_instance<T, 11, 2, false>(DCARL_ARGS); }
        else { if (steps) launch_nwv_instance<T, 16, 2, true>(DCARL_ARGS); else launch_nwv_instance<T, 16, 2, false>(DCARL_ARGS); }
        return true;
    }
#else
    (void)waves_per_slice;
#endif
    switch (A) {
        DCARL_CASE3(1); DCARL_CASE3(2); DCARL_CASE3(3); DCARL_CASE3(4); DCARL_CASE3(5); DCARL_CASE3(6); DCARL_CASE3(7);
        DCARL_CASE3(8); DCARL_CASE3(9); DCARL_CASE3(10); DCARL_CASE3(11); DCARL_CASE3(12); DCARL_CASE3(13);
        DCARL_CASE3(14); DCARL_CASE3(15); DCARL_CASE3(16);
    }
#undef DCARL_CASE3
#undef DCARL_ARGS
    return true;
}

}  // namespace dcarl
