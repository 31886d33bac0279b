// Online confidence estimation ("trace" mode) as PRODUCER / CONSUMER wavefront pairs, four pairs per workgroup.
// Same results as trace.hip / trace_tab_impl.h (S1:73-99 / S2:72-97); different mapping to the CU:
//
//   producer wave (one per 64-state slice)  loads the records, runs the statistics stage (S1:80) and the f64
//       evaluation (S1:87-90, count roots from the shared LDS table, trace_tab_impl.h) and posts, per record, the
//       tie-break-coded value and the LDS address of the key it replaces
//   consumer wave (one per slice)           overwrites the key (S1:86), reloads the candidate keys, arg-max tree,
//       latch (S1:93-99), stores the step traces
//
// Why: the per-state loop is sequential, so a slice cannot be split over time, and 65 536 states are only 1024 slices =
// one wavefront per SIMD.  A lone wavefront issues in order: its LDS instructions (~80 cycles per record) and its
// s_waitcnt stalls are not overlapped with anything.  Splitting the loop by STAGE puts two wavefronts on every SIMD
// (waves 0-3 of the workgroup are producers, 4-7 consumers; the hardware places wave i and i+4 on the same SIMD), so
// one wave's VALU work runs under the other's LDS traffic.
//
// Hand-over: a two-entry queue per pair in LDS plus two per-lane progress counters.  The LDS executes a wavefront's
// operations in order, so "write the entry, then write the counter" needs no s_waitcnt on the producer side and
// "read the counter, then read the entry" none beyond the data dependency on the consumer side; compiler-level
// ordering is enforced with asm memory clobbers around the counter accesses.  There is no s_barrier after the table
// fill, so the four pairs drift freely.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "trace_common.h"

namespace dcarl {

constexpr int PC_TAB_N = 3072;                           // counts 0 .. PC_TAB_N-1 in the shared count-root table
constexpr int PC_PAIRS = 4;                              // slices per workgroup
struct __attribute__((aligned(16))) PcRoots { double r, rho; };
struct PcQuadRoots { double r[4], rho[4]; };

template <class F, int... I>
__device__ __forceinline__ void pc_for_each_slot(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// LDS per slice: statistics NA x 64 x (16 + 4), keys (NA + 1) x 64 x 8, queue 2 x 4 x 64 x (8 + 4), counters 2 x 64 x 4
template <int NA> constexpr int pc_slice_bytes() { return NA * WAVE * 20 + (NA + 1) * WAVE * 8 + 2 * 4 * WAVE * 12 + 2 * WAVE * 4; }
template <int NA> constexpr int pc_lds_bytes() { return PC_TAB_N * 16 + PC_PAIRS * pc_slice_bytes<NA>(); }

#define PC_ORDER() asm volatile("" ::: "memory")
constexpr int PC_SPIN_LIMIT = 1 << 24;                   // a hand-over that never arrives traps instead of hanging

template <typename T, int NA>
__global__ __launch_bounds__(2 * PC_PAIRS * WAVE) void trace_pc_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, int S, int A, DevParams p, T* __restrict__ step_val,
    uint8_t* __restrict__ step_act, int32_t* __restrict__ act_step, double* __restrict__ V_out,
    int32_t* __restrict__ n_out, float* __restrict__ vmax, int32_t* __restrict__ amax) {
    using Q4 = typename Quad<T>::type;
    constexpr int PF = sizeof(T) == 4 ? 8 : 4;           // prefetch ring depth in quads (f64 storage: 256 VGPRs are tight)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PcRoots* tab = reinterpret_cast<PcRoots*>(smem);

    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wid & (PC_PAIRS - 1);
    const bool producer = wid < PC_PAIRS;
    const int W = (S + WAVE - 1) / WAVE;
    const int w = blockIdx.x * PC_PAIRS + pair;

    {   // fill the table as far as this workgroup's longest slice can count
        int64_t need = 0;
        for (int i = 0; i < PC_PAIRS; ++i) {
            const int wi = min(blockIdx.x * PC_PAIRS + i, W - 1);
            need = max(need, slice_row_off[wi + 1] - slice_row_off[wi]);
        }
        const int fill = (int)min((int64_t)PC_TAB_N, need + 2);
        for (int i = threadIdx.x; i < fill; i += 2 * PC_PAIRS * WAVE) {
            const CountRoots c = count_roots(max(i, 1));
            tab[i] = PcRoots{c.r, c.rho};
        }
    }
    unsigned char* mine = smem + PC_TAB_N * 16 + pair * pc_slice_bytes<NA>();
    SumPair (*lds_sum)[WAVE] = reinterpret_cast<SumPair (*)[WAVE]>(mine);
    int (*lds_cnt)[WAVE] = reinterpret_cast<int (*)[WAVE]>(mine + NA * WAVE * 16);
    double (*lds_key)[WAVE] = reinterpret_cast<double (*)[WAVE]>(mine + NA * WAVE * 20);
    double (*q_key)[4][WAVE] = reinterpret_cast<double (*)[4][WAVE]>(mine + NA * WAVE * 20 + (NA + 1) * WAVE * 8);
    int (*q_off)[4][WAVE] = reinterpret_cast<int (*)[4][WAVE]>(reinterpret_cast<unsigned char*>(q_key) + 2 * 4 * WAVE * 8);
    // quads posted / quads retired, one copy per lane.  Deliberately NOT volatile: the backend follows every volatile
    // access with s_waitcnt vmcnt(0) lgkmcnt(0), which would drain the HBM prefetch ring at every hand-over; the
    // asm memory clobbers (PC_ORDER) are what keeps the compiler from caching or moving these accesses.
    int* produced = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(q_off) + 2 * 4 * WAVE * 4);
    int* consumed = produced + WAVE;
    if (producer) { produced[lane] = 0; consumed[lane] = 0; }
    __syncthreads();                                     // table and counters visible; no barrier after this one
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
    const int lane_off = lane * (int)sizeof(double);
    const int trash_off = NA * WAVE * (int)sizeof(double) + lane_off;

    if (producer) {
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
        PcQuadRoots crt, nrt;
        auto roots_read = [&](PcQuadRoots& o, const QuadStat& q, int j0) {   // counts are < PC_TAB_N here (table_safe)
            const PcRoots t0 = tab[q.n[j0]], t1 = tab[q.n[j0 + 1]];
            o.r[j0] = t0.r; o.rho[j0] = t0.rho; o.r[j0 + 1] = t1.r; o.rho[j0 + 1] = t1.rho;
        };
        constexpr int CHECK_TURNS = 4;
        auto table_safe = [&]() {
            int m = 0;
#pragma unroll
            for (int a = 0; a < NA; ++a) m = max(m, lds_cnt[a][lane]);
            return __all(m + CHECK_TURNS * 4 * PF < PC_TAB_N) != 0;
        };
        // entry qi & 1 is free once the consumer has retired quad qi - 2
        auto wait_entry_free = [&](int qi) {
            int spins = 0;
            for (;;) {
                PC_ORDER();
                const int c = consumed[lane];
                PC_ORDER();
                if (__all(c >= qi - 1)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > PC_SPIN_LIMIT) __builtin_trap();
            }
        };
        auto post = [&](int qs, int j, int a, int n, double v) {
            q_key[qs][j][lane] = encode_key(v, a);
            q_off[qs][j][lane] = (n > p.n_thres) ? a * (WAVE * (int)sizeof(double)) + lane_off : trash_off;
        };
        auto publish = [&](int qi) { PC_ORDER(); produced[lane] = qi + 1; PC_ORDER(); };
        //   Aa1(q+1) | B(q) records 0,1 | Aa2(q+1), roots, Ab1(q+1) | B(q) records 2,3 | Ab2(q+1), roots | post(q)
        auto step = [&](int qi, auto slot, auto refill_c, auto more_c, auto tab_c) {
            constexpr int i = decltype(slot)::value;
            constexpr bool REFILL = decltype(refill_c)::value, MORE = decltype(more_c)::value, TAB = decltype(tab_c)::value;
            constexpr int in = (i + 1) % PF;
            if (REFILL) { rbuf[i] = Rq[(int64_t)(qi + PF) * WAVE]; abuf[i] = Aq[(int64_t)(qi + PF) * WAVE]; }
            PC_ORDER();
            const int retired = consumed[lane];           // read now, checked before the post: never drains the LDS queue
            PC_ORDER();
            if (MORE) pair_read<NA>(pa, shift, lds_sum, lds_cnt, lane, abuf[in].x, abuf[in].y, (double)rbuf[in].x,
                                    (double)rbuf[in].y);
            auto value = [&](int j) {
                return TAB ? value_from_roots(crt.r[j], crt.rho[j], cur.s[j], cur.q[j], shift, cur.a[j] == p.rule_act, p)
                           : value_from_sums(cur.n[j], cur.s[j], cur.q[j], shift, cur.a[j] == p.rule_act, p);
            };
            double v[4];
            v[0] = value(0);
            v[1] = value(1);
            if (MORE) {
                pair_update(nxt, 0, pa, lds_sum, lds_cnt, lane);
                if (TAB) roots_read(nrt, nxt, 0);
                pair_read<NA>(pb, shift, lds_sum, lds_cnt, lane, abuf[in].z, abuf[in].w, (double)rbuf[in].z,
                              (double)rbuf[in].w);
            }
            v[2] = value(2);
            v[3] = value(3);
            if (MORE) {
                pair_update(nxt, 2, pb, lds_sum, lds_cnt, lane);
                if (TAB) roots_read(nrt, nxt, 2);
            }
            if (!__all(retired >= qi - 1)) wait_entry_free(qi);   // rare: the consumer is the faster of the two
#pragma unroll
            for (int j = 0; j < 4; ++j) post(i & 1, j, cur.a[j], cur.n[j], v[j]);
            publish(qi);
            if (MORE) { cur = nxt; if (TAB) crt = nrt; }
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
            roots_read(crt, cur, 0);
            roots_read(crt, cur, 2);
            auto turn = [&](auto refill_c, auto tab_c) {
                pc_for_each_slot([&](auto slot) { step(qb + decltype(slot)::value, slot, refill_c, T_{}, tab_c); },
                                 std::make_integer_sequence<int, PF>{});
            };
            while (qb < nfast - PF) {
                const int group_end = min(qb + CHECK_TURNS * PF, nfast - PF);
                if (table_safe()) for (; qb < group_end; qb += PF) turn(T_{}, T_{});
                else for (; qb < group_end; qb += PF) turn(T_{}, F_{});
            }
            pc_for_each_slot([&](auto slot) { step(qb + decltype(slot)::value, slot, F_{}, T_{}, F_{}); },
                             std::make_integer_sequence<int, PF - 1>{});
            step(qb + PF - 1, integral_constant<int, PF - 1>{}, F_{}, F_{}, F_{});
            qb += PF;
        }
        // tail: ragged ends of the slice, per-lane guards; records past a lane's end are posted to the trash slot
        for (int qi = qb; qi < nquads; ++qi) {
            const int qs = qi & 1;
            wait_entry_free(qi);
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
            publish(qi);
        }
        if (s < S && n_out) {
#pragma unroll
            for (int a = 0; a < NA; ++a) if (a < A) n_out[(int64_t)s * A + a] = lds_cnt[a][lane];
        }
    } else {
        // ================================ consumer ================================================================
        double best;
        {                                                 // S1:50-53 initial table, tie-break coded
            double key[NA + 1];
#pragma unroll
            for (int a = 0; a < NA + 1; ++a) {
                key[a] = (a < A) ? encode_key(a == p.rule_act ? p.init_rule : p.init_other, a) : encode_key(-1e300, a & 31);
                lds_key[a][lane] = key[a];
            }
            best = tree_max<NA>(key);
        }
        int latch = 0x7fffffff;
        Q4* SVq = reinterpret_cast<Q4*>(step_val) + row0 / 4 * WAVE + lane;
        uchar4* SAq = reinterpret_cast<uchar4*>(step_act) + row0 / 4 * WAVE + lane;
        const bool has_sv = step_val != nullptr, has_sa = step_act != nullptr;       // wave-uniform
        unsigned char* key_base = reinterpret_cast<unsigned char*>(&lds_key[0][0]);

        auto consume = [&](int qi, int qs) {
            int spins = 0;
            for (;;) {                                    // entry qs holds quad qi once the producer says so
                PC_ORDER();
                const int c = produced[lane];
                PC_ORDER();
                if (__all(c >= qi + 1)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > PC_SPIN_LIMIT) __builtin_trap();
            }
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
            PC_ORDER();
            consumed[lane] = qi + 1;                      // the entry's reads were issued before this write
            PC_ORDER();
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
            if (has_sv) { Q4 o; o.x = (T)ov[0]; o.y = (T)ov[1]; o.z = (T)ov[2]; o.w = (T)ov[3]; SVq[(int64_t)qi * WAVE] = o; }
            if (has_sa) SAq[(int64_t)qi * WAVE] = make_uchar4(oa[0], oa[1], oa[2], oa[3]);
        };
        int qi = 0;
        for (; qi + 1 < nquads; qi += 2) { consume(qi, 0); consume(qi + 1, 1); }
        if (qi < nquads) consume(qi, 0);

        if (s < S) {
            if (act_step) act_step[s] = latch == 0x7fffffff ? -1 : latch;
            if (vmax) vmax[s] = (float)best;
            if (amax) amax[s] = decode_action(best);
            if (V_out) {
#pragma unroll
                for (int a = 0; a < NA; ++a) if (a < A) V_out[(int64_t)s * A + a] = strip_code(lds_key[a][lane]);
            }
        }
    }
}

template <typename T, int NA>
static void launch_pc_instance(int W, hipStream_t st, const T* R, const uint8_t* act, const int64_t* slice_row_off,
                               const int32_t* len, int S, int A, const DevParams& p, T* step_val, uint8_t* step_act,
                               int32_t* act_step, double* V_out, int32_t* n_out, float* vmax, int32_t* amax) {
    constexpr unsigned bytes = pc_lds_bytes<NA>();
    static_assert(bytes <= 160 * 1024, "LDS budget of a gfx950 CU");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_pc_kernel<T, NA>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    (void)attr;
    hipLaunchKernelGGL((trace_pc_kernel<T, NA>), dim3((W + PC_PAIRS - 1) / PC_PAIRS), dim3(2 * PC_PAIRS * WAVE), bytes, st,
                       R, act, slice_row_off, len, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax, amax);
}

// returns false if A has no instance here
template <typename T>
bool launch_trace_pair(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int S, int A,
                       const DevParams& p, T* step_val, uint8_t* step_act, int32_t* act_step, double* V_out,
                       int32_t* n_out, float* vmax, int32_t* amax, hipStream_t st) {
    const int W = (S + WAVE - 1) / WAVE;
    if (A != 11) return false;
    if (W == 0) return true;
    launch_pc_instance<T, 11>(W, st, R, act, slice_row_off, len, S, A, p, step_val, step_act, act_step, V_out, n_out, vmax,
                              amax);
    return true;
}

template bool launch_trace_pair<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                       const DevParams&, float*, uint8_t*, int32_t*, double*, int32_t*, float*,
                                       int32_t*, hipStream_t);
template bool launch_trace_pair<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, int, int,
                                        const DevParams&, double*, uint8_t*, int32_t*, double*, int32_t*, float*,
                                        int32_t*, hipStream_t);

}  // namespace dcarl
