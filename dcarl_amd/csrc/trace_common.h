// Pieces shared by the online ("trace") kernels: the quad-packed record types and the statistics stage (S1:80).
#pragma once
#include "common.h"

namespace dcarl {

// Continuation of the online loop (dcarl_trace_resume_*, include/dcarl.h): the caller's state arrays, per STATE.  n == nullptr:
// a one-shot launch (dcarl_trace_*).  Otherwise the kernel starts from the arrays unless `fresh`, and leaves the advanced state
// in them (V / n / act_step through its ordinary V_out / n_out / act_step arguments, which the launcher points at the state).
struct TraceCarry {
    const int32_t* n;
    double* sum;
    double* sumsq;
    double* shift;
    const double* V;
    const int32_t* act_step;
    int fresh;
};
// what a lane needs of its state's history before its first record of this launch
struct CarryIn {
    int t_base;       // records of the state in earlier launches (the sum of its bucket sizes)
    int latch;        // activation step latched earlier, -1 = not yet
};
__device__ __forceinline__ CarryIn carry_in(const TraceCarry& cy, bool live, int so, int A) {
    CarryIn c{0, -1};
    if (cy.n != nullptr && !cy.fresh && live) {
        for (int a = 0; a < A; ++a) c.t_base += cy.n[(int64_t)so * A + a];
        c.latch = cy.act_step[so];
    }
    return c;
}
// S1:98-99 across launches: an earlier latch stands; otherwise this launch's (counted from its own first record) + t_base
__device__ __forceinline__ int carry_latch(const CarryIn& c, int local_latch, int never) {
    if (c.latch >= 0) return c.latch;
    return local_latch >= never ? -1 : local_latch + c.t_base;
}

template <typename T> struct Quad;
template <> struct Quad<float> { using type = float4; };
template <> struct Quad<double> { using type = double4; };

struct __attribute__((aligned(16))) SumPair { double s, q; };

// non-temporal access to a quad (four consecutive records of one state): the builtins want a native vector type, HIP's float4 /
// double4 are structs
template <typename T> struct NtVec;
template <> struct NtVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct NtVec<double> { typedef double type __attribute__((ext_vector_type(4))); };
template <typename T> __device__ __forceinline__ typename Quad<T>::type nt_load_quad(const typename Quad<T>::type* p) {
    const typename NtVec<T>::type v = __builtin_nontemporal_load(reinterpret_cast<const typename NtVec<T>::type*>(p));
    typename Quad<T>::type r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
    return r;
}
template <typename T> __device__ __forceinline__ void nt_store_quad(typename Quad<T>::type* p, const typename Quad<T>::type& o) {
    const typename NtVec<T>::type v = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(v, reinterpret_cast<typename NtVec<T>::type*>(p));
}

// max_a V[s][a] as it is stored into the step trace (S1:93): the running maximum is a tie-break-coded key; with f64
// record storage its 5 code bits would reach the caller (the initial 100.0 would read 100.00000000000044), so they are
// cleared (one v_and); the conversion to f32 storage drops them anyway.
template <typename T> __device__ __forceinline__ T step_out(double key) {
    if constexpr (sizeof(T) == 8) return strip_code(key);
    else return (T)key;
}

// ---- the per-slice state in LDS: ROWS of 64 lanes x 8 bytes ([row][lane]; lane = state) ------------------------------------------
//   rows [0, NA)            sum(x - K) of bucket a                    } the bucket's sufficient statistic (S1:80); its size lives in a
//   rows [NA, 2 NA)         sum((x - K)^2) of bucket a                } separate array of 4-byte counts, [a][lane]
//   rows [2 NA, 2 NA + KR)  the current values V[s][.] as tie-break-coded f64 keys: KR = key_rows<NA>(): rows 0..NA-1 of them are the
//                           candidates, row NA the TRASH row — records whose bucket is still below the threshold (S1:86) write there
// A per-lane action id is an LDS address, never a register index: ONE v_lshl_add_u32 (row_base + (a << 9)) gives the lane's element
// of bucket a's sum row, and its sum of squares and its key sit at compile-time offsets from it (NA * 512, 2 NA * 512: the DS
// instructions' offset fields).  Both statistics travel in one ds_read2st64_b64 / ds_write2st64_b64, the keys come back two rows per
// ds_read2st64_b64.  (Rounds 1-5: {sum, sum of squares} as one 16-byte cell per (a, lane) and two keys per 16-byte cell — one
// address computation per array and three VALU operations for a key slot's address)
typedef double LdsRow[WAVE];
typedef LdsRow KeyRow;
typedef __attribute__((address_space(3))) double LdsF64;
typedef __attribute__((address_space(3))) unsigned long long LdsU64;
typedef __attribute__((address_space(3))) unsigned LdsU32;
template <int NA> constexpr int key_cells() { return NA / 2 + 1; }           // 16-byte units per lane: key_rows / 2
template <int NA> constexpr int key_rows() { return 2 * key_cells<NA>(); }   // rows 0..NA-1 = candidates, row NA = trash (+ one unused for even NA)
template <int NA> constexpr int state_rows() { return 2 * NA + key_rows<NA>(); }
template <int NA> __device__ __forceinline__ KeyRow* key_rows_of(LdsRow* rows) { return rows + 2 * NA; }

// ---- the one-wave kernels' fast path: four consecutive records (a "quad") of one state, every lane live, as a software pipeline ---
//   Aa1(q+1) issue the LDS reads of the buckets of records 0,1 of the next quad        (S1:80, statistics)
//   B(q)     four independent f64 evaluations                                          (S1:87-90)
//   C1(q)    LDS traffic of the four commits, back to back                             (S1:86, write key / reload keys)
//   Aa2(q+1) append records 0,1, write back;  Ab1(q+1) issue the reads of records 2,3
//   C2(q)    four max trees, arg-max decode, latch                                     (S1:93-99)
//   Ab2(q+1) append records 2,3, write back
// so that every LDS round trip completes behind VALU work of another stage (at 1-2 waves per SIMD instruction-
// level overlap is the only latency hiding there is).  The LDS executes in order, which is what makes a read see every
// earlier write-back and the reload of commit j see the keys of commits 0..j.  Additions happen in arrival order, so
// the sums are bit-identical to a record-by-record update.
// n[j]: one-wave kernels: the bucket's size AFTER the append.  Multi-wave kernel: what the bucket's counter held BEFORE the append, in
// its unit of 16 per sample (the counter doubles as the byte offset into the count-root table): count_quad below.
struct QuadStat { int a[4], n[4]; double s[4], q[4]; };   // bucket statistics right after each of the four appends

// The statistics stage works on PAIRS of records: both bucket reads are issued together and, if the two records hit
// the same bucket, the second takes the first one's updated statistics from registers (one compare + five selects).
// Doing all four records of a quad at once would need six such checks; the second pair's reads are simply issued
// after the first pair's write-back, so they already see it.
struct PairRaw { int a0, a1; double x0, x1; SumPair b0, b1; int c0, c1; };

template <int NA>
__device__ __forceinline__ void pair_read(PairRaw& r, double shift, LdsRow* rows, int (*lds_cnt)[WAVE],
                                          int lane, int act0, int act1, double xr0, double xr1) {
    r.a0 = min(act0, NA - 1); r.a1 = min(act1, NA - 1);
    r.x0 = xr0 - shift; r.x1 = xr1 - shift;
    r.b0 = SumPair{rows[r.a0][lane], rows[NA + r.a0][lane]}; r.b1 = SumPair{rows[r.a1][lane], rows[NA + r.a1][lane]};
    r.c0 = lds_cnt[r.a0][lane]; r.c1 = lds_cnt[r.a1][lane];
}
// appends the two samples, writes the statistics back and records them as entries (j0, j0+1) of the quad
template <int NA>
__device__ __forceinline__ void pair_update(QuadStat& o, int j0, const PairRaw& r, LdsRow* rows, int (*lds_cnt)[WAVE], int lane) {
    double s0 = r.b0.s + r.x0, q0 = fma(r.x0, r.x0, r.b0.q);
    int n0 = r.c0 + 1;
    const bool same = (r.a0 == r.a1);
    double s1 = (same ? s0 : r.b1.s) + r.x1;
    double q1 = fma(r.x1, r.x1, same ? q0 : r.b1.q);
    int n1 = (same ? n0 : r.c1) + 1;
    rows[r.a0][lane] = s0; rows[NA + r.a0][lane] = q0; lds_cnt[r.a0][lane] = n0;
    rows[r.a1][lane] = s1; rows[NA + r.a1][lane] = q1; lds_cnt[r.a1][lane] = n1;
    o.a[j0] = r.a0; o.a[j0 + 1] = r.a1;
    o.n[j0] = n0;   o.n[j0 + 1] = n1;
    o.s[j0] = s0;   o.s[j0 + 1] = s1;
    o.q[j0] = q0;   o.q[j0 + 1] = q1;
}

// ---- the multi-wave kernel's statistics stage ------------------------------------------------------------------------------------
// Everything that does not touch the LDS is done up front (clamped ids, shifted samples, the rows' addresses): the caller puts
// quad_in BEFORE its wait for the previous quad's statistics, so that the stage the waves of a slice hand over holds LDS traffic only:
//   count_quad       four ds_add_rtn_u32 (+NS) on the buckets' counters, back to back: the LDS applies them in issue order, so records
//                    of one bucket need no forwarding and nothing waits — the counts leave the chain of round trips below;
//   prepared_append  one record at a time: read {sum, sum of squares} (one ds_read2st64_b64), add, write back — the LDS executes in
//                    order, so a read issued after the previous record's write-back sees it and no same-bucket forwarding is needed
//                    (the 2.5 selects per record of a forwarding cost more than the two extra round trips per quad: round 2; f64 LDS
//                    atomics for the sums cost 8 %: round 6, tools/experiments/atomic_statistics_stage.patch).
struct QuadIn { int a[4]; unsigned ra[4], rc[4]; double x[4]; };    // ra / rc: LDS address of the lane's element of bucket a's sum row / counter
template <int NA>
__device__ __forceinline__ void quad_in(QuadIn& in, double shift, unsigned row_base, unsigned cnt_base, const int (&act)[4], const double (&xr)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        in.a[j] = min(act[j], NA - 1);
        in.ra[j] = row_base + ((unsigned)in.a[j] << 9);
        in.rc[j] = cnt_base + ((unsigned)in.a[j] << 8);
        in.x[j] = xr[j] - shift;
    }
}
template <int NS>
__device__ __forceinline__ void count_quad(QuadStat& o, const QuadIn& in) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
        o.n[j] = (int)__hip_atomic_fetch_add((LdsU32*)(size_t)in.rc[j], (unsigned)NS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
template <int NA>
__device__ __forceinline__ void prepared_append(QuadStat& o, int j, const QuadIn& in) {
    LdsF64* ps = (LdsF64*)(size_t)in.ra[j];
    const double x = in.x[j];
    const double s = ps[0] + x, q = fma(x, x, ps[NA * WAVE]);
    ps[0] = s;
    ps[NA * WAVE] = q;
    asm volatile("" ::: "memory");
    o.a[j] = in.a[j]; o.s[j] = s; o.q[j] = q;
}

// sign mask of a key's high word as ONE v_ashrrev_i32: left to itself the compiler turns the shift into a 64-bit
// compare + selects, one VALU operation more per record in the online loop (in the final-state kernels the plain
// shift is the better choice: the asm costs them registers)
struct AsmSign {
    __device__ __forceinline__ static int of(int hi) {
        int s;
        asm("v_ashrrev_i32 %0, 31, %1" : "=v"(s) : "v"(hi));
        return s;
    }
};
template <int NA>
struct LaneState {
    double best;        // max over the keys
    double shift;       // K of the shifted sums: the state's first reward
    int latch;          // activation step (S1:98-99); >= LATCH_NEVER until the arg-max first leaves rule_act
};

// commit one evaluated record: S1:86 threshold, S1:93-95 max / first arg-max, S1:98-99 latch.
// Split in two so that the LDS traffic of four consecutive commits (write key, reload all keys) can be issued
// back to back (the LDS executes in order, so record j's reload sees records 0..j) and the four max trees then
// run on data that arrives behind ONE round trip instead of four.
// (n: the bucket's size AFTER the append, in samples)
template <int NA>
__device__ __forceinline__ void commit_issue(double (&key)[NA], KeyRow* lds_key, int lane, int a, int n,
                                             double v, const DevParams& p) {
    const double k = encode_key<AsmSign>(v, a);
    const int slot = (n > p.n_thres) ? a : NA;                    // below the threshold: the trash row
    lds_key[slot][lane] = k;
#pragma unroll
    for (int c = 0; c < NA; ++c) key[c] = lds_key[c][lane];
}
template <int NA>
__device__ __forceinline__ void commit_finish(LaneState<NA>& st, const double (&key)[NA], double& out_val, int& out_act) {
    const double best = tree_max<NA>(key);
    st.best = best;
    out_val = best;
    out_act = decode_action<AsmSign>(best);
}
// ---- the four commits of a quad in ONE pass over the keys (round 6) ----------------------------------------------------------------
// commit_issue / commit_finish re-load all NA keys and run a max tree after EVERY record: 4 x (1 write + ceil(NA/2) 16-byte reads)
// LDS operations and 4 x (NA-1) v_max_f64 per quad.  A quad touches at most four key slots.  So:
//   1. exchange the (<= 4) touched slots with KNOCKED = a finite key below every real one (ds_wrxchg_rtn_b64, in record order: the
//      returned o_j is the slot's value BEFORE the quad, or KNOCKED when an earlier record of the quad already took it);
//   2. re-load the keys ONCE: M = max over the slots the quad does not touch;
//   3. write the four new keys in record order (the last write to a slot is its newest value: the LDS executes in order);
//      -- the hand-over to the next wave of the slice happens here: nothing below touches the LDS --
//   4. max after record j = max(M, k_i of the records i <= j unless a later record i' <= j overwrote the same slot,
//                                  o_i of the records i > j):  6 compares, 6 selects (high word only), 13 + (NA-1) v_max_f64.
// A record whose bucket is still below the threshold (S1:86) goes to the trash slot with a KNOCKED key, so the trash slot holds a
// knocked key for ever and its exchange returns one.  Every max is over exactly the keys the record-by-record form sees:
// the same best key, bit for bit, hence the same step value and arg-max.
// Three pieces, so that the caller can place the hand-over waits: prepare (pure VALU: keys, slot addresses — BEFORE the wait for the
// previous quad's commit), issue (steps 1-3: LDS operations only, between the wait and the hand-over), finish (step 4, after it).
struct QuadCommit {
    unsigned addr[4];                                      // LDS byte address of the lane's element of the record's SUM row (its key row: + 2 NA rows)
    double kk[4];                                          // the record's new key, or a knocked one below the threshold
    double o[4];                                           // what the exchange returned
};
constexpr int KNOCK_HI = (int)0xffefffff;                  // high word of the most negative finite doubles: any low word will do
// (thr_raw = n_thres in the counter's unit: the size after the append exceeds n_thres  <=>  the counter BEFORE it was >= thr_raw;
//  addr = the lane's element of the SUM row of the record's bucket, or of the row that has the trash row as its key: the key itself
//  sits 2 NA rows further — an instruction offset)
template <int NA>
__device__ __forceinline__ void quad_commit_prepare(QuadCommit& qc, unsigned row_base, const QuadIn& in, const QuadStat& cur, const double (&v)[4],
                                                    int thr_raw) {
    // (a wave-uniform, sticky "every bucket of every lane is past the threshold" flag that skips the compare and the two selects below
    //  — 3 VALU operations per record less in the steady state of long streams — measured the SAME time on configs[1] and +1 % on
    //  configs[3] / [4]: profiles/r06_ab_online_all_past.txt; the kernel is not bound by its VALU count alone)
    const unsigned trash = row_base + ((unsigned)NA << 9);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool past = cur.n[j] >= thr_raw;
        // the key of a record below the threshold is knocked BEFORE the code goes in (one select on the high word; the low word
        // then carries a code nobody reads)
        const int hi = past ? __double2hiint(v[j]) : KNOCK_HI;
        qc.kk[j] = encode_key<AsmSign>(__hiloint2double(hi, __double2loint(v[j])), cur.a[j]);
        qc.addr[j] = past ? in.ra[j] : trash;
    }
}
template <int NA>
__device__ __forceinline__ void quad_commit_issue(QuadCommit& qc, double (&key)[NA], KeyRow* lds_key, int lane) {
    constexpr int KEY_OFF = 2 * NA * WAVE;                 // (in elements) from a bucket's sum row to its key row
    const unsigned long long knocked = (unsigned long long)(unsigned)KNOCK_HI << 32;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
        qc.o[j] = __longlong_as_double((long long)__hip_atomic_exchange((LdsU64*)(size_t)qc.addr[j] + KEY_OFF, knocked, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < NA; ++c) key[c] = lds_key[c][lane];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ((LdsF64*)(size_t)qc.addr[j])[KEY_OFF] = qc.kk[j];
        asm volatile("" ::: "memory");
    }
}
template <int NA>
__device__ __forceinline__ void quad_commit_finish(const QuadCommit& qc, const double (&key)[NA], double (&ov)[4], int (&oa)[4]) {
    const double M = tree_max<NA>(key);
    const unsigned* addr = qc.addr;
    const double* kk = qc.kk;
    const double* o = qc.o;
    // k_i as it still stands after record j: knocked once a later record wrote the same slot
    const bool e01 = addr[0] == addr[1], e02 = addr[0] == addr[2], e03 = addr[0] == addr[3], e12 = addr[1] == addr[2],
               e13 = addr[1] == addr[3], e23 = addr[2] == addr[3];
    auto knock = [](bool c, double k) __attribute__((always_inline)) { return __hiloint2double(c ? KNOCK_HI : __double2hiint(k), __double2loint(k)); };
    const double k0_1 = knock(e01, kk[0]), k0_2 = knock(e02, k0_1), k0_3 = knock(e03, k0_2);
    const double k1_2 = knock(e12, kk[1]), k1_3 = knock(e13, k1_2);
    const double k2_3 = knock(e23, kk[2]);
    const double P3 = fmax(M, o[3]), P23 = fmax(P3, o[2]), P123 = fmax(P23, o[1]);
    ov[0] = fmax(P123, kk[0]);
    ov[1] = fmax(P23, fmax(k0_1, kk[1]));
    ov[2] = fmax(P3, fmax(fmax(k0_2, k1_2), kk[2]));
    ov[3] = fmax(M, fmax(fmax(k0_3, k1_3), fmax(k2_3, kk[3])));
#pragma unroll
    for (int j = 0; j < 4; ++j) oa[j] = decode_action<AsmSign>(ov[j]);
}

// (maximum, runner-up) of N keys: a tournament of (hi, lo) pairs, 4 operations per merge
template <int N>
__device__ __forceinline__ void top2(const double* k, double& hi, double& lo) {
    if constexpr (N == 1) { hi = k[0]; lo = -__builtin_huge_val(); }
    else if constexpr (N == 2) { hi = fmax(k[0], k[1]); lo = fmin(k[0], k[1]); }
    else {
        double h1, l1, h2, l2;
        top2<N / 2>(k, h1, l1);
        top2<N - N / 2>(k + N / 2, h2, l2);
        hi = fmax(h1, h2);
        const double m = fmin(h1, h2);
        if constexpr (N / 2 == 1) lo = fmax(m, l2);
        else lo = fmax(m, fmax(l1, l2));
    }
}
// S1:98-99 latch: first step whose arg-max is not the rule action.  Values >= LATCH_NEVER mean "not yet".
constexpr int LATCH_NEVER = 0x10000000;
__device__ __forceinline__ void latch_record(int& latch, int b, int t, const DevParams& p) {
    latch = min(latch, (b != p.rule_act) ? t + 1 : 0x7fffffff);
}
// the same for the four records of a quad from their packed arg-max bytes: the first byte that differs from the rule
// action is found with one v_ffbl_b32 (which returns -1 for "none": (-1 >> 3) + t is >= LATCH_NEVER)
__device__ __forceinline__ void latch_quad(int& latch, unsigned packed_actions, unsigned rule4, int t0) {
    unsigned first;
    asm("v_ffbl_b32 %0, %1" : "=v"(first) : "v"(packed_actions ^ rule4));
    latch = min(latch, (int)(first >> 3) + t0 + 1);
}
template <int NA>
__device__ __forceinline__ void commit_record(LaneState<NA>& st, KeyRow* lds_key, int lane, int a, int n,
                                              double v, int t, const DevParams& p, double& out_val, int& out_act) {
    double key[NA];
    commit_issue<NA>(key, lds_key, lane, a, n, v, p);
    commit_finish<NA>(st, key, out_val, out_act);
    latch_record(st.latch, out_act, t, p);
}

// Tail: some lanes' streams have ended.  Same arithmetic, one record at a time under the lane's own guard.  NS = the counters' unit
// (samples x NS: the multi-wave kernel counts in steps of 16, trace_nwave_impl.h).
template <int NA, int NS = 1>
__device__ __forceinline__ void guarded_record(LaneState<NA>& st, LdsRow* rows, int (*lds_cnt)[WAVE], int lane, int a_in, double x_raw, int t,
                                               const DevParams& p, double& out_val, int& out_act) {
    const int a = min(a_in, NA - 1);
    const double x = x_raw - st.shift;
    const int c = lds_cnt[a][lane] + NS;
    const double s = rows[a][lane] + x, q = fma(x, x, rows[NA + a][lane]);
    rows[a][lane] = s;
    rows[NA + a][lane] = q;
    lds_cnt[a][lane] = c;
    const int n = NS == 1 ? c : (int)((unsigned)c / (unsigned)NS);
    const double v = value_from_sums(n, s, q, st.shift, a == p.rule_act, p);
    commit_record<NA>(st, key_rows_of<NA>(rows), lane, a, n, v, t, p, out_val, out_act);
}

}  // namespace dcarl
