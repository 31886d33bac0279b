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

// ---- the fast path: four consecutive records (a "quad") of one state, every lane live, as a software pipeline ---
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
struct QuadStat { int a[4], n[4]; double s[4], q[4]; };   // bucket statistics right after each of the four appends

// The statistics stage works on PAIRS of records: both bucket reads are issued together and, if the two records hit
// the same bucket, the second takes the first one's updated statistics from registers (one compare + five selects).
// Doing all four records of a quad at once would need six such checks; the second pair's reads are simply issued
// after the first pair's write-back, so they already see it.
struct PairRaw { int a0, a1; double x0, x1; SumPair b0, b1; int c0, c1; };

template <int NA>
__device__ __forceinline__ void pair_read(PairRaw& r, double shift, SumPair (*lds_sum)[WAVE], int (*lds_cnt)[WAVE],
                                          int lane, int act0, int act1, double xr0, double xr1) {
    r.a0 = min(act0, NA - 1); r.a1 = min(act1, NA - 1);
    r.x0 = xr0 - shift; r.x1 = xr1 - shift;
    r.b0 = lds_sum[r.a0][lane]; r.b1 = lds_sum[r.a1][lane];
    r.c0 = lds_cnt[r.a0][lane]; r.c1 = lds_cnt[r.a1][lane];
}
// appends the two samples, writes the statistics back and records them as entries (j0, j0+1) of the quad
__device__ __forceinline__ void pair_update(QuadStat& o, int j0, const PairRaw& r, SumPair (*lds_sum)[WAVE],
                                            int (*lds_cnt)[WAVE], int lane) {
    double s0 = r.b0.s + r.x0, q0 = fma(r.x0, r.x0, r.b0.q);
    int n0 = r.c0 + 1;
    const bool same = (r.a0 == r.a1);
    double s1 = (same ? s0 : r.b1.s) + r.x1;
    double q1 = fma(r.x1, r.x1, same ? q0 : r.b1.q);
    int n1 = (same ? n0 : r.c1) + 1;
    lds_sum[r.a0][lane] = SumPair{s0, q0}; lds_cnt[r.a0][lane] = n0;
    lds_sum[r.a1][lane] = SumPair{s1, q1}; lds_cnt[r.a1][lane] = n1;
    o.a[j0] = r.a0; o.a[j0 + 1] = r.a1;
    o.n[j0] = n0;   o.n[j0 + 1] = n1;
    o.s[j0] = s0;   o.s[j0 + 1] = s1;
    o.q[j0] = q0;   o.q[j0 + 1] = q1;
}

// one record at a time: the LDS executes in order, so a read issued after the previous record's write-back sees it and no
// same-bucket forwarding is needed (2.5 selects per record less, two more LDS round trips per quad)
template <int NA>
__device__ __forceinline__ void single_append(QuadStat& o, int j, double shift, SumPair (*lds_sum)[WAVE], int (*lds_cnt)[WAVE],
                                              int lane, int act, double xr) {
    const int a = min(act, NA - 1);
    const double x = xr - shift;
    const SumPair b = lds_sum[a][lane];
    const int n = lds_cnt[a][lane] + 1;
    const double s = b.s + x, q = fma(x, x, b.q);
    lds_sum[a][lane] = SumPair{s, q};
    lds_cnt[a][lane] = n;
    asm volatile("" ::: "memory");
    o.a[j] = a; o.n[j] = n; o.s[j] = s; o.q[j] = q;
}

// The NA current values V[s][.] are kept in LDS as tie-break-coded f64 keys, two per 16-byte cell
// ([a/2][lane][a&1]): overwriting key[a] for a per-lane action id is ONE ds_write_b64 (registers cannot be indexed
// per lane; the register version needed a v_cmp + 2 v_cndmask per candidate, ~5.6 cycles each at 1 wave/SIMD),
// and the arg-max reloads all keys with ceil(NA/2) ds_read_b128.  Slot NA is a trash slot: records whose bucket is
// still below the threshold (S1:86) write there (for odd NA it is the free half of the last cell, which keeps the
// 11-candidate instance at 20 224 B of LDS = 8 resident blocks per CU).
struct __attribute__((aligned(16))) KeyPair { double k0, k1; };

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
template <int NA> constexpr int key_cells() { return NA / 2 + 1; }     // slots 0..NA-1 = candidates, slot NA = trash

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
template <int NA>
__device__ __forceinline__ void commit_issue(double (&key)[NA], KeyPair (*lds_key)[WAVE], int lane, int a, int n,
                                             double v, const DevParams& p) {
    const double k = encode_key<AsmSign>(v, a);
    const int slot = (n > p.n_thres) ? a : NA;                    // below the threshold: the trash slot
    // byte offset of key `slot` inside [slot/2][lane][slot&1]: (slot/2)*1024 + (slot&1)*8, as ONE multiply and mask:
    // slot*0x208 = slot*512 + slot*8 puts slot/2 at bit 10 and slot&1 at bit 3 (plus bits the mask drops)
    const unsigned off = ((unsigned)slot * 0x208u) & 0xfc08u;
    *reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(&lds_key[0][lane]) + off) = k;
#pragma unroll
    for (int c = 0; c < (NA + 1) / 2; ++c) {
        const KeyPair kp = lds_key[c][lane];
        key[2 * c] = kp.k0;
        if (2 * c + 1 < NA) key[2 * c + 1] = kp.k1;
    }
}
template <int NA>
__device__ __forceinline__ void commit_finish(LaneState<NA>& st, const double (&key)[NA], double& out_val, int& out_act) {
    const double best = tree_max<NA>(key);
    st.best = best;
    out_val = best;
    out_act = decode_action<AsmSign>(best);
}
// ---- the arg-max without re-reading the keys (three-wave kernel) -------------------------------------------------------
// Re-loading all NA keys for every record (ceil(NA/2) ds_read_b128) is 48 of the ~82 LDS cycles a record costs, and the LDS
// is what four slices per CU saturate.  A record changes ONE key, so the state's maximum can be carried along:
//     best = max_i key[i]  (exact; its 5 code bits name the leader),     u >= max_{i != leader} key[i]  (an upper bound)
//   record for candidate a, new key k (buckets still below the threshold change nothing):
//     a != leader:  best' = max(best, k);  u' = max(u, min(k, best))      -- exact again: if k takes the lead the old leader
//                                                                            is the runner-up, otherwise k joins the rest
//     a == leader:  k > u  -> best' = k, still the leader (everything else is <= u);   u unchanged
//                   else   -> unknown: RE-SCAN the keys for the exact (maximum, runner-up)
// u only loosens between re-scans, and a re-scan (any lane of the wavefront needs one -> the wavefront does it, every lane
// takes the exact pair) resets it.  Measured on the headline streams: 8 % of the records re-scan (Sim1's four best candidates
// lie within 1 of each other, the leader changes for ever); random Q*: 2 %.  best is always one of the keys, bit for bit, so
// the trace is identical to the full arg-max.  The pair lives in LDS ({best, u} per lane) because the three waves of a slice
// take turns: read once per quad, carried in registers across its four records, written back before the hand-over.
struct __attribute__((aligned(16))) BestPair { double best, u; };

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
template <int NA> constexpr int lazy_key_cells() { return (NA + 1) / 2; }     // cells holding keys; cell lazy_key_cells is {best, u}

// one record: write the key (below the threshold: to the lane's trash word -- an exec-masked store costs this kernel 13 %),
// carry (best, u), re-scan if some lane must.  key_addr / trash_addr: LDS byte addresses of lds_key[0][lane] and of the
// lane's trash word.
typedef __attribute__((address_space(3))) double LdsDouble;
template <int NA>
__device__ __forceinline__ void lazy_commit(double& best, double& u, int& lead, KeyPair (*lds_key)[WAVE], int lane,
                                            unsigned key_addr, unsigned trash_addr, int a, int n, double v, const DevParams& p) {
    constexpr int KC = lazy_key_cells<NA>();
    const double k = encode_key<AsmSign>(v, a);
    // The lane predicates live in SGPR pairs and feed v_cndmask directly (hand-written: the compiler turns every reuse of a
    // predicate into a v_cndmask 0/1 + v_cmp pair and selects an f64 through two predicates with four v_cndmask):
    //   mv = records past the threshold (S1:86), ml = ... that belong to the leader, mo = ... to another candidate
    unsigned long long mv, ml, mo;
    asm("v_cmp_lt_i32 %0, %3, %4\n\t"
        "v_cmp_eq_u32 vcc, %5, %6\n\t"
        "s_and_b64 %1, vcc, %0\n\t"
        "s_andn2_b64 %2, %0, vcc"
        : "=&s"(mv), "=&s"(ml), "=&s"(mo) : "s"(p.n_thres), "v"(n), "v"(a), "v"(lead) : "vcc", "scc");
    // byte offset of key a inside [a/2][lane][a&1]: (a/2)*1024 + (a&1)*8 = (a*0x208) & 0xfc08
    const unsigned kaddr = key_addr + (((unsigned)a * 0x208u) & 0xfc08u);
    unsigned waddr;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(waddr) : "v"(trash_addr), "v"(kaddr), "s"(mv));
    *reinterpret_cast<LdsDouble*>((size_t)waddr) = k;
    const int klo = __double2loint(k), khi = __double2hiint(k);
    int kol, koh;                                                // ko = other ? k : -inf
    asm("v_cndmask_b32 %0, 0, %2, %4\n\t"
        "v_cndmask_b32 %1, %5, %3, %4"
        : "=&v"(kol), "=&v"(koh) : "v"(klo), "v"(khi), "s"(mo), "v"((int)0xfff00000));
    const double ko = __hiloint2double(koh, kol);
    u = fmax(u, fmin(ko, best));
    const double nb = fmax(best, ko);
    int bl, bh;                                                  // best = is_lead ? k : max(best, ko)
    asm("v_cndmask_b32 %0, %2, %4, %6\n\t"
        "v_cndmask_b32 %1, %3, %5, %6"
        : "=&v"(bl), "=&v"(bh) : "v"(__double2loint(nb)), "v"(__double2hiint(nb)), "v"(klo), "v"(khi), "s"(ml));
    best = __hiloint2double(bh, bl);
    unsigned long long need;                                     // the leader's new key does not clear the bound: re-scan
    asm("v_cmp_ngt_f64 vcc, %1, %2\n\t"
        "s_and_b64 %0, vcc, %3"
        : "=s"(need) : "v"(k), "v"(u), "s"(ml) : "vcc", "scc");
    if (need != 0ull) {
        double key[2 * KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const KeyPair kp = lds_key[c][lane];
            key[2 * c] = kp.k0;
            key[2 * c + 1] = kp.k1;
        }
        top2<2 * KC>(key, best, u);
    }
    lead = decode_action<AsmSign>(best);
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
__device__ __forceinline__ void commit_record(LaneState<NA>& st, KeyPair (*lds_key)[WAVE], int lane, int a, int n,
                                              double v, int t, const DevParams& p, double& out_val, int& out_act) {
    double key[NA];
    commit_issue<NA>(key, lds_key, lane, a, n, v, p);
    commit_finish<NA>(st, key, out_val, out_act);
    latch_record(st.latch, out_act, t, p);
}

// Tail: some lanes' streams have ended.  Same arithmetic, one record at a time under the lane's own guard.
template <int NA>
__device__ __forceinline__ void guarded_record(LaneState<NA>& st, SumPair (*lds_sum)[WAVE], int (*lds_cnt)[WAVE],
                                               KeyPair (*lds_key)[WAVE], int lane, int a_in, double x_raw, int t, const DevParams& p,
                                               double& out_val, int& out_act) {
    const int a = min(a_in, NA - 1);
    const double x = x_raw - st.shift;
    SumPair sp = lds_sum[a][lane];
    const int n = lds_cnt[a][lane] + 1;
    sp.s += x;
    sp.q = fma(x, x, sp.q);
    lds_sum[a][lane] = sp;
    lds_cnt[a][lane] = n;
    const double v = value_from_sums(n, sp.s, sp.q, st.shift, a == p.rule_act, p);
    commit_record<NA>(st, lds_key, lane, a, n, v, t, p, out_val, out_act);
}

}  // namespace dcarl
