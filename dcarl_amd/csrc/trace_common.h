// Pieces shared by the online ("trace") kernels: the quad-packed record types and the statistics stage (S1:80).
#pragma once
#include "common.h"

namespace dcarl {

template <typename T> struct Quad;
template <> struct Quad<float> { using type = float4; };
template <> struct Quad<double> { using type = double4; };

struct __attribute__((aligned(16))) SumPair { double s, q; };

// ---- the fast path: four consecutive records of one state, every lane live, as a software pipeline ------------
//   A1(q+1) issue the LDS reads of the next quad's buckets            (S1:80, statistics)
//   B(q)    four independent f64 evaluations                           (S1:87-90)
//   C1(q)   LDS traffic of the four commits, back to back              (S1:86, write key / reload keys)
//   A2(q+1) forward same-bucket statistics in registers, add the samples, write back
//   C2(q)   four max trees, arg-max decode, latch                      (S1:93-99)
// so that every LDS round trip completes behind VALU work of another stage (at 1-2 waves per SIMD instruction-
// level overlap is the only latency hiding there is).  The LDS executes in order, which is what makes the reads of
// A1(q+1) see the writes of A2(q) and the reload of commit j see the keys of commits 0..j.
struct QuadRaw {          // A1: bucket statistics as read from LDS, before forwarding
    int a0, a1, a2, a3;
    double x0, x1, x2, x3;
    SumPair b0, b1, b2, b3;
    int c0, c1, c2, c3;
};
struct QuadStat { int a[4], n[4]; double s[4], q[4]; };   // A2: statistics after the four appends

template <int NA>
__device__ __forceinline__ void stage_a1(QuadRaw& r, double shift, SumPair (*lds_sum)[WAVE], int (*lds_cnt)[WAVE],
                                         int lane, const uchar4& av, const double (&xr)[4]) {
    r.a0 = min((int)av.x, NA - 1); r.a1 = min((int)av.y, NA - 1); r.a2 = min((int)av.z, NA - 1); r.a3 = min((int)av.w, NA - 1);
    r.x0 = xr[0] - shift; r.x1 = xr[1] - shift; r.x2 = xr[2] - shift; r.x3 = xr[3] - shift;
    r.b0 = lds_sum[r.a0][lane]; r.b1 = lds_sum[r.a1][lane]; r.b2 = lds_sum[r.a2][lane]; r.b3 = lds_sum[r.a3][lane];
    r.c0 = lds_cnt[r.a0][lane]; r.c1 = lds_cnt[r.a1][lane]; r.c2 = lds_cnt[r.a2][lane]; r.c3 = lds_cnt[r.a3][lane];
}

// All four bucket reads were issued together; a later record of the same bucket takes the earlier record's updated
// statistics from registers (scalars, not arrays: the select-forwarding must stay in VGPRs).  Additions happen in
// arrival order, so the sums are bit-identical to a record-by-record update.
__device__ __forceinline__ void stage_a2(QuadStat& o, const QuadRaw& r, SumPair (*lds_sum)[WAVE], int (*lds_cnt)[WAVE],
                                         int lane) {
    const int a0 = r.a0, a1 = r.a1, a2 = r.a2, a3 = r.a3;
    const double x0 = r.x0, x1 = r.x1, x2 = r.x2, x3 = r.x3;
    double s0 = r.b0.s, q0 = r.b0.q, s1 = r.b1.s, q1 = r.b1.q, s2 = r.b2.s, q2 = r.b2.q, s3 = r.b3.s, q3 = r.b3.q;
    int n0 = r.c0, n1 = r.c1, n2 = r.c2, n3 = r.c3;
#define DCARL_UPD(j) { n##j += 1; s##j += x##j; q##j = fma(x##j, x##j, q##j); }
#define DCARL_FWD(j, i) { const bool same = (a##i == a##j); s##j = same ? s##i : s##j; q##j = same ? q##i : q##j; \
                          n##j = same ? n##i : n##j; }
    DCARL_UPD(0)
    DCARL_FWD(1, 0) DCARL_UPD(1)
    DCARL_FWD(2, 0) DCARL_FWD(2, 1) DCARL_UPD(2)
    DCARL_FWD(3, 0) DCARL_FWD(3, 1) DCARL_FWD(3, 2) DCARL_UPD(3)
#undef DCARL_UPD
#undef DCARL_FWD
    lds_sum[a0][lane] = SumPair{s0, q0}; lds_cnt[a0][lane] = n0;
    lds_sum[a1][lane] = SumPair{s1, q1}; lds_cnt[a1][lane] = n1;
    lds_sum[a2][lane] = SumPair{s2, q2}; lds_cnt[a2][lane] = n2;
    lds_sum[a3][lane] = SumPair{s3, q3}; lds_cnt[a3][lane] = n3;
    o.a[0] = a0; o.a[1] = a1; o.a[2] = a2; o.a[3] = a3;
    o.n[0] = n0; o.n[1] = n1; o.n[2] = n2; o.n[3] = n3;
    o.s[0] = s0; o.s[1] = s1; o.s[2] = s2; o.s[3] = s3;
    o.q[0] = q0; o.q[1] = q1; o.q[2] = q2; o.q[3] = q3;
}

}  // namespace dcarl
