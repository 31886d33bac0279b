// Two by-products of the online loop that are not part of its hot kernel:
//   * the THIRD per-record trace of the reference, true_step_TSRL_value[idx].append(true_action_values[idx][TSRL_act])
//     (S1:96 / S2:94): a gather of Q*[state][step_act] over the step_act trace the online kernel wrote;
//   * the top-2 gap census SURVEY.md section 7 asks for next to every parity run: how far the arg-max of S1:93-94 is from
//     flipping, and how many comparisons fall inside the 32-ulp block in which this library's tie-break code (common.h
//     encode_key) — not the values — orders two candidates.
#include "trace_common.h"

namespace dcarl {

// ---- true_step_TSRL_value -----------------------------------------------------------------------------------------------------
// out[e(s,t)] = Q[state(s)][step_act[e(s,t)]] in the sliced layout of step_act.  One wavefront = one slice (lane = slot), a block =
// four wavefronts walking the slice's quad rows round-robin; the 64 Q rows of the slice sit in LDS as [a][lane] (a per-lane
// action id is an LDS address).  HBM: 1 B read + sizeof(T) written per record.
template <typename T>
__global__ __launch_bounds__(256) void true_step_kernel(const uint8_t* __restrict__ step_act, const int64_t* __restrict__ slice_row_off,
                                                        const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, int S, int A,
                                                        const double* __restrict__ Q, int q_rows, T* __restrict__ out, int chunks) {
    using Q4 = typename Quad<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* qt = reinterpret_cast<double*>(smem);                   // [A][WAVE]
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const int w = blockIdx.x / chunks, ch = blockIdx.x - w * chunks;
    const int s = w * WAVE + lane;
    const int so = (s < S) ? (slot_state ? slot_state[s] : s) : 0;
    if (wv == 0) {
        const double* row = Q + (q_rows == 1 ? 0 : (int64_t)so * A);
        for (int a = 0; a < A; ++a) qt[a * WAVE + lane] = (s < S) ? row[a] : 0.0;
    }
    __syncthreads();
    const int64_t row0 = slice_row_off[w];
    const int rows = (int)(slice_row_off[w + 1] - row0);
    const int my_len = (s < S) ? min(len[s], rows) : 0;
    int max_len = my_len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, __shfl_xor(max_len, off));
    const int nquads = (__builtin_amdgcn_readfirstlane(max_len) + 3) >> 2;
    const unsigned* Aq = reinterpret_cast<const unsigned*>(step_act) + row0 / 4 * WAVE + lane;
    Q4* Oq = reinterpret_cast<Q4*>(out) + row0 / 4 * WAVE + lane;
    const int amax_id = A - 1;
    for (int q = ch * 4 + wv; q < nquads; q += 4 * chunks) {
        if (q * 4 < my_len) {
            const unsigned av = Aq[(int64_t)q * WAVE];
            T v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a = min((int)((av >> (8 * j)) & 255u), amax_id);
                v[j] = (q * 4 + j < my_len) ? (T)qt[a * WAVE + lane] : T(0);
            }
            Q4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
            Oq[(int64_t)q * WAVE] = o;
        }
    }
}

template <typename T>
int launch_true_step(const uint8_t* step_act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int S, int A,
                     const double* Q, int q_rows, int64_t total_rows, T* out, hipStream_t st) {
    const int W = slices_of(S);
    if (W == 0) return 0;
    // enough blocks to fill the chip whatever the table's shape: a slice's rows are cut into `chunks` interleaved pieces
    const int64_t mean_quads = total_rows / 4 / W;
    int chunks = (int)((8192 + W - 1) / W);
    if (chunks > mean_quads / 16) chunks = (int)(mean_quads / 16);
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL((true_step_kernel<T>), dim3((unsigned)W * chunks), dim3(256), (unsigned)A * WAVE * 8u, st, step_act, slice_row_off, len,
                       slot_state, S, A, Q, q_rows, out, chunks);
    return 0;
}
template int launch_true_step<float>(const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const double*, int, int64_t, float*,
                                     hipStream_t);
template int launch_true_step<double>(const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int, const double*, int, int64_t,
                                      double*, hipStream_t);

// ---- top-2 gap census ---------------------------------------------------------------------------------------------------------
// out (u64 [DCARL_CENSUS_WORDS], accumulated: the caller zeroes it, except word 67 which starts at ~0):
//   [0..63]  histogram of the RELATIVE gap (best - runner_up) / |best| of the stripped values: bin b holds 2^(b-53) <= rel < 2^(b-52);
//            bin 0 also takes everything below (exact ties included), bin 63 everything from 2^10 up
//   [64]     arg-max evaluations counted (online: one per record; table: one per state)
//   [65]     ... whose top two keys share a 32-ulp block (equal once the 5 code bits are cleared): ordered by candidate id, not by value
//   [66]     ... of those, both still at the never-evaluated prior init_other (a true tie: the reference's first-max rule picks the
//            lower id too, S1:51,94)
//   [67]     bit pattern of the smallest relative gap among the evaluations NOT counted in [65]
//   [68]     evaluations without a runner-up (A == 1)
struct CensusLane {
    unsigned same = 0, prior = 0, lone = 0, evals = 0;
    unsigned long long min_rel = ~0ull;
};
__device__ __forceinline__ void census_pair(CensusLane& c, unsigned* hist, double hi_key, double lo_key, double prior_key_stripped) {
    const double vh = strip_code(hi_key), vl = strip_code(lo_key);
    const double rel = (vh - vl) / fmax(fabs(vh), 1e-300);
    const int e = ((__double2hiint(rel) >> 20) & 0x7ff) - 1023;
    atomicAdd(&hist[min(max(e + 53, 0), 63)], 1u);
    c.evals++;
    if (__double_as_longlong(vh) == __double_as_longlong(vl)) {
        c.same++;
        if (__double_as_longlong(vh) == __double_as_longlong(prior_key_stripped)) c.prior++;
    } else {
        c.min_rel = min(c.min_rel, (unsigned long long)__double_as_longlong(rel));      // rel > 0: ordered like its bit pattern
    }
}
__device__ __forceinline__ void census_flush(const CensusLane& c, const unsigned* hist, unsigned long long* out) {
    __syncthreads();
    for (int b = threadIdx.x; b < 64; b += blockDim.x)
        if (hist[b]) atomicAdd(&out[b], (unsigned long long)hist[b]);
    unsigned long long same = c.same, prior = c.prior, lone = c.lone, evals = c.evals, mn = c.min_rel;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        same += __shfl_xor(same, off); prior += __shfl_xor(prior, off); lone += __shfl_xor(lone, off); evals += __shfl_xor(evals, off);
        mn = min(mn, (unsigned long long)__shfl_xor(mn, off));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        if (evals) atomicAdd(&out[64], evals);
        if (same) atomicAdd(&out[65], same);
        if (prior) atomicAdd(&out[66], prior);
        if (mn != ~0ull) atomicMin(&out[67], mn);
        if (lone) atomicAdd(&out[68], lone);
    }
}

// final-state mode: the table V [S*A] as dcarl_trace_* / dcarl_bounds_csr_* return it (code bits cleared; encode_key puts the same
// code back: encode_key(strip_code(key)) == key, so these are the keys the kernels compared)
__global__ __launch_bounds__(256) void census_table_kernel(const double* __restrict__ V, int S, int A, DevParams p, unsigned long long* out) {
    __shared__ unsigned hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    CensusLane c;
    const double prior = strip_code(encode_key(p.init_other, 0));
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < S; s += (int64_t)gridDim.x * blockDim.x) {
        double hi = encode_key(V[s * A], 0), lo = -__builtin_huge_val();
        for (int a = 1; a < A; ++a) {
            const double k = encode_key(V[s * A + a], a);
            lo = fmax(lo, fmin(hi, k));
            hi = fmax(hi, k);
        }
        if (A == 1) { c.lone++; c.evals++; }
        else census_pair(c, hist, hi, lo, prior);
    }
    census_flush(c, hist, out);
}

// online mode: the loop itself, one wavefront per slice, lane = state, record by record through guarded_record (trace_common.h:
// the arithmetic of every online kernel's tail, bit for bit the keys their fast paths hold), and after every record the two largest
// of the state's keys.  Not a fast kernel (no pipelining; ~5x the online kernel): it runs next to parity tests and once per bench line.
template <typename T, int NA>
__global__ __launch_bounds__(WAVE) void census_trace_kernel(const T* __restrict__ R, const uint8_t* __restrict__ act,
                                                            const int64_t* __restrict__ slice_row_off, const int32_t* __restrict__ len, int S, int A,
                                                            DevParams p, unsigned long long* out) {
    using Q4 = typename Quad<T>::type;
    constexpr int NP = key_cells<NA>();
    __shared__ __attribute__((aligned(16))) double lds_rows[2 * NA + 2 * NP][WAVE];   // sums, sums of squares, keys (trace_common.h)
    __shared__ int lds_cnt[NA][WAVE];
    LdsRow* lrows = lds_rows;
    KeyRow* lds_key = key_rows_of<NA>(lrows);
    __shared__ unsigned hist[64];
    const int lane = threadIdx.x, w = blockIdx.x, s = w * WAVE + lane;
    hist[lane] = 0;
    const int64_t row0 = slice_row_off[w];
    const int rows = (int)(slice_row_off[w + 1] - row0);
    const int my_len = (s < S) ? min(len[s], rows) : 0;
    int max_len = my_len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, __shfl_xor(max_len, off));
    max_len = __builtin_amdgcn_readfirstlane(max_len);
#pragma unroll
    for (int a = 0; a < NA; ++a) { lrows[a][lane] = 0.0; lrows[NA + a][lane] = 0.0; lds_cnt[a][lane] = 0; }
    LaneState<NA> st;
    {
        double key[2 * NP];
#pragma unroll
        for (int a = 0; a < 2 * NP; ++a) {
            const double v0 = a == p.rule_act ? p.init_rule : p.init_other;
            key[a] = (a < A) ? encode_key(v0, a) : encode_key(-1e300, a & 31);
        }
#pragma unroll
        for (int c = 0; c < 2 * NP; ++c) lds_key[c][lane] = key[c];
        st.best = tree_max<NA>(key);
    }
    st.latch = 0x7fffffff;
    st.shift = (my_len > 0) ? (double)R[(row0 * WAVE) + lane * 4] : 0.0;
    const Q4* Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
    const uchar4* Aq = reinterpret_cast<const uchar4*>(act) + row0 / 4 * WAVE + lane;
    CensusLane c;
    const double prior = strip_code(encode_key(p.init_other, 0));
    const int nquads = (max_len + 3) >> 2;
    for (int qi = 0; qi < nquads; ++qi) {
        if (qi * 4 < my_len) {
            const Q4 rv = Rq[(int64_t)qi * WAVE];
            const uchar4 av = Aq[(int64_t)qi * WAVE];
            const double xr[4] = {(double)rv.x, (double)rv.y, (double)rv.z, (double)rv.w};
            const int aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (qi * 4 + j < my_len) {
                    double ov;
                    int oa;
                    guarded_record<NA>(st, lrows, lds_cnt, lane, aa[j], xr[j], qi * 4 + j, p, ov, oa);
                    double key[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) key[a] = lds_key[a][lane];
                    if (A == 1) { c.lone++; c.evals++; }
                    else {
                        double hi, lo;
                        top2<NA>(key, hi, lo);
                        census_pair(c, hist, hi, lo, prior);
                    }
                }
            }
        }
    }
    census_flush(c, hist, out);
}

int launch_census_table(const double* V, int S, int A, const DevParams& p, unsigned long long* out, hipStream_t st) {
    if (S == 0) return 0;
    const int blocks = (int)min((int64_t)4096, ((int64_t)S + 255) / 256);
    hipLaunchKernelGGL(census_table_kernel, dim3(blocks), dim3(256), 0, st, V, S, A, p, out);
    return 0;
}

template <typename T>
int launch_census_trace(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, int S, int A, const DevParams& p,
                        unsigned long long* out, hipStream_t st) {
    const int W = slices_of(S);
    if (W == 0) return 0;
    // key registers / LDS rows: the next compiled candidate count (padding candidates hold -1e300 and never enter the top two unless A == 1)
#define DCARL_CENSUS_CASE(NA) \
    if (A <= NA) { hipLaunchKernelGGL((census_trace_kernel<T, NA>), dim3(W), dim3(WAVE), 0, st, R, act, slice_row_off, len, S, A, p, out); return 0; }
    DCARL_CENSUS_CASE(4) DCARL_CENSUS_CASE(8) DCARL_CENSUS_CASE(12) DCARL_CENSUS_CASE(16) DCARL_CENSUS_CASE(24) DCARL_CENSUS_CASE(32)
#undef DCARL_CENSUS_CASE
    return -1;
}
template int launch_census_trace<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, int, int, const DevParams&, unsigned long long*,
                                        hipStream_t);
template int launch_census_trace<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, int, int, const DevParams&,
                                         unsigned long long*, hipStream_t);

}  // namespace dcarl
