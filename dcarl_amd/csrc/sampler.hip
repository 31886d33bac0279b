// Monte-Carlo return sampler: replaces data_sampling.py's add_an_act_data (DS:5-9), random_state_norm
// (DS:12-17) and the Data_Generation loop (DS:45-55) with a counter-based generator (Philox-4x32-10,
// Salmon et al. SC'11) + Box-Muller, so any record can be (re)generated independently by any lane on any GPU.
// Pure streaming writes: 5 B (trace layout) or 12 B ({s,a,R} pairs) per sample; ALU: 10 Philox rounds.
#include "common.h"
#ifndef DCARL_SAMPLER_NT
// the samplers' outputs are written once and read by a later kernel: non-temporal stores.  Same-box A/B of two builds on two boxes
// (tools/experiments/ab_nt_legs.sh, tools/experiments/ab_sampler_nt.sh): 2^30 pairs 2.60-2.77 -> 2.32-2.47 ms on one, 3.24 -> 3.37 on the other (a box on which this
// 12.9-GB stream is slow either way); 2^28 pairs 0.61 -> 0.59 there
#define DCARL_SAMPLER_NT 1
#endif
#include "philox.h"

namespace dcarl {

// This block's CONTIGUOUS range [lo, hi) of `total` work items (a multiple of 256 long), instead of a grid stride: with a grid stride the
// resident blocks write inside one compact, lockstep-advancing window per output array, and whether those windows collide on DRAM banks
// depends on where the allocator put the arrays (sample_pairs_kernel below: 2.17 ... 3.40 ms over twelve placements; with ranges 2.19 ... 2.50).
__device__ __forceinline__ void block_range(int64_t total, int64_t& lo, int64_t& hi) {
    // (ranges that are not a power of two long — + 256 or + 768 items — measured no better: 2.17 ... 2.87 / 2.16 ... 2.57)
    const int64_t per = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    lo = (int64_t)blockIdx.x * per;
    hi = lo + per < total ? lo + per : total;
}

__global__ __launch_bounds__(256) void sample_state_records_kernel(
    const float* __restrict__ Q, int q_rows, int S, int A, int64_t T, float sigma, uint32_t k0, uint32_t k1,
    uint32_t stream_id, float* __restrict__ R, uint8_t* __restrict__ act) {
    const int64_t Tq = (T + 3) >> 2;
    const int64_t W = slices_of(S);
    const int64_t total = W * Tq * WAVE;
    int64_t g_lo, g_hi;
    block_range(total, g_lo, g_hi);
    for (int64_t g = g_lo + threadIdx.x; g < g_hi; g += blockDim.x) {
        const int lane = (int)(g & (WAVE - 1));
        const int64_t quad = g >> 6;
        const int64_t w = quad / Tq, qi = quad - w * Tq;
        const int64_t s = w * WAVE + lane;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        int av[4] = {0, 0, 0, 0};
        if (s < S) {
            const float* q = Q + (q_rows == 1 ? 0 : s * A);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t t = qi * 4 + j;
                if (t < T) {
                    const U4 x = philox4x32_10((uint32_t)t, (uint32_t)s, stream_id, 0u, k0, k1);
                    const int a = (int)__umulhi(x.x0, (uint32_t)A);            // DS:54 uniform action
                    const float z = bm_radius(x.x1) * __builtin_amdgcn_cosf(unit_open(x.x2));
                    rv[j] = fmaf(sigma, z, q[a]);                               // DS:9  Q[act] + 50*z
                    av[j] = a;
                }
            }
        }
#if DCARL_SAMPLER_NT
        nt_store16(reinterpret_cast<float4*>(R) + g, __float_as_uint(rv[0]), __float_as_uint(rv[1]), __float_as_uint(rv[2]), __float_as_uint(rv[3]));
        nt_store4(reinterpret_cast<uchar4*>(act) + g, (unsigned)av[0] | ((unsigned)av[1] << 8) | ((unsigned)av[2] << 16) | ((unsigned)av[3] << 24));
#else
        reinterpret_cast<float4*>(R)[g] = make_float4(rv[0], rv[1], rv[2], rv[3]);
        reinterpret_cast<uchar4*>(act)[g] = make_uchar4(av[0], av[1], av[2], av[3]);
#endif
    }
}

// Ragged variant: slot k (state slot_state[k], or k itself) owns len[k] records in the sliced layout given by
// slice_row_off; record t of STATE sid uses counter (t, sid, stream, 0) and the action is uniform over the first
// n_live[sid] candidates (or all A): the same stream as the dense kernel above when every length is T.  A thread
// produces one quad (16-byte store); its slice is found by bisection over slice_row_off (wave-uniform).
__global__ __launch_bounds__(256) void sample_state_records_ragged_kernel(
    const float* __restrict__ Q, int q_rows, int S, int A, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, const int32_t* __restrict__ n_live,
    float sigma, uint32_t k0, uint32_t k1, uint32_t stream_id, uint32_t state_id_base, const int32_t* __restrict__ state_ids,
    float* __restrict__ R, uint8_t* __restrict__ act) {
    const int W = slices_of(S);
    const int64_t total = (slice_row_off[W] >> 2) * WAVE;
    int64_t g_lo, g_hi;
    block_range(total, g_lo, g_hi);
    for (int64_t g = g_lo + threadIdx.x; g < g_hi; g += blockDim.x) {
        const int lane = (int)(g & (WAVE - 1));
        const int64_t row = (g >> 6) << 2;
        int lo = 0, hi = W;                                  // largest w with slice_row_off[w] <= row
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (slice_row_off[mid] <= row) lo = mid; else hi = mid;
        }
        const int k = lo * WAVE + lane;
        const int64_t t0 = row - slice_row_off[lo];
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        int av[4] = {0, 0, 0, 0};
        if (k < S) {
            const int n = len[k];
            const int sid = slot_state ? slot_state[k] : k;
            const int nl = n_live ? n_live[sid] : A;
            const float* q = Q + (q_rows == 1 ? 0 : (int64_t)sid * A);
            // the state's Philox subsequence: its GLOBAL id, so that a rank holding any subset of a larger table's states
            // draws exactly the rows the whole table holds for them
            const uint32_t gid = state_ids ? (uint32_t)state_ids[sid] : (uint32_t)sid + state_id_base;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t t = t0 + j;
                if (t < n) {
                    const U4 x = philox4x32_10((uint32_t)t, gid, stream_id, 0u, k0, k1);
                    const int a = (int)__umulhi(x.x0, (uint32_t)nl);
                    const float z = bm_radius(x.x1) * __builtin_amdgcn_cosf(unit_open(x.x2));
                    rv[j] = fmaf(sigma, z, q[a]);
                    av[j] = a;
                }
            }
        }
#if DCARL_SAMPLER_NT
        nt_store16(reinterpret_cast<float4*>(R) + g, __float_as_uint(rv[0]), __float_as_uint(rv[1]), __float_as_uint(rv[2]), __float_as_uint(rv[3]));
        nt_store4(reinterpret_cast<uchar4*>(act) + g, (unsigned)av[0] | ((unsigned)av[1] << 8) | ((unsigned)av[2] << 16) | ((unsigned)av[3] << 24));
#else
        reinterpret_cast<float4*>(R)[g] = make_float4(rv[0], rv[1], rv[2], rv[3]);
        reinterpret_cast<uchar4*>(act)[g] = make_uchar4(av[0], av[1], av[2], av[3]);
#endif
    }
}

// {s,a,R} pairs.  Draw g (global index offset+i) belongs to group G = g/4 and is draw k = g%4 of it; the group owns
// the 12 words of Philox calls with counters 3G, 3G+1, 3G+2 (64-bit, split lo/hi) and draw k uses words 3k..3k+2 =
// (action word, Box-Muller word 1, word 2): all four outputs of every Philox block are consumed (3 blocks per 4
// draws instead of 4), one thread produces a whole group and stores it with 16-byte vectors.
// IDX24: S, A < 2^24 and S*A < 2^32, so the Q index is one full-rate v_mad_u32_u24 instead of a quarter-rate 64-bit multiply.
// The visit index is INDEX work and therefore exact: idx = floor((3 + 1*z)/6*S) evaluated in f64 on the f32 normal z the
// kernel drew, operation by operation like NumPy (DS:14-15); the quotient by 6 as two fma around RN(1/6), which equals the
// IEEE quotient for every f32 z (tools/div6_f64_check.c, exhaustive).  ZV: also hand out z (the normals behind DS:45's
// indices), so that a checker can redo the index arithmetic on the very same draws.
__device__ __forceinline__ double visit_floor_f64(float z, double S) {
    const double x = 3.0 + (double)z;
    double q = x * (1.0 / 6.0);
    asm volatile("" : "+v"(q));                              // (no contraction of the product into the fma below)
    q = fma(fma(-6.0, q, x), 1.0 / 6.0, q);
    double v = q * S;
    asm volatile("" : "+v"(v));
    return floor(v);
}
template <bool IDX24, bool ZV>
__global__ __launch_bounds__(256) void sample_pairs_kernel(
    const float* __restrict__ Q, int S, int A, int64_t N, float sigma, uint32_t k0, uint32_t k1, uint64_t offset,
    uint32_t stream_id, int32_t* __restrict__ idx, int32_t* __restrict__ act, float* __restrict__ R, float* __restrict__ z_visit) {
    const uint64_t G0 = offset >> 2;
    const uint64_t ngroups = ((offset + (uint64_t)N + 3) >> 2) - G0;
    const bool aligned = (offset & 3) == 0;
    // A CONTIGUOUS range of groups per block, not a grid stride (round 5).  With the grid stride all resident blocks write inside three
    // compact 8-MiB windows, one per output array, that advance in lockstep: whether those windows fall on the same DRAM banks is decided
    // by where the allocator put the three arrays, and twelve placements in one process gave 2.17 ... 3.40 ms for 2^30 pairs (what round 4
    // took for box-to-box variance; a single-stream fill of the same memory: 1.87 ms every time).  With a range per block the resident
    // blocks write all over the three arrays: 2.19 ... 2.50 ms over the same twelve placements (profiles/r05_sampler_variance.txt).
    int64_t j_lo, j_hi;
    block_range((int64_t)ngroups, j_lo, j_hi);
    for (uint64_t j = (uint64_t)j_lo + threadIdx.x; j < (uint64_t)j_hi; j += blockDim.x) {
        const uint64_t G = G0 + j;
        uint32_t w[12];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint64_t ctr = 3 * G + c;
            const U4 x = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), stream_id, 0u, k0, k1);
            w[4 * c] = x.x0; w[4 * c + 1] = x.x1; w[4 * c + 2] = x.x2; w[4 * c + 3] = x.x3;
        }
        int si[4], ai[4];
        float ri[4], zi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = (int)__umulhi(w[3 * k], (uint32_t)A);                 // DS:54 uniform action
            const float rad = bm_radius(w[3 * k + 1]), tu = unit_open(w[3 * k + 2]);
            const float zr = rad * __builtin_amdgcn_cosf(tu), zs = rad * __builtin_amdgcn_sinf(tu);
            const double v = visit_floor_f64(zs, (double)S);                    // DS:14-15, exact in f64
            const int s = (v < 0.0 || v >= (double)S) ? -1 : (int)v;            // DS:50-51
            si[k] = s;
            ai[k] = a;
            zi[k] = zs;
            const uint64_t qi = IDX24 ? (uint64_t)(__umul24((uint32_t)s, (uint32_t)A) + (uint32_t)a) : (uint64_t)s * A + a;
            ri[k] = s < 0 ? 0.f : fmaf(sigma, zr, Q[qi]);                       // DS:9
        }
        const int64_t i0 = (int64_t)(4 * G) - (int64_t)offset;                  // output index of draw 0 of the group
        if (aligned && i0 + 3 < N) {
#if DCARL_SAMPLER_NT
            nt_store16(reinterpret_cast<int4*>(idx) + (i0 >> 2), (unsigned)si[0], (unsigned)si[1], (unsigned)si[2], (unsigned)si[3]);
            nt_store16(reinterpret_cast<int4*>(act) + (i0 >> 2), (unsigned)ai[0], (unsigned)ai[1], (unsigned)ai[2], (unsigned)ai[3]);
            nt_store16(reinterpret_cast<float4*>(R) + (i0 >> 2), __float_as_uint(ri[0]), __float_as_uint(ri[1]), __float_as_uint(ri[2]), __float_as_uint(ri[3]));
            if (ZV) nt_store16(reinterpret_cast<float4*>(z_visit) + (i0 >> 2), __float_as_uint(zi[0]), __float_as_uint(zi[1]), __float_as_uint(zi[2]), __float_as_uint(zi[3]));
#else
            reinterpret_cast<int4*>(idx)[i0 >> 2] = make_int4(si[0], si[1], si[2], si[3]);
            reinterpret_cast<int4*>(act)[i0 >> 2] = make_int4(ai[0], ai[1], ai[2], ai[3]);
            reinterpret_cast<float4*>(R)[i0 >> 2] = make_float4(ri[0], ri[1], ri[2], ri[3]);
            if (ZV) reinterpret_cast<float4*>(z_visit)[i0 >> 2] = make_float4(zi[0], zi[1], zi[2], zi[3]);
#endif
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = i0 + k;
                if (i >= 0 && i < N) { idx[i] = si[k]; act[i] = ai[k]; R[i] = ri[k]; if (ZV) z_visit[i] = zi[k]; }
            }
        }
    }
}

// ---- injected-noise path, float64, bit-exact with the reference arithmetic ---------------------------
// NumPy rounds after every operation; an empty asm on the product keeps the backend from fusing a*b+c into an fma.
__device__ __forceinline__ double mul_rounded(double a, double b) {
    double p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
__global__ __launch_bounds__(256) void visit_index_kernel(const double* __restrict__ z, int64_t M, int S,
                                                          int32_t* __restrict__ idx) {
#pragma clang fp contract(off)   // the reference rounds after every operation
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    // norm.rvs(loc=3, scale=1) == 3 + 1*z ; floor(x/6*state_num).astype(int)   (DS:14-15)
    const double x = 3.0 + mul_rounded(1.0, z[i]);
    const double v = floor(mul_rounded(x / 6.0, (double)S));
    idx[i] = (v < 0.0 || v >= (double)S) ? -1 : (int32_t)v;                    // DS:50-51
}

// DS:14-15 as random_state_norm returns it: the raw floor values (they may lie outside [0, state_num); DS:50-51 filters later)
__global__ __launch_bounds__(256) void visit_floor_kernel(const double* __restrict__ z, int64_t M, int S,
                                                          int64_t* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double x = 3.0 + mul_rounded(1.0, z[i]);
    out[i] = (int64_t)floor(mul_rounded(x / 6.0, (double)S));                   // np.floor(...).astype(int)
}

// DS:19-28 random_state_manual on injected streams: u[i] = the i-th random.random(), r[j] = the j-th random.randint(1,
// state_num-1) (one per i with u[i] > 0.1, in order); out[i] = u[i] > 0.1 ? r[kept_rank[i]] : 0.
__global__ __launch_bounds__(256) void state_manual_kernel(const double* __restrict__ u, const int64_t* __restrict__ kept_rank,
                                                           const int32_t* __restrict__ r, int64_t M, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    out[i] = u[i] > 0.1 ? r[kept_rank[i]] : 0;
}

__global__ __launch_bounds__(256) void sample_from_noise_kernel(
    const int32_t* __restrict__ idx, const int64_t* __restrict__ kept_rank, int64_t M,
    const double* __restrict__ states, const double* __restrict__ Q64, int S, int A,
    const int32_t* __restrict__ acts, const double* __restrict__ z_reward, double sigma,
    double* __restrict__ out_rows) {
#pragma clang fp contract(off)   // Q + 50*z must round twice, like NumPy
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int s = idx[i];
    if (s < 0 || s >= S) return;
    const int64_t j = kept_rank[i];
    const int a = acts[j];
    // norm.rvs(loc=Q, scale=50) == Q + 50*z with two roundings (no fma)   (DS:9)
    const double r = Q64[(int64_t)s * A + a] + mul_rounded(sigma, z_reward[j]);
    double4 row = make_double4((double)s, states[s], (double)a, r);           // DS:55 record layout
    reinterpret_cast<double4*>(out_rows)[j] = row;
}

int launch_sample_state_records(const float* Q, int q_rows, int S, int A, int64_t T, double sigma, uint64_t seed,
                                uint32_t stream_id, float* R, uint8_t* act, hipStream_t st) {
    const int64_t total = (int64_t)(slices_of(S)) * ((T + 3) >> 2) * WAVE;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(sample_state_records_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Q, q_rows, S, A, T,
                       (float)sigma, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, R, act);
    return 0;
}

int launch_sample_state_records_ragged(const float* Q, int q_rows, int S, int A, const int64_t* slice_row_off,
                                       int64_t total_rows, const int32_t* len, const int32_t* slot_state,
                                       const int32_t* n_live, double sigma, uint64_t seed, uint32_t stream_id,
                                       uint32_t state_id_base, const int32_t* state_ids, float* R, uint8_t* act, hipStream_t st) {
    const int64_t total = (total_rows >> 2) * WAVE;
    if (S == 0 || total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(sample_state_records_ragged_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Q, q_rows, S, A,
                       slice_row_off, len, slot_state, n_live, (float)sigma, (uint32_t)seed, (uint32_t)(seed >> 32),
                       stream_id, state_id_base, state_ids, R, act);
    return 0;
}

int launch_sample_pairs(const float* Q, int S, int A, int64_t N, double sigma, uint64_t seed, uint64_t offset,
                        uint32_t stream_id, int32_t* idx, int32_t* act, float* R, float* z_visit, hipStream_t st) {
    if (N == 0) return 0;
    int64_t blocks = (N / 4 + 1 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const bool idx24 = S < (1 << 24) && A < (1 << 24) && (int64_t)S * A < ((int64_t)1 << 32);
    auto kern = z_visit ? (idx24 ? sample_pairs_kernel<true, true> : sample_pairs_kernel<false, true>)
                        : (idx24 ? sample_pairs_kernel<true, false> : sample_pairs_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st,
                       Q, S, A, N, (float)sigma, (uint32_t)seed, (uint32_t)(seed >> 32), offset, stream_id, idx, act, R, z_visit);
    return 0;
}

int launch_visit_index(const double* z, int64_t M, int S, int32_t* idx, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(visit_index_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, z, M, S, idx);
    return 0;
}

int launch_visit_floor(const double* z, int64_t M, int S, int64_t* out, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(visit_floor_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, z, M, S, out);
    return 0;
}
int launch_state_manual(const double* u, const int64_t* kept_rank, const int32_t* r, int64_t M, int32_t* out, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(state_manual_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, u, kept_rank, r, M, out);
    return 0;
}

int launch_sample_from_noise(const int32_t* idx, const int64_t* kept_rank, int64_t M, const double* states,
                             const double* Q64, int S, int A, const int32_t* acts, const double* z_reward,
                             double sigma, double* out_rows, hipStream_t st) {
    if (M == 0) return 0;
    hipLaunchKernelGGL(sample_from_noise_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, idx,
                       kept_rank, M, states, Q64, S, A, acts, z_reward, sigma, out_rows);
    return 0;
}

}  // namespace dcarl
