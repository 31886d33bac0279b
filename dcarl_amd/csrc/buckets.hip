// The reference's buckets themselves: `data_state_act[idx][act].append(R)` (S1:80).  The online kernels never
// materialise them (they keep sufficient statistics); the final-state kernels (bounds.hip) stream them.  This file
// turns an online record table (sliced time-major layout, include/dcarl.h) into the (state, action) bucket layout:
//   count_records_kernel  : n[s][a] = len(data_state_act[s][a]) after the whole table
//   group_records_kernel  : values[seg_off[s*A+a] + k] = k-th reward appended to bucket (s,a) (arrival order kept)
// and draws samples straight into the bucket layout (sample_buckets_kernel: add_an_act_data, DS:5-9, n times per
// bucket).  Mapping: lane = state, wavefront = slice, the lane walks its state's quads exactly like the online
// kernels; per-lane per-action cursors live in LDS [action][lane] (the bank depends on the lane only: conflict-free).
#include "common.h"
#include "philox.h"

namespace dcarl {

constexpr int GROUP_WAVES = 4;      // 4 slices per block: 4 x 32 x 64 x 4 B = 32 KiB of cursors

template <typename T, bool SCATTER>
__global__ __launch_bounds__(GROUP_WAVES* WAVE) void group_records_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, int S, int A, const int64_t* __restrict__ seg_off,
    T* __restrict__ values, int32_t* __restrict__ n_out) {
    __shared__ uint32_t cur[GROUP_WAVES][DCARL_MAX_ACTIONS][WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const int w = blockIdx.x * GROUP_WAVES + wv;                 // slice
    const int s = w * WAVE + lane;
    if (w * WAVE >= S) return;                                   // wave-uniform
    const int n = s < S ? len[s] : 0;
    const int so = (s < S && slot_state) ? slot_state[s] : s;    // buckets are numbered by STATE, whatever the slot order
    int64_t base = 0;                                            // the state's first sample in `values`
    if (SCATTER && s < S) base = seg_off[(int64_t)so * A];
    for (int a = 0; a < A; ++a)
        cur[wv][a][lane] = (SCATTER && s < S) ? (uint32_t)(seg_off[(int64_t)so * A + a] - base) : 0u;
    const int64_t row0 = slice_row_off[w];
    const int nmax = (int)(slice_row_off[w + 1] - row0);         // rows of the slice (multiple of 4)
    for (int t = 0; t < nmax; t += 4) {
        if (t >= n) continue;                                    // (no early exit: lanes of a slice differ in length)
        const int64_t e = (row0 + t) * WAVE + lane * 4;
        const uchar4 a4 = *reinterpret_cast<const uchar4*>(act + e);
        const uint8_t av[4] = {a4.x, a4.y, a4.z, a4.w};
        T rv[4];
        if (SCATTER) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = R[e + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (t + j < n) {
                const int a = av[j] < A ? av[j] : A - 1;          // ids are validated on the host; never index out of range
                const uint32_t k = cur[wv][a][lane]++;
                if (SCATTER) values[base + k] = rv[j];
            }
        }
    }
    if (!SCATTER && s < S)
        for (int a = 0; a < A; ++a) n_out[(int64_t)so * A + a] = (int32_t)cur[wv][a][lane];
}

template <typename T>
int launch_group_records(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int S, int A,
                         const int64_t* seg_off, T* values, int32_t* n_out, hipStream_t st) {
    if (S == 0) return 0;
    const int W = slices_of(S);
    dim3 grid((W + GROUP_WAVES - 1) / GROUP_WAVES), block(GROUP_WAVES * WAVE);
    if (values)
        hipLaunchKernelGGL((group_records_kernel<T, true>), grid, block, 0, st, R, act, slice_row_off, len, slot_state, S, A, seg_off,
                           values, n_out);
    else
        hipLaunchKernelGGL((group_records_kernel<T, false>), grid, block, 0, st, R, act, slice_row_off, len, slot_state, S, A, seg_off,
                           values, n_out);
    return 0;
}
template int launch_group_records<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int,
                                         const int64_t*, float*, int32_t*, hipStream_t);
template int launch_group_records<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, const int32_t*, int, int,
                                          const int64_t*, double*, int32_t*, hipStream_t);

// ---- samples drawn straight into the bucket layout --------------------------------------------------------------
// Sample i of the flat value array is normal k = i%4 of the Philox block with counter (lo(i/4), hi(i/4), stream, 1):
// k = 0,1 = Box-Muller (cos, sin) of words (x0, x1), k = 2,3 of words (x2, x3); value = Q[bucket(i)] + sigma*z (DS:9).
// A block owns whole states (grid-stride): the A+1 offsets of the state go to LDS, a thread produces one aligned
// group of four samples, finds the bucket of its first sample by scanning the offsets and walks on from there.
__global__ __launch_bounds__(256) void sample_buckets_kernel(
    const float* __restrict__ Q, int q_rows, int S, int A, const int64_t* __restrict__ seg_off, int64_t n_dense,
    float sigma, uint32_t k0, uint32_t k1, uint32_t stream_id, float* __restrict__ values) {
    __shared__ int64_t off[DCARL_MAX_ACTIONS + 1];
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        __syncthreads();
        if (threadIdx.x <= A)
            off[threadIdx.x] = seg_off ? seg_off[(int64_t)s * A + threadIdx.x] : ((int64_t)s * A + threadIdx.x) * n_dense;
        __syncthreads();
        const int64_t b0 = off[0], e0 = off[A];
        const float* q = Q + (q_rows == 1 ? 0 : (int64_t)s * A);
        for (int64_t v = (b0 >> 2) + threadIdx.x; v < ((e0 + 3) >> 2); v += blockDim.x) {
            const U4 x = philox4x32_10((uint32_t)v, (uint32_t)((uint64_t)v >> 32), stream_id, 1u, k0, k1);
            const float r0 = bm_radius(x.x0), r1 = bm_radius(x.x2);
            const float t0 = unit_open(x.x1), t1 = unit_open(x.x3);
            const float z[4] = {r0 * __builtin_amdgcn_cosf(t0), r0 * __builtin_amdgcn_sinf(t0),
                                r1 * __builtin_amdgcn_cosf(t1), r1 * __builtin_amdgcn_sinf(t1)};
            const int64_t i0 = v << 2;
            int a = 0;
            const int64_t first = i0 < b0 ? b0 : i0;
            while (a < A - 1 && off[a + 1] <= first) ++a;
            float out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = i0 + j;
                while (a < A - 1 && off[a + 1] <= i) ++a;
                out[j] = fmaf(sigma, z[j], q[a]);
            }
            if (i0 >= b0 && i0 + 4 <= e0) reinterpret_cast<float4*>(values)[v] = make_float4(out[0], out[1], out[2], out[3]);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j >= b0 && i0 + j < e0) values[i0 + j] = out[j];
            }
        }
    }
}

int launch_sample_buckets(const float* Q, int q_rows, int S, int A, const int64_t* seg_off, int64_t n_dense, double sigma,
                          uint64_t seed, uint32_t stream_id, float* values, hipStream_t st) {
    if (S == 0) return 0;
    const int blocks = S < 256 * 16 ? S : 256 * 16;
    hipLaunchKernelGGL(sample_buckets_kernel, dim3(blocks), dim3(256), 0, st, Q, q_rows, S, A, seg_off, n_dense,
                       (float)sigma, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, values);
    return 0;
}

}  // namespace dcarl
