// Record ingest (reference (N,4) f64 table -> sliced device layout), Sim2's overall_value delta, f64 scan.
#include "common.h"

namespace dcarl {

// e(s,t) of include/dcarl.h
__device__ __forceinline__ int64_t elem_index(const int64_t* slice_row_off, int s, int64_t t) {
    return (slice_row_off[s >> 6] + (t & ~(int64_t)3)) * WAVE + (int64_t)(s & 63) * 4 + (t & 3);
}

// a11 / S1:73-78: one thread per grouped position p; row = data[order[p]] = {state, feature, action, reward}.
template <typename T>
__global__ __launch_bounds__(256) void pack_records_kernel(
    const double* __restrict__ data, const int64_t* __restrict__ order, const int64_t* __restrict__ state_off,
    const int64_t* __restrict__ slice_row_off, int64_t N, int S, T* __restrict__ R, uint8_t* __restrict__ act,
    int64_t* __restrict__ rec_elem) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int64_t k = order[p];
    const double4 row = reinterpret_cast<const double4*>(data)[k];
    const int s = (int)row.x;                                   // S1:77 idx = int(idx_ori)
    if (s < 0 || s >= S) return;                                // ids are validated on the host; never scatter out of range
    const int64_t t = p - state_off[s];
    const int64_t e = elem_index(slice_row_off, s, t);
    R[e] = (T)row.w;
    act[e] = (uint8_t)(int)row.z;                               // S1:78 act = int(act_ori)
    if (rec_elem) rec_elem[k] = e;
}

// S2:99-105 as a delta stream: state i contributes (max V[i] + 0.9) once activated (activation_value == -1
// forever, S2:59, so "- activation_value*0.9" is "+0.9").  delta[k] = c_i(t) - c_i(t-1).
template <typename T>
__global__ __launch_bounds__(256) void overall_delta_kernel(
    const T* __restrict__ step_val, const int32_t* __restrict__ act_step, const int32_t* __restrict__ rec_state,
    const int64_t* __restrict__ rec_elem, const int32_t* __restrict__ rec_t, int64_t N,
    double* __restrict__ delta) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int i = rec_state[k];
    const int t = rec_t[k];
    const int64_t e = rec_elem[k];
    const int latch = act_step[i];
    double cur = 0.0, prev = 0.0;
    if (latch != -1 && t + 1 >= latch) cur = (double)step_val[e] + 0.9;
    if (latch != -1 && t >= latch && t > 0) {
        const int64_t ep = (t & 3) ? e - 1 : e - (4 * WAVE - 3);   // element of record t-1 of the same state
        prev = (double)step_val[ep] + 0.9;
    }
    delta[k] = cur - prev;
}

// ---- inclusive f64 prefix sum: per-tile scan -> scan of tile totals (one block) -> add carry ---------------
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ double block_exclusive_scan(double v, double* total) {
    __shared__ double wsum[SCAN_THREADS / WAVE];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { double o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == WAVE - 1) wsum[wid] = inc;
    __syncthreads();
    double carry = 0.0, tot = 0.0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / WAVE; ++i) { if (i < wid) carry += wsum[i]; tot += wsum[i]; }
    __syncthreads();
    *total = tot;
    return carry + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const double* __restrict__ in,
                                                                  double* __restrict__ out, int64_t N,
                                                                  double* __restrict__ tile_sum) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    double x[SCAN_ITEMS], run = 0.0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { x[i] = (base + i < N) ? in[base + i] : 0.0; run += x[i]; x[i] = run; }
    double tot;
    const double ex = block_exclusive_scan(run, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < N) out[base + i] = x[i] + ex;
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_totals_kernel(double* __restrict__ tile_sum, int64_t ntiles) {
    double carry = 0.0;                                          // exclusive scan of tile totals, chunked
    for (int64_t c = 0; c < ntiles; c += SCAN_THREADS) {
        const int64_t i = c + threadIdx.x;
        const double v = i < ntiles ? tile_sum[i] : 0.0;
        double tot;
        const double ex = block_exclusive_scan(v, &tot);
        if (i < ntiles) tile_sum[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(double* __restrict__ out, int64_t N,
                                                                const double* __restrict__ tile_sum) {
    const double c = tile_sum[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < N) out[base + i] += c;
}

template <typename T>
int launch_pack_records(const double* data, const int64_t* order, const int64_t* state_off,
                        const int64_t* slice_row_off, int64_t N, int S, T* R, uint8_t* act, int64_t* rec_elem,
                        hipStream_t st) {
    if (N == 0) return 0;
    hipLaunchKernelGGL((pack_records_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, data, order,
                       state_off, slice_row_off, N, S, R, act, rec_elem);
    return 0;
}
template int launch_pack_records<float>(const double*, const int64_t*, const int64_t*, const int64_t*, int64_t,
                                        int, float*, uint8_t*, int64_t*, hipStream_t);
template int launch_pack_records<double>(const double*, const int64_t*, const int64_t*, const int64_t*, int64_t,
                                         int, double*, uint8_t*, int64_t*, hipStream_t);

template <typename T>
int launch_overall_delta(const T* step_val, const int32_t* act_step, const int32_t* rec_state,
                         const int64_t* rec_elem, const int32_t* rec_t, int64_t N, double* delta, hipStream_t st) {
    if (N == 0) return 0;
    hipLaunchKernelGGL((overall_delta_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, step_val,
                       act_step, rec_state, rec_elem, rec_t, N, delta);
    return 0;
}
template int launch_overall_delta<float>(const float*, const int32_t*, const int32_t*, const int64_t*,
                                         const int32_t*, int64_t, double*, hipStream_t);
template int launch_overall_delta<double>(const double*, const int32_t*, const int32_t*, const int64_t*,
                                          const int32_t*, int64_t, double*, hipStream_t);

// continuous observations -> integer cell coordinates (the step that turns CARLA records into the
// integer state ids the confidence path assumes).  cells[i][k] = floor(obs[i][k] / width[k]); one thread per element.
__global__ __launch_bounds__(256) void state_cells_kernel(const double* __restrict__ obs, int64_t N, int D,
                                                          const double* __restrict__ width, int32_t* __restrict__ cells) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * D) return;
    cells[e] = (int32_t)floor(obs[e] / width[e % D]);
}

int launch_state_cells(const double* obs, int64_t N, int D, const double* width, int32_t* cells, hipStream_t st) {
    if (N * D == 0) return 0;
    hipLaunchKernelGGL(state_cells_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, st, obs, N, D, width, cells);
    return 0;
}

int64_t scan_workspace_bytes(int64_t N) { return ((N + SCAN_TILE - 1) / SCAN_TILE + 1) * (int64_t)sizeof(double); }

int launch_scan(const double* in, double* out, int64_t N, void* ws, hipStream_t st) {
    if (N == 0) return 0;
    const int64_t ntiles = (N + SCAN_TILE - 1) / SCAN_TILE;
    double* tile_sum = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, in, out, N, tile_sum);
    if (ntiles > 1) {
        hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, tile_sum, ntiles);
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, out, N, tile_sum);
    }
    return 0;
}

}  // namespace dcarl
