// Sim2's overall_value delta, f64 scan, state ids, per-rank summary statistics.  (Record ingest: ingest.hip.)
#include "common.h"

namespace dcarl {

// e(s,t) of include/dcarl.h
__device__ __forceinline__ int64_t elem_index(const int64_t* slice_row_off, int s, int64_t t) {
    return (slice_row_off[s >> 6] + (t & ~(int64_t)3)) * WAVE + (int64_t)(s & 63) * 4 + (t & 3);
}

// S2:99-105 as a delta stream: state i contributes (max V[i] + 0.9) once activated (activation_value == -1
// forever, S2:59, so "- activation_value*0.9" is "+0.9").  delta[k] = c_i(t) - c_i(t-1).
template <typename T>
__global__ __launch_bounds__(256) void overall_delta_kernel(
    const T* __restrict__ step_val, const int32_t* __restrict__ act_step, const int32_t* __restrict__ rec_state,
    const int64_t* __restrict__ rec_elem, const int32_t* __restrict__ rec_t, int64_t N,
    double* __restrict__ delta, const int32_t* __restrict__ t_base, const double* __restrict__ prev_val) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const int i = rec_state[k];
    const int t = rec_t[k];
    const int64_t e = rec_elem[k];
    const int latch = act_step[i];
    // a CONTINUED loop (dcarl_trace_resume_*): t counts inside this chunk, the latch over all chunks; the state's records before
    // the chunk are t_base[i], the value its last one left is prev_val[i]
    const int tb = t_base ? t_base[i] : 0;
    double cur = 0.0, prev = 0.0;
    if (latch != -1 && tb + t + 1 >= latch) cur = (double)step_val[e] + 0.9;
    if (latch != -1 && tb + t >= latch && tb + t > 0) {
        if (t > 0) {
            const int64_t ep = (t & 3) ? e - 1 : e - (4 * WAVE - 3);   // element of record t-1 of the same state
            prev = (double)step_val[ep] + 0.9;
        } else {
            prev = prev_val[i] + 0.9;                              // the state's previous record lies in an earlier chunk
        }
    }
    delta[k] = cur - prev;
}

// ---- NaN / Inf census of a value buffer ---------------------------------------------------------------------------------
// The estimator's arg-max is defined for finite rewards (np.argmax would pick the first NaN; the kernels are built
// -fno-honor-nans and order NaN keys arbitrarily), so every builder of device tables checks its inputs: the ingest kernels
// on the fly (ingest.hip), everything else through this one pass.  Integer tests on the raw words (a floating-point test may
// be folded away under -fno-honor-nans).  count[0] += number of NaN / Inf elements.
template <int VB>
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const void* __restrict__ v_, int64_t n, unsigned long long* __restrict__ count) {
    unsigned bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (VB == 4) {
        const uint32_t* v = static_cast<const uint32_t*>(v_);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) bad += (v[i] & 0x7f800000u) == 0x7f800000u;
    } else {
        const uint2* v = static_cast<const uint2*>(v_);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) bad += (v[i].y & 0x7ff00000u) == 0x7ff00000u;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) bad += __shfl_xor(bad, off);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, (unsigned long long)bad);
}
int launch_count_nonfinite(const void* v, int value_bytes, int64_t n, int64_t* count, hipStream_t st) {
    (void)hipMemsetAsync(count, 0, sizeof(int64_t), st);
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (value_bytes == 4) hipLaunchKernelGGL((count_nonfinite_kernel<4>), dim3(blocks), dim3(256), 0, st, v, n, reinterpret_cast<unsigned long long*>(count));
    else hipLaunchKernelGGL((count_nonfinite_kernel<8>), dim3(blocks), dim3(256), 0, st, v, n, reinterpret_cast<unsigned long long*>(count));
    return 0;
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
int launch_overall_delta(const T* step_val, const int32_t* act_step, const int32_t* rec_state,
                         const int64_t* rec_elem, const int32_t* rec_t, int64_t N, double* delta, const int32_t* t_base,
                         const double* prev_val, hipStream_t st) {
    if (N == 0) return 0;
    hipLaunchKernelGGL((overall_delta_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, step_val,
                       act_step, rec_state, rec_elem, rec_t, N, delta, t_base, prev_val);
    return 0;
}
template int launch_overall_delta<float>(const float*, const int32_t*, const int32_t*, const int64_t*,
                                         const int32_t*, int64_t, double*, const int32_t*, const double*, hipStream_t);
template int launch_overall_delta<double>(const double*, const int32_t*, const int32_t*, const int64_t*,
                                          const int32_t*, int64_t, double*, const int32_t*, const double*, hipStream_t);

// continuous observations -> integer cell coordinates (the step that turns CARLA records into the
// integer state ids the confidence path assumes).  cells[i][k] = floor(obs[i][k] / width[k]); one thread per element.
__global__ __launch_bounds__(256) void state_cells_kernel(const double* __restrict__ obs, int64_t N, int D,
                                                          const double* __restrict__ width, int32_t* __restrict__ cells) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * D) return;
    cells[e] = (int32_t)floor(obs[e] / width[e % D]);
}

__device__ __forceinline__ void hash_step(unsigned long long& h, int32_t c) {   // one multiply-xorshift round per coordinate (splitmix64 style)
    h ^= (unsigned long long)(unsigned)c + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
}
// The same with the row's 64-bit hash on the way out (what dcarl_state_ids would otherwise re-read every row for).  A wavefront
// owns 64 consecutive rows = 64*D consecutive elements: it reads and writes them element-wise (lane = element: fully
// coalesced, like the kernel above), leaves the cells in an LDS tile, and then lane = ROW reads its D cells back as 16-byte
// vectors (row stride D words, D = 4 (2m+1) or padded: the 16 lanes of a read group hit 16 different 4-bank groups for D = 20)
// and hashes them from registers.  D a multiple of 4, D <= 64.
constexpr int CH_WAVES = 4;
struct StateSlot { unsigned long long key; int32_t rep; int32_t pad; };      // the id table's slots (below)
static_assert(sizeof(StateSlot) == 16, "one 16-byte load per probe");
__device__ __forceinline__ void insert_row(unsigned long long h, int64_t i, StateSlot* tab, int64_t cap, int32_t* __restrict__ slot_of, int64_t* out);
// INSERT: the row also enters the id table right here (dcarl_index_states: the probes' round trips to the memory side pass under
// this kernel's streaming instead of making a latency-bound kernel of their own, and the hashes never travel through HBM).
template <bool INSERT>
__global__ __launch_bounds__(CH_WAVES * WAVE) void state_cells_hash_kernel(const double* __restrict__ obs, int64_t N, int D,
                                                                           const double* __restrict__ width, int32_t* __restrict__ cells,
                                                                           unsigned long long* __restrict__ hash, StateSlot* tab, int64_t cap,
                                                                           int32_t* __restrict__ slot_of, int64_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int stride = D + ((D / 4) % 2 == 0 ? 4 : 0);           // words per tile row: an odd number of 16-byte vectors
    int32_t* tile = reinterpret_cast<int32_t*>(smem) + wv * WAVE * stride;
    const int64_t row0 = ((int64_t)blockIdx.x * CH_WAVES + wv) * WAVE;
    if (row0 >= N) return;
    const int64_t e0 = row0 * D;
    const int64_t ne = ((N - row0 < WAVE ? N - row0 : WAVE)) * D;  // elements of this wavefront's rows
    for (int j = 0; j < D; ++j) {
        const int64_t e = (int64_t)j * WAVE + lane;                // element inside the wavefront's stretch
        if (e < ne) {
            const int r = (int)(e / D), k = (int)(e - (int64_t)r * D);
            const int32_t c = (int32_t)floor(obs[e0 + e] / width[k]);
            cells[e0 + e] = c;
            tile[r * stride + k] = c;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (row0 + lane < N) {
        unsigned long long h = 0x9E3779B97F4A7C15ull;
        const int4* r4 = reinterpret_cast<const int4*>(tile + lane * stride);
        for (int k = 0; k < D / 4; ++k) {
            const int4 c = r4[k];
            hash_step(h, c.x); hash_step(h, c.y); hash_step(h, c.z); hash_step(h, c.w);
        }
        h |= 1ull;                                               // 0 marks an empty slot of the table
        if constexpr (INSERT) insert_row(h, row0 + lane, tab, cap, slot_of, out);
        else hash[row0 + lane] = h;
    }
}

// The row tiles of these kernels are dynamic LDS sized by D: at the ABI's limit D = 64 they need 69 632 B, above the 64 KiB a
// kernel may use without asking -- raise the limit once per kernel (a CU has 160 KiB).
template <class K>
static void allow_large_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}
static void allow_large_lds_once() {
    static const bool done = [] {
        allow_large_lds(state_cells_hash_kernel<false>);
        allow_large_lds(state_cells_hash_kernel<true>);
        return true;
    }();
    (void)done;
}

int launch_state_cells(const double* obs, int64_t N, int D, const double* width, int32_t* cells, unsigned long long* hash, hipStream_t st) {
    if (N * D == 0) return 0;
    allow_large_lds_once();
    if (hash) hipLaunchKernelGGL(state_cells_hash_kernel<false>, dim3((unsigned)((N + CH_WAVES * WAVE - 1) / (CH_WAVES * WAVE))), dim3(CH_WAVES * WAVE),
                                 (unsigned)(CH_WAVES * WAVE * (D + 4) * 4), st, obs, N, D, width, cells, hash, nullptr, 0, nullptr, nullptr);
    else hipLaunchKernelGGL(state_cells_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, st, obs, N, D, width, cells);
    return 0;
}

// ---- cells -> dense state ids, sort-free --------------------------------------------------------
// Rows with equal cell coordinates share an id; ids are dense and numbered in order of FIRST APPEARANCE (arrival order,
// like everything else on the path).  Three passes over the rows, no sort, no N-sized prefix sum:
//   insert : 64-bit hash of the row -> open-addressing table of 16-byte slots {hash, representative} (capacity >= 2 x the
//            distinct-state estimate, linear probing, one 64-bit CAS per probe); the slot's representative is the smallest row
//            index that hashed there (atomicMin: deterministic);
//   verify : every row compares its D coordinates with its representative's (a 64-bit hash collision between different
//            cells is counted, not ignored: the caller must not use the ids then); "this row is its state's first" leaves as
//            ONE BIT per row (a ballot per wavefront: 8 bytes per 64 rows where round 2 wrote an f64 flag per row and ran an
//            N-element f64 prefix sum over them);
//   words  : exclusive prefix of the popcounts of the N/64 bit words (three small kernels over 4 B per 64 rows);
//   assign : id[i] = prefix[word of rep[i]] + bits of that word below rep[i].
struct StateIdWs {
    StateSlot* tab; int32_t* slot; unsigned long long* bits; uint32_t* wpre; uint32_t* tsum; int64_t cap, words, tiles;
};
constexpr int BW_THREADS = 256, BW_ITEMS = 8, BW_TILE = BW_THREADS * BW_ITEMS;      // bit words per block of the prefix kernels
// capacity: a power of two >= 2 x the number of distinct states the caller expects (0 / more than N: N).  A table sized for the
// states instead of the records stays in the L2 (2^17 states: 4 MiB instead of 512 MiB for 2^24 records) — what the atomics of
// the insert pass cost depends on little else.
__host__ __device__ inline int64_t state_ids_capacity(int64_t N, int64_t max_states) {
    const int64_t want = (max_states > 0 && max_states < N) ? max_states : N;
    int64_t cap = 64;
    while (cap < 2 * want) cap <<= 1;
    return cap;
}
static StateIdWs state_ids_layout(void* ws, int64_t N, int64_t max_states) {
    StateIdWs w;
    w.cap = state_ids_capacity(N, max_states);
    w.words = (N + 63) / 64;
    w.tiles = (w.words + BW_TILE - 1) / BW_TILE;
    unsigned char* p = reinterpret_cast<unsigned char*>(ws);
    w.tab = reinterpret_cast<StateSlot*>(p); p += w.cap * 16;
    w.bits = reinterpret_cast<unsigned long long*>(p); p += w.words * 8;
    w.slot = reinterpret_cast<int32_t*>(p); p += N * 4;
    w.wpre = reinterpret_cast<uint32_t*>(p); p += w.words * 4;
    w.tsum = reinterpret_cast<uint32_t*>(p);
    return w;
}
int64_t state_ids_workspace_bytes(int64_t N, int64_t max_states) {
    const int64_t cap = state_ids_capacity(N, max_states), words = (N + 63) / 64;
    return cap * 16 + words * 12 + N * 4 + ((words + BW_TILE - 1) / BW_TILE + 1) * 4 + 64;
}

__global__ __launch_bounds__(256) void state_ids_clear_kernel(StateSlot* tab, int64_t cap, int64_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) *reinterpret_cast<uint4*>(&tab[i]) = make_uint4(0u, 0u, 0x7fffffffu, 0u);
    if (i == 0) { out[0] = 0; out[1] = 0; out[2] = 0; }
}
// Rows are read as 16-byte vectors when D is a multiple of 4 (the CARLA observation has 20 coordinates): a thread owns a
// row, so its loads are strided by the row size whatever their width -- 4x fewer of them is what matters (VEC4).
template <bool VEC4>
__device__ __forceinline__ unsigned long long hash_cells(const int32_t* __restrict__ row, int D) {
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    if (VEC4) {
        const int4* r4 = reinterpret_cast<const int4*>(row);
        for (int k = 0; k < D / 4; ++k) {
            const int4 c = r4[k];
            hash_step(h, c.x); hash_step(h, c.y); hash_step(h, c.z); hash_step(h, c.w);
        }
    } else {
        for (int k = 0; k < D; ++k) hash_step(h, row[k]);
    }
    return h | 1ull;                                            // 0 marks an empty slot
}
// one device-coherent 16-byte load of a slot (the table is shared by the XCDs, whose L2s do not snoop each other: an agent-scope
// atomic load per FIELD is two round trips to the memory side; torn reads are harmless — the key is written once, the
// representative only ever decreases, and both are re-checked by the atomic that follows a miss)
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_slot(const StateSlot* p) {
    v4u_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void insert_row(unsigned long long h, int64_t i, StateSlot* tab, int64_t cap, int32_t* __restrict__ slot_of, int64_t* out) {
    int64_t s = (int64_t)(h >> 1) & (cap - 1);
    int64_t probes = 0;
    int32_t seen_rep;
    for (;;) {
        // look before the CAS: in tables that revisit states (CARLA: ~100 records per state) the key is there already for all but
        // the first row of a state, and an atomic on a word 100 rows fight over costs what a load of it does not.  (A stale
        // read only costs the atomic it tried to save: keys never change once set.)
        const uint4 v = load_slot(&tab[s]);
        unsigned long long old = ((unsigned long long)v.y << 32) | v.x;
        seen_rep = (int32_t)v.z;
        if (old != h) { old = atomicCAS(&tab[s].key, 0ull, h); seen_rep = 0x7fffffff; }
        if (old == 0ull || old == h) break;
        s = (s + 1) & (cap - 1);
        if (++probes >= cap) {                                  // the table is full: more distinct states than the caller sized it for
            out[2] = 1;
            slot_of[i] = 0;
            return;
        }
    }
    // the representative only ever decreases: a row that has seen a smaller index than its own has nothing to add
    if (seen_rep > (int32_t)i) atomicMin(&tab[s].rep, (int32_t)i);
    slot_of[i] = (int32_t)s;
}
template <bool VEC4>
__global__ __launch_bounds__(256) void state_ids_insert_kernel(const int32_t* __restrict__ cells, const unsigned long long* __restrict__ hash,
                                                               int64_t N, int D, StateSlot* tab, int64_t cap,
                                                               int32_t* __restrict__ slot_of, int64_t* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    insert_row(hash ? hash[i] : hash_cells<VEC4>(cells + i * D, D), i, tab, cap, slot_of, out);
}
template <bool VEC4>
__global__ __launch_bounds__(256) void state_ids_verify_kernel(const int32_t* __restrict__ cells, int64_t N, int D,
                                                               const StateSlot* __restrict__ tab, int32_t* __restrict__ slot_of,
                                                               unsigned long long* __restrict__ bits, int64_t* out) {
    // VEC4: the wavefront's 64 rows are one contiguous stretch: it is read with coalesced 16-byte loads (lane = vector), parked
    // in LDS, and lane = ROW takes its vectors back from there (row stride: an odd number of vectors, conflict-free) — a thread
    // reading its own row straight from memory strides by the row size (1.07 -> 0.7 ms for 2^24 rows of 20 cells).  The
    // representative's row is a gather either way (and an L2 hit: tables revisit states).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row0 = i - lane;
    if (row0 >= N) return;                                      // wave-uniform
    const int nv = D / 4, stride = nv + (nv % 2 == 0 ? 1 : 0);    // vectors per tile row
    int4* tile = reinterpret_cast<int4*>(smem) + wv * WAVE * stride;
    if (VEC4) {
        const int64_t nvec = (N - row0 < WAVE ? N - row0 : WAVE) * nv;
        const int4* src = reinterpret_cast<const int4*>(cells + row0 * D);
        for (int j = 0; j < nv; ++j) {
            const int64_t v = (int64_t)j * WAVE + lane;
            if (v < nvec) { const int r = (int)(v / nv); tile[r * stride + (int)(v - (int64_t)r * nv)] = src[v]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    bool first = false;
    if (i < N) {
        const int32_t r = tab[slot_of[i]].rep;
        bool same = true;
        if (r != (int32_t)i) {                                  // (a representative is its own row)
            if (VEC4) {
                const int4* a = tile + lane * stride;
                const int4* b = reinterpret_cast<const int4*>(cells + (int64_t)r * D);
                for (int k = 0; k < nv; ++k) {
                    const int4 x = a[k], y = b[k];
                    same &= (x.x == y.x) & (x.y == y.y) & (x.z == y.z) & (x.w == y.w);
                }
            } else {
                for (int k = 0; k < D; ++k) same &= cells[i * D + k] == cells[(int64_t)r * D + k];
            }
        }
        if (!same) atomicAdd(reinterpret_cast<unsigned long long*>(&out[1]), 1ull);   // different cells, equal 64-bit hash
        first = r == (int32_t)i;
        slot_of[i] = r;                                         // from here on: the representative row
    }
    const unsigned long long m = __ballot(first);
    if (lane == 0) bits[row0 >> 6] = m;                         // (blocks are multiples of 64 rows: word = wavefront)
}
// exclusive prefix of popcount(bits[w]) over the W words: tile sums, one block over the tile sums, per-word prefixes
__global__ __launch_bounds__(BW_THREADS) void bits_tile_sum_kernel(const unsigned long long* __restrict__ bits, int64_t W,
                                                                   uint32_t* __restrict__ tsum) {
    __shared__ uint32_t ws[BW_THREADS / WAVE];
    const int64_t base = (int64_t)blockIdx.x * BW_TILE + (int64_t)threadIdx.x * BW_ITEMS;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < BW_ITEMS; ++k) if (base + k < W) run += (uint32_t)__popcll(bits[base + k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run += __shfl_xor(run, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = run;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < BW_THREADS / WAVE; ++w) t += ws[w]; tsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void bits_tile_scan_kernel(uint32_t* __restrict__ tsum, int64_t ntiles, int64_t* __restrict__ out) {
    __shared__ uint32_t ws[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int64_t c = 0; c < ntiles; c += 1024) {
        const int64_t i = c + threadIdx.x;
        const uint32_t v = i < ntiles ? tsum[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        __syncthreads();
        if (lane == WAVE - 1) ws[wid] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (int j = 0; j < 16; ++j) { if (j < wid) pre += ws[j]; tot += ws[j]; }
        if (i < ntiles) tsum[i] = carry + pre + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) out[0] = (int64_t)carry;              // the number of distinct states
}
__global__ __launch_bounds__(BW_THREADS) void bits_word_prefix_kernel(const unsigned long long* __restrict__ bits, int64_t W,
                                                                      const uint32_t* __restrict__ tsum, uint32_t* __restrict__ wpre) {
    __shared__ uint32_t ws[BW_THREADS / WAVE];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * BW_TILE + (int64_t)threadIdx.x * BW_ITEMS;
    uint32_t x[BW_ITEMS], run = 0;
#pragma unroll
    for (int k = 0; k < BW_ITEMS; ++k) { x[k] = base + k < W ? (uint32_t)__popcll(bits[base + k]) : 0u; run += x[k]; }
    uint32_t inc = run;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const uint32_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == WAVE - 1) ws[wid] = inc;
    __syncthreads();
    uint32_t ex = tsum[blockIdx.x] + inc - run;
    for (int j = 0; j < wid; ++j) ex += ws[j];
#pragma unroll
    for (int k = 0; k < BW_ITEMS; ++k) { if (base + k < W) wpre[base + k] = ex; ex += x[k]; }
}
__global__ __launch_bounds__(256) void state_ids_assign_kernel(const int32_t* __restrict__ rep_of, const unsigned long long* __restrict__ bits,
                                                               const uint32_t* __restrict__ wpre, int64_t N, int32_t* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t r = rep_of[i];
    const unsigned long long below = bits[r >> 6] & ((1ull << (r & 63)) - 1ull);
    ids[i] = (int32_t)(wpre[r >> 6] + (uint32_t)__popcll(below));
}

// verify -> bit words -> prefix -> ids, once every row is in the table
static void finish_state_ids(const int32_t* cells, int64_t N, int D, const StateIdWs& w, int32_t* ids, int64_t* out, hipStream_t st) {
    const unsigned nb = (unsigned)((N + 255) / 256);
    const bool vec4 = D % 4 == 0 && (reinterpret_cast<uintptr_t>(cells) & 15u) == 0;
    static const bool lds_ok = [] { allow_large_lds(state_ids_verify_kernel<true>); return true; }();
    (void)lds_ok;
    hipLaunchKernelGGL(vec4 ? state_ids_verify_kernel<true> : state_ids_verify_kernel<false>, dim3(nb), dim3(256),
                       vec4 ? (unsigned)(256 * (D / 4 + 1) * 16) : 0u, st, cells, N, D, w.tab, w.slot, w.bits, out);
    hipLaunchKernelGGL(bits_tile_sum_kernel, dim3((unsigned)w.tiles), dim3(BW_THREADS), 0, st, w.bits, w.words, w.tsum);
    hipLaunchKernelGGL(bits_tile_scan_kernel, dim3(1), dim3(1024), 0, st, w.tsum, w.tiles, out);
    hipLaunchKernelGGL(bits_word_prefix_kernel, dim3((unsigned)w.tiles), dim3(BW_THREADS), 0, st, w.bits, w.words, w.tsum, w.wpre);
    hipLaunchKernelGGL(state_ids_assign_kernel, dim3(nb), dim3(256), 0, st, w.slot, w.bits, w.wpre, N, ids);
}
int launch_state_ids(const int32_t* cells, const unsigned long long* hash, int64_t N, int D, int64_t max_states, void* workspace, int32_t* ids,
                     int64_t* out, hipStream_t st) {
    if (N == 0) return 0;
    const StateIdWs w = state_ids_layout(workspace, N, max_states);
    const unsigned nb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(state_ids_clear_kernel, dim3((unsigned)((w.cap + 255) / 256)), dim3(256), 0, st, w.tab, w.cap, out);
    const bool vec4 = D % 4 == 0 && (reinterpret_cast<uintptr_t>(cells) & 15u) == 0;
    hipLaunchKernelGGL(vec4 ? state_ids_insert_kernel<true> : state_ids_insert_kernel<false>, dim3(nb), dim3(256), 0, st, cells, hash,
                       N, D, w.tab, w.cap, w.slot, out);
    finish_state_ids(cells, N, D, w, ids, out, st);
    return 0;
}
// observations -> cells -> ids in one call: the cells kernel inserts its rows itself (D a multiple of 4, obs / cells 16-byte aligned)
int launch_index_states(const double* obs, int64_t N, int D, const double* width, int64_t max_states, void* workspace, int32_t* cells,
                        int32_t* ids, int64_t* out, hipStream_t st) {
    if (N == 0) return 0;
    const StateIdWs w = state_ids_layout(workspace, N, max_states);
    allow_large_lds_once();
    hipLaunchKernelGGL(state_ids_clear_kernel, dim3((unsigned)((w.cap + 255) / 256)), dim3(256), 0, st, w.tab, w.cap, out);
    hipLaunchKernelGGL(state_cells_hash_kernel<true>, dim3((unsigned)((N + CH_WAVES * WAVE - 1) / (CH_WAVES * WAVE))), dim3(CH_WAVES * WAVE),
                       (unsigned)(CH_WAVES * WAVE * (D + 4) * 4), st, obs, N, D, width, cells, nullptr, w.tab, w.cap, w.slot, out);
    finish_state_ids(cells, N, D, w, ids, out, st);
    return 0;
}

// ---- per-rank global statistics of the per-state summaries ("when only global statistics are wanted") ----
// {states whose arg-max has left the rule action (S1:98-99), sum of max V (f64), histogram of the arg-max policy} for a
// rank's block of states: 144 bytes per rank go through the all-gather instead of 12 bytes per state.  Counts: ballot +
// popcount per candidate (wave-uniform, no atomics); the f64 sum: per-lane partials in grid-stride order, wavefront
// butterfly, one partial per block, reduced by ONE block in block order -> run-to-run identical for a given S.
constexpr int SUMMARY_BLOCKS = 512;
__global__ __launch_bounds__(256) void summary_partial_kernel(const int32_t* __restrict__ amax, const float* __restrict__ vmax,
                                                              const int32_t* __restrict__ act_step, int S, int A,
                                                              dcarl_summary_t* __restrict__ part) {
    __shared__ dcarl_summary_t wsum[256 / WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    long long activated = 0;
    double sum = 0.0;
    int hist = 0;                                               // lane a of the wavefront accumulates candidate a's count
    for (int base = (blockIdx.x * 256 + wv * WAVE); base < S; base += gridDim.x * 256) {
        const int s = base + lane;
        const bool in = s < S;
        const int a = in ? amax[s] : -1;
        sum += in ? (double)vmax[s] : 0.0;
        activated += __popcll(__ballot(in && act_step[s] >= 0)) * (lane == 0);
        for (int c = 0; c < A; ++c) {                           // S1:94 arg-max policy per state -> histogram
            const int n = __popcll(__ballot(a == c));
            hist += (lane == c) ? n : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < DCARL_MAX_ACTIONS) wsum[wv].policy_hist[lane] = hist;
    if (lane == 0) { wsum[wv].activated = activated; wsum[wv].sum_vmax = sum; }
    __syncthreads();
    if (threadIdx.x < DCARL_MAX_ACTIONS) {
        long long h = 0;
        for (int w = 0; w < 256 / WAVE; ++w) h += wsum[w].policy_hist[threadIdx.x];
        part[blockIdx.x].policy_hist[threadIdx.x] = h;
    }
    if (threadIdx.x == 0) {
        long long n = 0;
        double t = 0.0;
        for (int w = 0; w < 256 / WAVE; ++w) { n += wsum[w].activated; t += wsum[w].sum_vmax; }
        part[blockIdx.x].activated = n;
        part[blockIdx.x].sum_vmax = t;
    }
}
__global__ __launch_bounds__(64) void summary_final_kernel(const dcarl_summary_t* __restrict__ part, int nblocks,
                                                           dcarl_summary_t* __restrict__ out) {
    const int lane = threadIdx.x;
    if (lane < DCARL_MAX_ACTIONS) {
        long long h = 0;
        for (int b = 0; b < nblocks; ++b) h += part[b].policy_hist[lane];
        out->policy_hist[lane] = h;
    }
    if (lane == 0) {
        long long n = 0;
        double t = 0.0;
        for (int b = 0; b < nblocks; ++b) { n += part[b].activated; t += part[b].sum_vmax; }   // block order: deterministic
        out->activated = n;
        out->sum_vmax = t;
    }
}
static int summary_blocks(int S) { const int b = (int)ceil_div64(S, 256); return b < 1 ? 1 : (b < SUMMARY_BLOCKS ? b : SUMMARY_BLOCKS); }
int64_t summary_workspace_bytes(int64_t S) { return (int64_t)summary_blocks((int)(S > 0x7fffffff ? 0x7fffffff : S)) * sizeof(dcarl_summary_t); }
int launch_summary_stats(const int32_t* amax, const float* vmax, const int32_t* act_step, int S, int A, void* ws,
                         dcarl_summary_t* out, hipStream_t st) {
    const int nb = summary_blocks(S);
    dcarl_summary_t* part = reinterpret_cast<dcarl_summary_t*>(ws);
    hipLaunchKernelGGL(summary_partial_kernel, dim3(nb), dim3(256), 0, st, amax, vmax, act_step, S, A, part);
    hipLaunchKernelGGL(summary_final_kernel, dim3(1), dim3(64), 0, st, part, nb, out);
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
