// Record ingest: the reference's arrival-ordered (N,4) float64 table {state idx, state feature, action, cumulative reward}
// -> the device layouts of this library, without a library sort, a permutation array or a copy of the table.
//
// What the reference does with the table is `data_state_act[idx][act].append(R)` row by row (S1:73-80): a STABLE grouping
// by state (online path) or by (state, action) (final-state path).  Here that is a hand-written least-significant-digit
// radix sort of compact records {key = state << 5 | action (u32), reward (f32 / f64), arrival index (u32, optional)}:
//
//   ingest_compact_kernel   reads the 32-byte rows once, validates the ids (S1:80 would raise IndexError) and the rewards,
//                           writes the compact records and the first digit's per-block histogram;
//   rx_hist / rx_scan / rx_scatter   one stable partition pass per digit (<= 8 bits): per-block digit histograms in LDS,
//                           one scan per digit over the blocks, then the ranked scatter — a block walks its records in
//                           tiles of 8 192, a wave ranks its 64 keys with one ballot per digit bit (lanes holding the same
//                           digit form a peer mask; rank = peers below the lane), per-wave digit counters in LDS carry the
//                           ranks across the 16 groups of a wave, the tile is re-ordered by digit in LDS and leaves as
//                           runs of consecutive addresses;
//   run_bounds_kernel       first / one-past-last position of every state (or bucket) in the sorted stream -> counts;
//   table mode: slots = states by descending stream length (the same radix passes over S {length, state} pairs), rows per
//                           slice, prefix sums, and ingest_pack_kernel: a wave owns 32 rows of one slice, reads each
//                           state's 32 records as one contiguous 128-byte piece, transposes them through LDS and writes
//                           whole 1 KiB rows of the sliced layout e(s,t) (include/dcarl.h), padding as zeros;
//   bucket mode: the last pass writes the rewards straight into the caller's CSR value array; seg_off = scan of the counts.
//
// No global atomics on the data path (the only ones reduce id ranges / flags once per wave), positions are u32 (N < 2^31).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#ifndef DCARL_DP_NT
// partition: bit 0 non-temporal loads of the rows, bit 1 non-temporal stores of the partitioned records (nothing reads them before the whole
// table is through).  Same-box A/B of three builds (tools/experiments/ab_nt_legs2.sh): stores 20.9 -> 20.6 ms end to end on configs[1], 25.6 -> 24.5 on the
// random order; loads +2 ms (the rows' lines are shared by the two loads of a row and by neighbouring lanes: they need the cache)
#define DCARL_DP_NT 2
#endif

namespace dcarl {

namespace {

constexpr int RX_THREADS = 512;
[[maybe_unused]] constexpr int RX_WAVES = RX_THREADS / WAVE;        // 8
constexpr int RX_GROUPS = 16;                      // 64-record groups per wave and tile
constexpr int RX_TILE = RX_THREADS * RX_GROUPS;    // 8 192 records
constexpr int RX_DIGITS = 256;
constexpr int RX_MAXBLK = 2048;
constexpr int ACT_BITS = 5;                        // key = state << 5 | action (A <= 32)
constexpr int PACK_ROWS = 32;                      // rows of a slice one wave of ingest_pack_kernel transposes

__host__ __device__ inline int bits_for(int64_t n_values) {      // bits needed for values 0 .. n_values-1
    int b = 0;
    while (((int64_t)1 << b) < n_values) ++b;
    return b;
}

// records of a group from its first / one-past-last position; a group without records holds (0, 0) or (~0, 0)
__device__ __forceinline__ uint32_t run_len(uint32_t first, uint32_t end) { return end > first ? end - first : 0u; }

// ---- block-wide helpers (RX_THREADS threads) -----------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const uint32_t o = __shfl_up(v, off); if (lane >= off) v += o; }
    return v;
}
// exclusive scan over the block's threads; wsum = RX_WAVES + 1 words of LDS; *total = sum of all
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == WAVE - 1) wsum[wid] = inc;
    __syncthreads();
    uint32_t carry = 0, tot = 0;
    for (int i = 0; i < nw; ++i) { const uint32_t s = wsum[i]; if (i < wid) carry += s; tot += s; }
    *total = tot;
    return carry + inc - v;
}

// ---- info block (int64[16], device) -----------------------------------------------------------------------------------
enum { I_ROWS = 0, I_BANDS = 1, I_MAXLEN = 2, I_MAXACT = 3, I_MINSTATE = 4, I_MAXSTATE = 5, I_MINACT = 6, I_FLAGS = 7,
       I_N = 8, I_KEPT = 9, I_COUNT = 16 };     // I_KEPT: records the pairs ingest kept (idx != -1); the row ingest keeps all N

__global__ void ingest_init_info_kernel(int64_t* info, int64_t n) {
    const int i = threadIdx.x;
    if (i >= I_COUNT) return;
    int64_t v = 0;
    if (i == I_MAXACT || i == I_MAXSTATE) v = INT64_MIN;
    if (i == I_MINSTATE || i == I_MINACT) v = INT64_MAX;
    if (i == I_N) v = n;
    info[i] = v;
}

// ---- pass 0: rows -> compact records + first histogram -----------------------------------------------------------------
template <typename T, bool ARRIVAL, bool PAIRS = false>
__global__ __launch_bounds__(RX_THREADS) void ingest_compact_kernel(
    const double* __restrict__ data, uint32_t n, int S, int A, uint32_t blk, int shift, int bits,
    uint32_t* __restrict__ key, T* __restrict__ val, uint32_t* __restrict__ idx, int32_t* __restrict__ rec_state,
    uint32_t* __restrict__ hist, int nblk, int64_t* __restrict__ info) {
    __shared__ uint32_t h[RX_DIGITS];
    if (threadIdx.x < RX_DIGITS) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * blk, hi = (n - lo < blk) ? n : lo + blk;
    const uint32_t mask = (1u << bits) - 1u;
    int smin = INT32_MAX, smax = INT32_MIN, amin = INT32_MAX, amax = INT32_MIN;
    uint32_t flags = 0;
    constexpr int UNR = 4;                                         // rows in flight per thread
    for (uint32_t p0 = lo + threadIdx.x; p0 < hi; p0 += UNR * RX_THREADS) {
        uint4 q01[UNR], q23[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t p = p0 + u * RX_THREADS;
            if (p < hi) {
                q01[u] = reinterpret_cast<const uint4*>(data)[2 * (size_t)p];
                q23[u] = reinterpret_cast<const uint4*>(data)[2 * (size_t)p + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint32_t p = p0 + u * RX_THREADS;
            if (p >= hi) break;
            // the row as raw words: the NaN / Inf tests below are INTEGER tests on bits that never were a double for the
            // compiler (the library is built -fno-honor-nans, under which a test on a double may be folded away)
            const uint4 r01 = q01[u], r23 = q23[u];
            const bool s_nf = (r01.y & 0x7ff00000u) == 0x7ff00000u, a_nf = (r23.y & 0x7ff00000u) == 0x7ff00000u,
                       w_nf = (r23.w & 0x7ff00000u) == 0x7ff00000u;
            const double sd = __hiloint2double((int)r01.y, (int)r01.x), ad = __hiloint2double((int)r23.y, (int)r23.x),
                         wd = __hiloint2double((int)r23.w, (int)r23.z);
            // idx = int(idx_ori), act = int(act_ori) (S1:77-78): truncation toward zero; ids outside the table are reported
            // (the reference raises IndexError at S1:80; negative ids would wrap there and are refused here)
            // (32-bit conversions: ids beyond +-2e9 are out of range for any table and are reported as INT32_MIN / INT32_MAX)
            const bool s_fin = !s_nf, a_fin = !a_nf;
            const int si = !s_fin ? INT32_MIN : fabs(sd) < 2.0e9 ? (int)sd : (r01.y >> 31) ? INT32_MIN : INT32_MAX;
            const int ai = !a_fin ? INT32_MIN : fabs(ad) < 2.0e9 ? (int)ad : (r23.y >> 31) ? INT32_MIN : INT32_MAX;
            smin = si < smin ? si : smin; smax = si > smax ? si : smax;
            amin = ai < amin ? ai : amin; amax = ai > amax ? ai : amax;
            if (!s_fin || !a_fin) flags |= 2u;
            const uint32_t s = (si >= 0 && si < S) ? (uint32_t)si : 0u, a = (ai >= 0 && ai < A) ? (uint32_t)ai : 0u;
            const T r = (T)wd;
            if (w_nf || (sizeof(T) == 4 && fabs(wd) > 3.4028234663852886e38)) flags |= 1u;      // NaN / Inf, or beyond the f32 range
            const uint32_t k = (s << ACT_BITS) | a;
            if constexpr (PAIRS) {                                 // {key, f32 value} as one 8-byte record (rx_scatter_lines_kernel)
                reinterpret_cast<uint2*>(key)[p] = make_uint2(k, __float_as_uint((float)r));
            } else {
                key[p] = k;
                val[p] = r;
            }
            if (ARRIVAL) { if (!PAIRS) idx[p] = p; rec_state[p] = (int32_t)s; }   // (pairs: the passes log where every record goes instead)
            if (bits) atomicAdd(&h[(k >> shift) & mask], 1u);
        }
    }
    // id ranges / flags: one set of atomics per wave
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const int a0 = __shfl_xor(smin, off), a1 = __shfl_xor(smax, off), a2 = __shfl_xor(amin, off), a3 = __shfl_xor(amax, off);
        smin = a0 < smin ? a0 : smin; smax = a1 > smax ? a1 : smax; amin = a2 < amin ? a2 : amin; amax = a3 > amax ? a3 : amax;
        flags |= __shfl_xor(flags, off);
    }
    if ((threadIdx.x & 63) == 0 && lo < hi) {
        __hip_atomic_fetch_min(&info[I_MINSTATE], (int64_t)smin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&info[I_MAXSTATE], (int64_t)smax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_min(&info[I_MINACT], (int64_t)amin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&info[I_MAXACT], (int64_t)amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flags) __hip_atomic_fetch_or(&info[I_FLAGS], (int64_t)flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bits) {
        __syncthreads();
        if (threadIdx.x < (1 << bits)) hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
    }
}

// ---- one radix pass ----------------------------------------------------------------------------------------------------
// (stride = words from one key to the next: 1, or 2 when the records are {key, value} pairs)
__global__ __launch_bounds__(RX_THREADS) void rx_hist_kernel(const uint32_t* __restrict__ key, uint32_t n, int shift, int bits,
                                                             uint32_t blk, uint32_t* __restrict__ hist, int nblk, int stride = 1) {
    __shared__ uint32_t h[RX_DIGITS];
    if (threadIdx.x < RX_DIGITS) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * blk, hi = (n - lo < blk) ? n : lo + blk;
    const uint32_t mask = (1u << bits) - 1u;
    for (uint32_t p = lo + threadIdx.x; p < hi; p += RX_THREADS) atomicAdd(&h[(key[(size_t)p * stride] >> shift) & mask], 1u);
    __syncthreads();
    if (threadIdx.x < (1 << bits)) hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// block d: exclusive scan of digit d's counts over the blocks (in place), total -> tot[d].  nblk <= RX_MAXBLK.
__global__ __launch_bounds__(256) void rx_scan_kernel(uint32_t* __restrict__ hist, int nblk, uint32_t* __restrict__ tot) {
    __shared__ uint32_t wsum[8];
    constexpr int ITEMS = RX_MAXBLK / 256;
    uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    uint32_t x[ITEMS], run = 0;
    const int base = threadIdx.x * ITEMS;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { x[i] = (base + i < nblk) ? row[base + i] : 0u; run += x[i]; }
    uint32_t total;
    uint32_t ex = block_excl_scan(run, wsum, &total);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { if (base + i < nblk) row[base + i] = ex; ex += x[i]; }
    if (threadIdx.x == 0) tot[blockIdx.x] = total;
}

template <int VB> struct Word { using type = uint32_t; };
template <> struct Word<8> { using type = uint64_t; };

// TH threads per block: 512 (tile of 8 192 records, two blocks per CU) or 256 (tile of 4 096, four blocks per CU); the
// launcher chooses (launch_scatter).
template <int VB, bool IDX, int TH>
constexpr unsigned rx_scatter_lds() {
    return ((TH / WAVE) * RX_DIGITS + 3 * RX_DIGITS + 16) * 4 + TH * RX_GROUPS * (4 + VB + (IDX ? 4 : 0));
}

// The ranked scatter of one pass.  Stable: a block owns a contiguous range of the input, walks it in order, and inside a
// tile wave w owns records [1024 w, 1024 w + 1024), group g of it the next 64, lane l the l-th of those.
template <int VB, bool IDX, int TH>
__global__ __launch_bounds__(TH) void rx_scatter_kernel(
    const uint32_t* __restrict__ key_in, const void* __restrict__ val_in_, const uint32_t* __restrict__ idx_in,
    uint32_t* __restrict__ key_out, void* __restrict__ val_out_, uint32_t* __restrict__ idx_out, uint32_t n, int shift,
    int bits, uint32_t blk, const uint32_t* __restrict__ hist, int nblk, const uint32_t* __restrict__ tot) {
    using V = typename Word<VB>::type;
    constexpr int NWV = TH / WAVE, TILE = TH * RX_GROUPS;
    const V* __restrict__ val_in = static_cast<const V*>(val_in_);
    V* __restrict__ val_out = static_cast<V*>(val_out_);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(smem);            // [NWV][RX_DIGITS]
    uint32_t* tile_off = wcnt + NWV * RX_DIGITS;              // [RX_DIGITS]
    uint32_t* gbase = tile_off + RX_DIGITS;                        // [RX_DIGITS] next free position of digit d for this block
    uint32_t* gdst = gbase + RX_DIGITS;                            // [RX_DIGITS] gbase - tile_off of the current tile
    uint32_t* wsum = gdst + RX_DIGITS;                             // [16]
    V* s_val = reinterpret_cast<V*>(wsum + 16);                    // [TILE]   (offset is a multiple of 8 bytes)
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_val + TILE);
    uint32_t* s_idx = s_key + TILE;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ndig = 1 << bits;
    const uint32_t mask = (uint32_t)ndig - 1u;
    const uint32_t lo = blockIdx.x * blk, hi = (n - lo < blk) ? n : lo + blk;
    {   // first position of each digit for this block = digits below (all blocks) + this digit in earlier blocks
        uint32_t total;
        const uint32_t t = tid < ndig ? tot[tid] : 0u;
        const uint32_t ex = block_excl_scan(t, wsum, &total);
        if (tid < ndig) gbase[tid] = ex + hist[(size_t)tid * nblk + blockIdx.x];
    }
    uint32_t* mycnt = wcnt + wv * RX_DIGITS;
    // the tile's records live in registers from the load to the staging writes; the NEXT tile's loads are issued right after
    // those writes, so they fly under the write-out of this tile (two resident blocks per CU alone do not hide the HBM
    // round trip between the barriers of a tile)
    uint32_t k[RX_GROUPS];
    V v[RX_GROUPS];
    uint32_t ix[IDX ? RX_GROUPS : 1];
    // FULL = every record of the tile exists (all tiles of a block but its last): no per-lane guards, no exec juggling
    auto load_tile = [&](uint32_t t0, auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        const uint32_t b0 = t0 + (uint32_t)wv * (RX_GROUPS * WAVE) + lane;
#pragma unroll
        for (int g = 0; g < RX_GROUPS; ++g) {
            const uint32_t p = b0 + g * WAVE;
            const bool ok = FULL || p < hi;
#if defined(RX_EXP) && RX_EXP == 5
            { const uint2 kv = ok ? reinterpret_cast<const uint2*>(key_in)[p] : make_uint2(0u, 0u); k[g] = kv.x; v[g] = (V)kv.y; }   // EXPERIMENT 5: {key, value} as one 8-byte record
#else
            k[g] = ok ? key_in[p] : 0u;
            v[g] = ok ? val_in[p] : (V)0;
            if (IDX) ix[g] = ok ? idx_in[p] : 0u;
#endif
        }
    };
    auto load_any = [&](uint32_t t0) __attribute__((always_inline)) {
        if (t0 + TILE <= hi) load_tile(t0, std::true_type{}); else load_tile(t0, std::false_type{});
    };
    uint32_t local[RX_GROUPS];
    // ranks of the tile's records among this wave's records of the same digit.  All 8 digit bits are balloted (bits above
    // `bits` are zero in every lane and cost one no-op round each: the loop is straight-line code)
    auto rank_tile = [&](uint32_t base, auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int g = 0; g < RX_GROUPS; ++g) {
            const bool ok = FULL || base + g * WAVE < hi;
            const uint32_t d = (k[g] >> shift) & mask;
            unsigned long long peers = FULL ? ~0ull : __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(bit && ok);
                peers &= bit ? m : ~m;
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            const uint32_t cnt = (uint32_t)__popcll(peers);
            uint32_t old = 0;
            if (FULL || ok) old = mycnt[d];
            __builtin_amdgcn_wave_barrier();                       // every peer has read before the first of them writes
            if ((FULL || ok) && below == 0) mycnt[d] = old + cnt;
            __builtin_amdgcn_wave_barrier();
            local[g] = old + below;                                // rank among this wave's records of digit d in the tile
        }
    };
    if (lo < hi) load_any(lo);
    for (uint32_t t0 = lo; t0 < hi; t0 += TILE) {
        const uint32_t base = t0 + (uint32_t)wv * (RX_GROUPS * WAVE) + lane;
        const bool full = t0 + TILE <= hi;                        // block-uniform
#pragma unroll
        for (int i = 0; i < RX_DIGITS / WAVE; ++i) mycnt[lane + i * WAVE] = 0;
#if defined(RX_EXP) && RX_EXP == 3
        // EXPERIMENT 3: no ranking, no staging: the tile leaves in input order (a copy with this grid and these registers)
#pragma unroll
        for (int g = 0; g < RX_GROUPS; ++g) { const uint32_t p = base + g * WAVE; if (p < hi) { key_out[p] = k[g]; val_out[p] = v[g]; } }
        if (t0 + TILE < hi) load_any(t0 + TILE);
        continue;
#endif
        if (full) rank_tile(base, std::true_type{}); else rank_tile(base, std::false_type{});
        __syncthreads();
        uint32_t run = 0;
        if (tid < ndig) {
            for (int w = 0; w < NWV; ++w) { const uint32_t c = wcnt[w * RX_DIGITS + tid]; wcnt[w * RX_DIGITS + tid] = run; run += c; }
        }
        uint32_t tile_n;
        const uint32_t ex = block_excl_scan(run, wsum, &tile_n);
        if (tid < ndig) {
            tile_off[tid] = ex;
            const uint32_t gb = gbase[tid];
            gdst[tid] = gb - ex;
            gbase[tid] = gb + run;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < RX_GROUPS; ++g) {
            if (full || base + g * WAVE < hi) {
                const uint32_t d = (k[g] >> shift) & mask;
                const uint32_t pos = tile_off[d] + mycnt[d] + local[g];
                s_key[pos] = k[g];
                s_val[pos] = v[g];
                if (IDX) s_idx[pos] = ix[g];
            }
        }
        if (t0 + TILE < hi) load_any(t0 + TILE);                  // (block-uniform)
        __syncthreads();
        for (uint32_t i = tid; i < tile_n; i += TH) {
            const uint32_t kk = s_key[i];
#if defined(RX_EXP) && RX_EXP == 4
            const uint32_t dst = t0 + i;                           // EXPERIMENT 4: everything but the scattered destination (sequential writes)
#else
            const uint32_t dst = gdst[(kk >> shift) & mask] + i;
#endif
#if defined(RX_EXP) && RX_EXP == 1
            if (kk == 0xdeadbeefu)                                 // EXPERIMENT 1: no global stores
#endif
            {
#if defined(RX_EXP) && RX_EXP == 5
            reinterpret_cast<uint2*>(key_out)[dst] = make_uint2(kk, (uint32_t)s_val[i]);
#else
            key_out[dst] = kk;
            val_out[dst] = s_val[i];
            if (IDX) idx_out[dst] = s_idx[i];
#endif
            }
        }
        __syncthreads();
    }
}

// ---- the same pass on 8-byte {key, f32 value} records, every global store a whole aligned piece ----------------------------
// What bounds rx_scatter_kernel is where its stores land (tools/experiments/ubench_runs.hip: 256 streams per block fed in 128-byte runs at
// random alignment take 2.96 TB/s read + write, 64-byte-aligned runs 3.76, line-aligned runs 4.46; the two-array pass itself
// measures 2.6 to 3.4 TB/s from box to box): a digit's records of one tile begin wherever the previous tile's ended, so nearly
// every line is written in two pieces by two tiles.  Here the records are pairs in ONE array and a block writes a digit's
// records only in whole units of LN_REC records (8 = 64 bytes; 16 = a line): what a tile leaves over (< LN_REC per digit) waits
// in registers of the digit's owner threads and is staged in front of the digit's records of the next tile.  Partial units
// remain at the two ends of a (block, digit) range only.  Measured (tools/experiments/ubench_scatter_lines.hip, 2^28 random keys): 3.7-3.8
// TB/s on every box, for 64-byte units with tiles of 6 656, 128-byte units with tiles of 4 608 and 1 024-thread blocks alike;
// without its stores the pass takes 70-80 % of that time (random LDS accesses: SQ_LDS_IDX_ACTIVE), so both sides are near
// their ends.  f32 values without arrival indices (the big tables); everything else keeps rx_scatter_kernel.
//
// Ranks without ballots: every lane ORs its lane bit into the wave's 64-bit word of its digit (a commutative LDS atomic: the
// word does not depend on the order the lanes land in), reads the word back — the lanes of this group holding the same digit —
// and counts the bits below its own; the group's lowest such lane clears the word and advances the wave's digit counter.  (LDS
// serves a wave's instructions in order: the read sees every lane's OR, the clear comes after every lane's read.)  The words
// live in the staging buffer, free while a tile is ranked.  204 M VALU instructions per 2^28 records where the ballot form
// (8 ballots per group) has 457 M; mixing the two forms to balance VALU against LDS work changed nothing (1.15-1.20 ms for
// 0 ... 13 of 13 groups by ballots).
//
// BOUNDS (the LAST pass of a table sort): the pass also reports where every state's records begin and end in its output
// (start / end1, what run_bounds_kernel computes from the sorted stream in a pass of its own): a record whose predecessor in
// its digit's stream belongs to another state starts a run and ends the predecessor's.  The predecessor is the staged
// neighbour, or the last record the block wrote for the digit in an earlier tile (lastg); at the two ends of a (block, digit)
// range it is another block's, so those two reports are atomicMin / atomicMax (start initialised to ~0, end1 to 0) — every
// other report is the run's true first / one-past-last position, which the min / max cannot move.
// group id of a key: the state (table mode, A == 0) or state*A + action (bucket mode)
__device__ __forceinline__ uint32_t group_of(uint32_t k, int A) {
    return A ? (k >> ACT_BITS) * (uint32_t)A + (k & ((1u << ACT_BITS) - 1u)) : (k >> ACT_BITS);
}
template <int TH, int G, int LN_REC, bool BOUNDS>
constexpr unsigned rx_lines_lds() {
    return ((TH / WAVE) * RX_DIGITS + (BOUNDS ? 7 : 4) * RX_DIGITS + 16) * 4 + (TH * G + RX_DIGITS * (LN_REC - 1)) * 8;
}
constexpr uint32_t NO_GROUP = 0xffffffffu;
// VAL_ONLY (the last pass of a bucket sort): only the values leave, as 4-byte words into the caller's CSR array (rec_out points
// at it; LN_REC then counts 4-byte words: 16 = 64 bytes) — the keys have served once the runs are reported.  A: group_of's.
// DST (tables with arrival bookkeeping): the pass also logs, in INPUT order, the position every record goes to — a coalesced
// 4-byte store per record; the arrival -> position map is the passes' logs composed (arrival_positions_kernel), which replaces
// carrying an arrival index through every pass and scattering 12 bytes per record by it at the end.
template <int TH, int G, int LN_REC, bool BOUNDS, bool VAL_ONLY = false, bool DST = false>
__global__ __launch_bounds__(TH) void rx_scatter_lines_kernel(
    const uint2* __restrict__ rec_in, uint2* __restrict__ rec_out, uint32_t n, int shift, int bits, uint32_t blk,
    const uint32_t* __restrict__ hist, int nblk, const uint32_t* __restrict__ tot, uint32_t* __restrict__ start,
    uint32_t* __restrict__ end1, int A, uint32_t* __restrict__ dst_log) {
    static_assert(!VAL_ONLY || BOUNDS, "without the keys the runs must be reported here");
    uint32_t* __restrict__ val_out = reinterpret_cast<uint32_t*>(rec_out);
    constexpr int NWV = TH / WAVE, TILE = TH * G;
    constexpr int OWN = TH / RX_DIGITS;                            // owner threads per digit (2 or 4)
    constexpr int SLOTS = LN_REC / OWN;                            // waiting records each of them can hold
    static_assert(TH % RX_DIGITS == 0 && LN_REC % OWN == 0, "owner threads tile the digits");
    // per digit, rewritten every tile: x = global index - staged index, y = staged end of the records that leave,
    // (BOUNDS) z = staged position of the digit's first record, w = state of the last record written in an earlier tile
    using DW = typename std::conditional<BOUNDS, uint4, uint2>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DW* dstw = reinterpret_cast<DW*>(smem);                        // [RX_DIGITS]
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(dstw + RX_DIGITS);    // [NWV][RX_DIGITS] per-wave digit counts -> staged position of the wave's first record of the digit
    uint32_t* gpos = wcnt + NWV * RX_DIGITS;                       // [RX_DIGITS] global index of the digit's first unwritten record
    uint32_t* pnd = gpos + RX_DIGITS;                              // [RX_DIGITS] records of the digit waiting for their unit to fill
    uint32_t* lastg = pnd + RX_DIGITS;                             // [RX_DIGITS] (BOUNDS) state of the digit's last written record
    uint32_t* wsum = lastg + (BOUNDS ? RX_DIGITS : 0);             // [16]
    uint2* s_rec = reinterpret_cast<uint2*>(wsum + 16);            // [TILE + RX_DIGITS * (LN_REC - 1)]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ndig = 1 << bits;
    const uint32_t mask = (uint32_t)ndig - 1u;
    const uint32_t lo = blockIdx.x * blk, hi = (n - lo < blk) ? n : lo + blk;
    {
        uint32_t total;
        const uint32_t t = tid < ndig ? tot[tid] : 0u;
        const uint32_t ex = block_excl_scan(t, wsum, &total);
        if (tid < RX_DIGITS) {
            gpos[tid] = tid < ndig ? ex + hist[(size_t)tid * nblk + blockIdx.x] : 0u;
            pnd[tid] = 0u;
            if (BOUNDS) lastg[tid] = NO_GROUP;
        }
    }
    uint32_t* mycnt = wcnt + wv * RX_DIGITS;
    const int od = tid / OWN, oslot = (tid % OWN) * SLOTS;         // this thread keeps waiting records [oslot, oslot + SLOTS) of digit od
    uint2 pend[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) pend[q] = make_uint2(0u, 0u);
    uint32_t pcnt = 0;                                             // waiting records of digit od (the same number in all its owners)
    uint2 r[G];
    uint32_t local[G];
    auto load_tile = [&](uint32_t t0, auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        const uint32_t b0 = t0 + (uint32_t)wv * (G * WAVE) + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t p = b0 + g * WAVE;
            r[g] = (FULL || p < hi) ? rec_in[p] : make_uint2(0u, 0u);
        }
    };
    auto load_any = [&](uint32_t t0) __attribute__((always_inline)) {
        if (t0 + TILE <= hi) load_tile(t0, std::true_type{}); else load_tile(t0, std::false_type{});
    };
    unsigned long long* wmask = reinterpret_cast<unsigned long long*>(s_rec) + wv * RX_DIGITS;
    const unsigned long long lane_bit = 1ull << lane;
    auto rank_tile = [&](uint32_t base, auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int i = 0; i < RX_DIGITS / WAVE; ++i) wmask[lane + i * WAVE] = 0ull;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool ok = FULL || base + g * WAVE < hi;
            const uint32_t d = (r[g].x >> shift) & mask;
            unsigned long long peers = 0ull;
            uint32_t old = 0;
            if (ok) __hip_atomic_fetch_or(&wmask[d], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            if (ok) { peers = wmask[d]; old = mycnt[d]; }
            __builtin_amdgcn_wave_barrier();
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            if (ok && below == 0) { wmask[d] = 0ull; mycnt[d] = old + (uint32_t)__popcll(peers); }
            __builtin_amdgcn_wave_barrier();
            local[g] = old + below;
        }
    };
    // one record leaves for global index dst; prev = state of its predecessor in the digit's stream (NO_GROUP: another block's)
    auto report = [&](uint32_t g, uint32_t prev, uint32_t dst) __attribute__((always_inline)) {
        if (prev == g) return;
        if (prev == NO_GROUP) {
            atomicMin(&start[g], dst);
        } else {
            start[g] = dst;
            end1[prev] = dst;
        }
    };
    if (lo < hi) load_any(lo);
    for (uint32_t t0 = lo; t0 < hi; t0 += TILE) {
        const uint32_t base = t0 + (uint32_t)wv * (G * WAVE) + lane;
        const bool full = t0 + TILE <= hi;
#pragma unroll
        for (int i = 0; i < RX_DIGITS / WAVE; ++i) mycnt[lane + i * WAVE] = 0;
        if (full) rank_tile(base, std::true_type{}); else rank_tile(base, std::false_type{});
        __syncthreads();
        // per digit: c new records behind p waiting ones; the p + c records are staged contiguously and the part that
        // completes units leaves
        uint32_t c = 0, p_old = 0;
        if (tid < RX_DIGITS) {
#pragma unroll
            for (int w = 0; w < NWV; ++w) c += wcnt[w * RX_DIGITS + tid];
            p_old = pnd[tid];
        }
        uint32_t staged_n;
        const uint32_t so = block_excl_scan(c + p_old, wsum, &staged_n);
        if (tid < RX_DIGITS) {
            const uint32_t g0 = gpos[tid], end = g0 + p_old + c;
            const uint32_t wend = end & ~(uint32_t)(LN_REC - 1);
            const uint32_t wl = wend > g0 ? wend - g0 : 0u;        // records of this digit that leave now (whole units, but for the block's first)
            if constexpr (BOUNDS) dstw[tid] = make_uint4(g0 - so, so + wl, so, lastg[tid]);
            else dstw[tid] = make_uint2(g0 - so, so + wl);
            gpos[tid] = g0 + wl;
            pnd[tid] = p_old + c - wl;
            uint32_t at = so + p_old;                              // the digit's new records follow the waiting ones, wave by wave
#pragma unroll
            for (int w = 0; w < NWV; ++w) { const uint32_t x = wcnt[w * RX_DIGITS + tid]; wcnt[w * RX_DIGITS + tid] = at; at += x; }
        }
        __syncthreads();
        {   // stage: the waiting records first (their owners), then the tile's
            const uint32_t first = wcnt[od] - pcnt;                // (wave 0's first record of the digit follows them)
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) if ((uint32_t)(oslot + q) < pcnt) s_rec[first + oslot + q] = pend[q];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (full || base + g * WAVE < hi) {
                const uint32_t d = (r[g].x >> shift) & mask;
                const uint32_t at = mycnt[d] + local[g];
                s_rec[at] = r[g];
                if constexpr (DST) dst_log[base + g * WAVE] = dstw[d].x + at;      // global index = staged index + (g0 - so), whenever it leaves
            }
        }
        if (t0 + TILE < hi) load_any(t0 + TILE);
        __syncthreads();
        for (uint32_t i = tid; i < staged_n; i += TH) {
            const uint2 x = s_rec[i];
            const uint32_t d = (x.x >> shift) & mask;
            const DW dw = dstw[d];
#if defined(LN_EXP) && LN_EXP == 1
            if (i < dw.y && x.y == 0xdeadbeefu) rec_out[dw.x + i] = x;   // EXPERIMENT 1: no global stores (tools/experiments/ubench_scatter_lines.hip)
#else
            if (i < dw.y) {
                if constexpr (VAL_ONLY) val_out[dw.x + i] = x.y; else rec_out[dw.x + i] = x;
                if constexpr (BOUNDS) {
                    const uint32_t g = group_of(x.x, A);
                    report(g, i > dw.z ? group_of(s_rec[i - 1].x, A) : dw.w, dw.x + i);
                    if (i + 1 == dw.y) lastg[d] = g;               // (read again by the next tile's scan only)
                }
            }
#endif
        }
        {   // what stays goes back to the owners
            pcnt = pnd[od];
            const uint32_t from = dstw[od].y;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) if ((uint32_t)(oslot + q) < pcnt) pend[q] = s_rec[from + oslot + q];
        }
        __syncthreads();
    }
    // the block's last partial units: the owners put what still waits back into the (now free) staging buffer, LN_REC places
    // per digit, and the block writes them out like a tile's records
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) if ((uint32_t)(oslot + q) < pcnt) s_rec[od * LN_REC + oslot + q] = pend[q];
    __syncthreads();
    for (int i = tid; i < RX_DIGITS * LN_REC; i += TH) {
        const int d = i / LN_REC, j = i % LN_REC;
        if ((uint32_t)j < pnd[d]) {
            const uint2 x = s_rec[i];
            const uint32_t dst = gpos[d] + j;
            if constexpr (VAL_ONLY) val_out[dst] = x.y; else rec_out[dst] = x;
            if constexpr (BOUNDS) report(group_of(x.x, A), j > 0 ? group_of(s_rec[i - 1].x, A) : lastg[d], dst);
        }
    }
    if constexpr (BOUNDS) {
        // the digit's last record in this block ends a run that the next block may continue
        if (tid < RX_DIGITS) {
            const uint32_t p = pnd[tid];
            const uint32_t gl = p ? group_of(s_rec[tid * LN_REC + p - 1].x, A) : lastg[tid];
            if (gl != NO_GROUP) atomicMax(&end1[gl], gpos[tid] + p);
        }
    }
}

// ---- after the sort: where every group starts and ends ------------------------------------------------------------------
template <bool PAIRS>
__global__ __launch_bounds__(256) void run_bounds_kernel(const uint32_t* __restrict__ key, uint32_t n, int A,
                                                         uint32_t* __restrict__ start, uint32_t* __restrict__ end1) {
    const uint32_t p0 = (blockIdx.x * 256u + threadIdx.x) * 4u;      // four consecutive keys per thread (one or two 16-byte loads)
    if (p0 >= n) return;
    constexpr int ST = PAIRS ? 2 : 1;                                // words from one key to the next
    uint32_t k[4];
    if (p0 + 4 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(key + (size_t)p0 * ST);
        if (PAIRS) {
            const uint4 w = *reinterpret_cast<const uint4*>(key + (size_t)p0 * ST + 4);
            k[0] = v.x; k[1] = v.z; k[2] = w.x; k[3] = w.z;
        } else {
            k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = p0 + j < n ? key[(size_t)(p0 + j) * ST] : 0u;
    }
    uint32_t prev = p0 ? group_of(key[(size_t)(p0 - 1) * ST], A) : 0xffffffffu;
    const uint32_t nxt = p0 + 4 < n ? group_of(key[(size_t)(p0 + 4) * ST], A) : 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t p = p0 + j;
        if (p >= n) break;
        const uint32_t g = group_of(k[j], A);
        const uint32_t after = (j < 3 && p + 1 < n) ? group_of(k[j + 1], A) : (p + 1 < n ? nxt : 0xffffffffu);
        if (p == 0 || prev != g) start[g] = p;
        if (p == n - 1 || after != g) end1[g] = p + 1;
        prev = g;
    }
}

// lengths per state (+ the keys of the slot sort: descending length = ascending lmask - length)
__global__ __launch_bounds__(256) void lengths_kernel(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end1, int S,
                                                      int32_t* __restrict__ len_state, uint32_t lmask, uint32_t* __restrict__ lkey,
                                                      uint32_t* __restrict__ lval, int64_t* __restrict__ info) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    uint32_t len = 0;
    if (s < S) {
        len = run_len(start[s], end1[s]);
        len_state[s] = (int32_t)len;
        if (lkey) { lkey[s] = lmask - len; lval[s] = (uint32_t)s; }
    }
    uint32_t m = len;
#pragma unroll
    for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) __hip_atomic_fetch_max(&info[I_MAXLEN], (int64_t)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same from given lengths (dcarl_slot_order): keys of the slot sort + the longest stream
__global__ __launch_bounds__(256) void lengths_given_kernel(const int32_t* __restrict__ len_state, int S, uint32_t lmask,
                                                            uint32_t* __restrict__ lkey, uint32_t* __restrict__ lval, int64_t* __restrict__ info) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    uint32_t len = 0;
    if (s < S) {
        len = (uint32_t)(len_state[s] < 0 ? 0 : len_state[s]);
        if (lkey) { lkey[s] = lmask - (len < lmask ? len : lmask); lval[s] = (uint32_t)s; }
    }
    uint32_t m = len;
#pragma unroll
    for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) __hip_atomic_fetch_max(&info[I_MAXLEN], (int64_t)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// slot k holds state order[k] (NULL: identity)
__global__ __launch_bounds__(256) void slots_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ len_state, int S,
                                                    int32_t* __restrict__ len_slot, int32_t* __restrict__ slot_state,
                                                    int32_t* __restrict__ state_slot) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= S) return;
    const int st = order ? (int)order[k] : k;
    len_slot[k] = len_state[st];
    slot_state[k] = st;
    state_slot[st] = k;
}

// rows[w] = ceil4(longest stream of slice w); bands[w] = PACK_ROWS-row bands of it
__global__ __launch_bounds__(256) void slice_rows_kernel(const int32_t* __restrict__ len_slot, int S, int W, int64_t* __restrict__ sro,
                                                         uint32_t* __restrict__ band_off) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= W) return;
    const int s = w * WAVE + lane;
    int m = s < S ? len_slot[s] : 0;
#pragma unroll
    for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(m, off); m = o > m ? o : m; }
    if (lane == 0) {
        const int64_t rows = ((int64_t)m + 3) & ~(int64_t)3;
        sro[w + 1] = rows;
        band_off[w + 1] = (uint32_t)((rows + PACK_ROWS - 1) / PACK_ROWS);
    }
}
// in-place inclusive scans of sro[1..W] (int64) and band_off[1..W] (u32) by ONE block; totals -> info
__global__ __launch_bounds__(1024) void slice_scan_kernel(int64_t* __restrict__ sro, uint32_t* __restrict__ band_off, int W,
                                                          int64_t* __restrict__ info) {
    __shared__ int64_t ws_r[16];
    __shared__ uint32_t ws_b[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t carry_r = 0;
    uint32_t carry_b = 0;
    if (threadIdx.x == 0) { sro[0] = 0; band_off[0] = 0; }
    for (int c = 0; c < W; c += 1024) {
        const int i = c + threadIdx.x;
        int64_t r = i < W ? sro[i + 1] : 0;
        uint32_t b = i < W ? band_off[i + 1] : 0u;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int64_t o = __shfl_up(r, off);
            const uint32_t ob = __shfl_up(b, off);
            if (lane >= off) { r += o; b += ob; }
        }
        __syncthreads();
        if (lane == WAVE - 1) { ws_r[wid] = r; ws_b[wid] = b; }
        __syncthreads();
        int64_t pre_r = 0, tot_r = 0;
        uint32_t pre_b = 0, tot_b = 0;
        for (int j = 0; j < 16; ++j) { if (j < wid) { pre_r += ws_r[j]; pre_b += ws_b[j]; } tot_r += ws_r[j]; tot_b += ws_b[j]; }
        if (i < W) { sro[i + 1] = carry_r + pre_r + r; band_off[i + 1] = carry_b + pre_b + b; }
        carry_r += tot_r;
        carry_b += tot_b;
    }
    if (threadIdx.x == 0) { info[I_ROWS] = carry_r; info[I_BANDS] = (int64_t)carry_b; }
}

// unit u (one band of one slice) -> its slice: binary search in band_off
__global__ __launch_bounds__(256) void unit_slice_kernel(const uint32_t* __restrict__ band_off, int W, uint32_t units,
                                                         uint32_t* __restrict__ unit_slice) {
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u >= units) return;
    int lo = 0, hi = W;                                            // largest w with band_off[w] <= u
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (band_off[mid] <= u) lo = mid; else hi = mid; }
    unit_slice[u] = (uint32_t)lo;
}

// ---- sorted stream -> sliced layout ----------------------------------------------------------------------------------------
// A wave owns a BAND of PACK_ROWS = 32 rows of one slice: 64 states x 32 records.  Loads: lanes 0..31 take 32 consecutive records
// of state 2p, lanes 32..63 of state 2p+1 (two contiguous 128-byte pieces per instruction); the tile goes through LDS
// transposed ([state][record], row stride 36 / 68 words: the 16-byte reads of a 16-lane group hit 16 different 4-bank groups)
// and leaves as whole 1 KiB rows.  9 KiB of LDS per wave (32-row bands instead of 64: twice the resident waves per CU, which is
// what this latency-bound transposition needs).
template <int VB> constexpr int pack_stride() { return VB == 4 ? 36 : 68; }           // words per LDS tile row
constexpr int PACK_WAVES = 4;
template <int VB> constexpr unsigned pack_lds() { return PACK_WAVES * WAVE * pack_stride<VB>() * 4; }

// PAIRS: the sorted stream is ONE array of {key, f32 value} records (key points at it, val_ is unused): one load serves both phases.
template <int VB, bool IDX, bool PAIRS = false>
__global__ __launch_bounds__(PACK_WAVES * WAVE) void ingest_pack_kernel(
    const uint32_t* __restrict__ key, const void* __restrict__ val_, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ start, const int32_t* __restrict__ len_slot, const int32_t* __restrict__ slot_state,
    const int64_t* __restrict__ sro, const uint32_t* __restrict__ band_off, const uint32_t* __restrict__ unit_slice,
    uint32_t units, int S, void* __restrict__ R_, uint8_t* __restrict__ act, int64_t* __restrict__ rec_elem,
    int32_t* __restrict__ rec_t) {
    using V = typename Word<VB>::type;
    constexpr int STRIDE = pack_stride<VB>();
    const V* __restrict__ val = static_cast<const V*>(val_);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem) + wv * WAVE * STRIDE;
    const uint32_t u = blockIdx.x * PACK_WAVES + wv;
    if (u >= units) return;                                        // wave-uniform; no block-wide barrier below
    const int w = (int)unit_slice[u];
    const uint32_t t0 = (u - band_off[w]) * (uint32_t)PACK_ROWS;
    const int64_t row0 = sro[w];
    const uint32_t rows = (uint32_t)(sro[w + 1] - row0);
    const int slot = w * WAVE + lane;
    uint32_t mylen = 0, mybase = 0;
    if (slot < S) {
        mylen = (uint32_t)len_slot[slot];
        if (mylen) mybase = start[slot_state ? slot_state[slot] : slot];
    }
    const uint32_t rem = mylen > t0 ? (mylen - t0 < (uint32_t)PACK_ROWS ? mylen - t0 : (uint32_t)PACK_ROWS) : 0u;   // my state's records in this band
    const uint32_t src = mybase + t0;
    const int half = lane >> 5, r = lane & 31;                     // load role: record r of state 2p + half

    static_assert(!PAIRS || (VB == 4 && !IDX), "pairs carry f32 values and no arrival index");
    uint32_t acts[PAIRS ? 32 : 1];
    // rewards
#pragma unroll
    for (int pc = 0; pc < 32; pc += 16) {
        V x[16];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int j = 2 * (pc + pp) + half;
            const uint32_t b = __shfl(src, j), n = __shfl(rem, j);
            if constexpr (PAIRS) {
                const uint2 kv = (uint32_t)r < n ? reinterpret_cast<const uint2*>(key)[b + r] : make_uint2(0u, 0u);
                x[pp] = (V)kv.y;
                acts[pc + pp] = kv.x & ((1u << ACT_BITS) - 1u);
            } else {
                x[pp] = (uint32_t)r < n ? val[b + r] : (V)0;
            }
        }
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) reinterpret_cast<V*>(tile + (2 * (pc + pp) + half) * STRIDE)[r] = x[pp];
    }
    __builtin_amdgcn_wave_barrier();
    V* R = static_cast<V*>(R_);
#pragma unroll
    for (int q = 0; q < PACK_ROWS / 4; ++q) {
        if (t0 + 4 * q < rows) {
            const V* srcq = reinterpret_cast<const V*>(tile + lane * STRIDE) + 4 * q;
            V* dst = R + ((row0 + t0 + 4 * q) * WAVE + lane * 4);
            if (VB == 4) {
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(srcq);
            } else {
                reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(srcq)[0];
                reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(srcq)[1];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // actions (low bits of the key), arrival bookkeeping
#pragma unroll
    for (int pc = 0; pc < 32; pc += 16) {
        uint32_t x[16];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int j = 2 * (pc + pp) + half;
            const uint32_t b = __shfl(src, j), n = __shfl(rem, j);
            if constexpr (PAIRS) x[pp] = acts[pc + pp];
            else x[pp] = (uint32_t)r < n ? (key[b + r] & ((1u << ACT_BITS) - 1u)) : 0u;
            if (IDX && (uint32_t)r < n) {
                const uint32_t t = t0 + r, a = idx[b + r];
                rec_elem[a] = (row0 + (t & ~3u)) * WAVE + j * 4 + (t & 3u);
                rec_t[a] = (int32_t)t;
            }
        }
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) tile[(2 * (pc + pp) + half) * 36 + r] = x[pp];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < PACK_ROWS / 4; ++q) {
        if (t0 + 4 * q < rows) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(tile + lane * 36 + 4 * q);
            reinterpret_cast<uint32_t*>(act)[(row0 + t0 + 4 * q) * (WAVE / 4) + lane] = a4.x | (a4.y << 8) | (a4.z << 16) | (a4.w << 24);
        }
    }
}

// counts of the groups (u32, from run_bounds) -> exclusive prefix as int64 [M+1]: tile sums, one block over the tile sums, add
constexpr int CS_THREADS = 256, CS_ITEMS = 8, CS_TILE = CS_THREADS * CS_ITEMS;
__global__ __launch_bounds__(CS_THREADS) void counts_tile_kernel(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end1,
                                                                 int64_t M, int64_t* __restrict__ tile_sum) {
    __shared__ uint32_t wsum[8];
    const int64_t base = (int64_t)blockIdx.x * CS_TILE + (int64_t)threadIdx.x * CS_ITEMS;
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < CS_ITEMS; ++i) if (base + i < M) run += run_len(start[base + i], end1[base + i]);
    uint32_t total;
    block_excl_scan(run, wsum, &total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void tile_sums_scan_kernel(int64_t* __restrict__ tile_sum, int64_t ntiles) {
    __shared__ int64_t ws[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t carry = 0;
    for (int64_t c = 0; c < ntiles; c += 1024) {
        const int64_t i = c + threadIdx.x;
        const int64_t v = i < ntiles ? tile_sum[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { const int64_t o = __shfl_up(inc, off); if (lane >= off) inc += o; }
        __syncthreads();
        if (lane == WAVE - 1) ws[wid] = inc;
        __syncthreads();
        int64_t pre = 0, tot = 0;
        for (int j = 0; j < 16; ++j) { if (j < wid) pre += ws[j]; tot += ws[j]; }
        if (i < ntiles) tile_sum[i] = carry + pre + inc - v;
        carry += tot;
    }
}
__global__ __launch_bounds__(CS_THREADS) void counts_offsets_kernel(const uint32_t* __restrict__ start, const uint32_t* __restrict__ end1,
                                                                    int64_t M, const int64_t* __restrict__ tile_sum,
                                                                    int64_t* __restrict__ off) {
    __shared__ uint32_t wsum[8];
    const int64_t base = (int64_t)blockIdx.x * CS_TILE + (int64_t)threadIdx.x * CS_ITEMS;
    uint32_t x[CS_ITEMS], run = 0;
#pragma unroll
    for (int i = 0; i < CS_ITEMS; ++i) { x[i] = (base + i < M) ? run_len(start[base + i], end1[base + i]) : 0u; run += x[i]; }
    uint32_t total;
    int64_t ex = tile_sum[blockIdx.x] + block_excl_scan(run, wsum, &total);
#pragma unroll
    for (int i = 0; i < CS_ITEMS; ++i) { if (base + i < M) off[base + i] = ex; ex += x[i]; }
    if (base <= M - 1 && M - 1 < base + CS_ITEMS) off[M] = ex;     // the thread holding the last group writes the total
}

// ---- the inverse: a record table back into the reference's (N,4) float64 rows, in a given arrival order ---------------------
// arrival k -> (state, element): from rec_state / rec_elem, or (both NULL) the dense interleaving "every state receives its
// t-th record before any receives its (t+1)-th, in an order that changes with t": t = k / S, j = k % S, state = perm_t(j), e =
// e(slot(state), t).  perm_t is a bijection of [0, S) that looks random for power-of-two S — multiply by an odd constant, add a
// round constant, xor-shift, twice (each step is invertible mod 2^b) — so that neither the digits of consecutive arrivals nor
// their run lengths in a radix tile are regular (an affine map alone hands every 256 arrivals all 256 low digits once: the
// scatter passes ran 20 % faster on it than on random data); other S fall back to the affine map (j * mult + 7919 t) % S.
__device__ __forceinline__ int dense_order_state(int64_t j, int64_t t, int S, int64_t mult) {
    if ((S & (S - 1)) == 0 && S >= 4) {
        const uint32_t m = (uint32_t)S - 1u;
        const int b = 31 - __clz(S), h = (b + 1) / 2;
        uint32_t x = (uint32_t)j;
        x = (x * 0x9E3779B1u + (uint32_t)t * 0x85EBCA77u) & m;
        x ^= x >> h;
        x = (x * 0xC2B2AE3Du + ((uint32_t)t >> 3) + 0x27D4EB2Fu) & m;
        x ^= x >> h;
        x = (x * 0x165667B1u) & m;
        return (int)x;
    }
    return (int)((j * mult + t * 7919) % S);
}
template <typename T>
__global__ __launch_bounds__(256) void export_records_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ sro, const int32_t* __restrict__ state_slot,
    const double* __restrict__ state_value, int S, int64_t mult, const int32_t* __restrict__ rec_state,
    const int64_t* __restrict__ rec_elem, int64_t N, double* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    int s;
    int64_t e;
    if (rec_elem) {
        s = rec_state[k];
        e = rec_elem[k];
    } else {
        const int64_t t = k / S, j = k - t * S;
        s = dense_order_state(j, t, S, mult);
        const int slot = state_slot ? state_slot[s] : s;
        e = (sro[slot >> 6] + (t & ~(int64_t)3)) * WAVE + (int64_t)(slot & 63) * 4 + (t & 3);
    }
    double4 row;
    row.x = (double)s;
    row.y = state_value ? state_value[s] : 0.0;
    row.z = (double)act[e];
    row.w = (double)R[e];
    reinterpret_cast<double4*>(out)[k] = row;
}

// arrival k -> its record's place in the table: position in the sorted stream = the passes' logs composed, t = position - first
// position of its state, element e(slot, t).  rec_t may alias log0 (a thread reads log0[k] before it writes rec_t[k]).
__global__ __launch_bounds__(256) void arrival_positions_kernel(const uint32_t* log0, const uint32_t* __restrict__ log1,
                                                                const uint32_t* __restrict__ log2, const uint32_t* __restrict__ log3,
                                                                int nlogs, const int32_t* __restrict__ rec_state,
                                                                const uint32_t* __restrict__ start, const int32_t* __restrict__ state_slot,
                                                                const int64_t* __restrict__ sro, uint32_t n, int64_t* __restrict__ rec_elem,
                                                                int32_t* rec_t) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    uint32_t p = k;
    if (nlogs > 0) p = log0[p];
    if (nlogs > 1) p = log1[p];
    if (nlogs > 2) p = log2[p];
    if (nlogs > 3) p = log3[p];
    const int s = rec_state[k];
    const uint32_t t = p - start[s];
    const int slot = state_slot[s];
    rec_elem[k] = (sro[slot >> 6] + (int64_t)(t & ~3u)) * WAVE + (int64_t)(slot & 63) * 4 + (t & 3u);
    rec_t[k] = (int32_t)t;
}


// =====================================================================================================================================
// The DIRECT path (round 4): f32 tables of at most 65 536 states without arrival bookkeeping go from the (N,4) rows to the sliced
// layout with ONE write of the compact records and no global scatter pass at all.
//
//   dp_partition_kernel   reads the 32-byte rows once (same validation as ingest_compact_kernel) and writes the 8-byte {key, reward}
//                         records of every TILE of 6 656 arrivals back to the tile's own range of the record buffer, PARTITIONED by
//                         bucket (bucket = 256 consecutive state ids = the state's high byte; arrival order kept inside a bucket's
//                         run): the partition happens in LDS, the stores are sequential.  Next to it one u32 per (bucket, tile):
//                         {offset of the bucket's run in the tile, its length}.
//   dp_count_kernel       per item (bucket, group of <= 128 tiles): walks the bucket's runs of the group and counts the records of each of the
//                         bucket's 256 states — from a SIDE ARRAY of one byte per record (the state's low byte, written by the
//                         partition pass in the same order), not from the 8-byte records;
//   dp_scan_kernel        per state: exclusive scan of those counts over the groups = the arrival index t0 of the state's first
//                         record in each group; the total is the state's stream length -> slots, slice row offsets (the same small
//                         kernels as the sort path);
//   dp_pack_kernel        per item (bucket, group) again: gathers the runs into LDS, ranks them by state (the OR-mask ranks of
//                         rx_scatter_lines_kernel: stable), and writes every state's records at e(slot, t0 + j) of the sliced layout:
//                         whole quads as 16-byte (rewards) / 4-byte (actions) stores, the ragged ends of a state's piece as single
//                         elements.  A quad-row of a slice (1 KiB) is completed by the few items that are consecutive groups of
//                         the same bucket: the items are queued so that all items of a bucket run on ONE XCD one after the other
//                         (block b -> XCD b % 8), and that XCD's L2 merges the pieces into whole lines before they leave for HBM
//                         (when slot k holds state k; on length-sorted ragged tables a bucket's states lie in 256 different
//                         slices and the pieces leave as they are: DESIGN.md section 5).
//                         Count and pack are persistent kernels over per-XCD item queues (ItemQueue below).
//   dp_pad_kernel         zeroes the layout's padding (the pack kernel writes records only).
//
// HBM traffic per record: 32 + 9 (partition) + ~6 (count: 1 byte + the partly used 128-byte lines around a 26-byte run) + ~13 (the
// 208-byte runs of the records: 1.6 x 8) + 5 (pack) = 65 against 32 + 8 + 16 + 8 + 16 + 13 = 93 of the two-pass sort + pack.  Everything else (f64 storage, arrival bookkeeping, more than
// 65 536 states, the bucket layout) keeps the sort path.
constexpr int DP_TH = 512, DP_G = 13, DP_TILE = DP_TH * DP_G;     // 6 656 records per tile / chunk; two 8-wave blocks per CU, so that one
                                                                   // block's loads fly under the other's ranking (a 13 312-record tile
                                                                   // in one 16-wave block measured 14 % slower: its phases add up)
constexpr int DP_NWV = DP_TH / WAVE;
// The count / pack side works on GROUPS of DP_GT consecutive tiles; a pack block is PK_TH threads and stages up to PK_TILE records
// at a time (a group's stream of one bucket is about that long on a table that spreads its arrivals).  256 threads: four blocks
// per CU in four different phases (gather / rank / stage / write-out are separated by barriers, and 61 % of the pack's
// wave-cycles were waits with two 512-thread blocks per CU).  -DDCARL_PACK_TH=512: the two-block form (A/B builds).
#ifndef DCARL_PACK_TH
#define DCARL_PACK_TH 256
#endif
#ifndef DCARL_PARTITION_BATCHED
#define DCARL_PARTITION_BATCHED 1
#endif
constexpr int PK_TH = DCARL_PACK_TH, PK_NWV = PK_TH / WAVE, PK_TILE = PK_TH * DP_G;
constexpr int DP_GT = PK_TH / 2;                                   // tiles per group: 128 (x ~26 records of a bucket per tile = one chunk)
constexpr int DP_BS = 256;                                         // states per bucket
constexpr int DP_BSHIFT = 8;                                       // bucket = state >> 8
constexpr unsigned dp_partition_lds() { return (DP_NWV * RX_DIGITS + 16 * RX_DIGITS + 16) * 4 + DP_TILE * 8; }
struct __attribute__((aligned(16))) PkState { uint32_t so, t, c, pad; };   // a state of the bucket in one chunk: where its records start in
                                                                             // the staging buffer, its next arrival index, how many it has
constexpr unsigned dp_pack_head() { return DP_BS * 8 + DP_BS * 16 + (PK_NWV * RX_DIGITS + 16 + (DP_GT + 2) + DP_GT + 6) * 4; }
constexpr unsigned dp_pack_lds() { return dp_pack_head() + PK_TILE * 8; }
static_assert(dp_pack_head() % 16 == 0, "the staging buffer (64-bit LDS atomics, 8-byte records) stays aligned");
static_assert(PK_TH >= DP_BS && PK_TH >= DP_GT && (DP_GT & (DP_GT - 1)) == 0, "a thread per state of the bucket and per tile of the group; bisection over the runs");

// wave-local ranks of G records by an 8-bit digit (the OR-mask form of rx_scatter_lines_kernel): local[g] = records of the same digit
// before this one in the wave's part of the tile; mycnt[d] ends as the wave's count of digit d.  wmask: RX_DIGITS u64 words per wave.
template <int G, class DigitOf>
__device__ __forceinline__ void dp_rank(const uint2 (&r)[G], const bool (&ok)[G], uint32_t (&local)[G], uint32_t* mycnt,
                                        unsigned long long* wmask, int lane, DigitOf digit_of) {
    const unsigned long long lane_bit = 1ull << lane;
#pragma unroll
    for (int i = 0; i < RX_DIGITS / WAVE; ++i) { wmask[lane + i * WAVE] = 0ull; mycnt[lane + i * WAVE] = 0u; }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const uint32_t d = digit_of(r[g].x);
        unsigned long long peers = 0ull;
        uint32_t old = 0;
        if (ok[g]) __hip_atomic_fetch_or(&wmask[d], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        if (ok[g]) { peers = wmask[d]; old = mycnt[d]; }
        __builtin_amdgcn_wave_barrier();
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        if (ok[g] && below == 0) { wmask[d] = 0ull; mycnt[d] = old + (uint32_t)__popcll(peers); }
        __builtin_amdgcn_wave_barrier();
        local[g] = old + below;
    }
}

constexpr int DP_TB = 16;                                          // tiles whose table words a block collects before it writes them (64 bytes per bucket)
// SOA: the records arrive as the sampler's three arrays {state idx i32 (-1 = the visit was dropped, DS:50-51), action i32, reward
// f32} (dcarl_sample_pairs; 12 instead of 32 bytes per record) and not as (N,4) float64 rows; dropped visits simply do not enter
// the tile's partition (the tile's range of the record buffer is then only partly used; every later pass goes by the table words).
// PACKED (ABI 8): the records arrive as the 8-byte compact records themselves — {key = state << 5 | action, reward f32}, what
// convert() below makes of a row — compacted and validated on the HOST (dcarl_host_compact_rows_f32: a host-resident table then
// crosses the link as 8 instead of 32 bytes per record); the ids are checked again here (a corrupt record is filed under id 0 and
// flagged: nothing ever indexes out of range).
template <int MODE>
__global__ __launch_bounds__(DP_TH) __attribute__((amdgpu_waves_per_eu(4, 4))) void dp_partition_kernel(
    const double* __restrict__ data, const int32_t* __restrict__ p_idx, const int32_t* __restrict__ p_act, const uint32_t* __restrict__ p_rew,
    uint32_t n, int S, int A, uint32_t ntiles, uint32_t tpb, uint2* __restrict__ rec_out,
    uint8_t* __restrict__ xs_out, uint32_t* __restrict__ tab, int nb, int64_t* __restrict__ info) {
    constexpr bool SOA = MODE == 1, PACKED = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(smem);            // [NWV][RX_DIGITS]
    uint32_t* tabbuf = wcnt + DP_NWV * RX_DIGITS;                  // [DP_TB][RX_DIGITS] table words of the last tiles, not yet written
    uint32_t* wsum = tabbuf + DP_TB * RX_DIGITS;                   // [16]
    uint2* s_rec = reinterpret_cast<uint2*>(wsum + 16);            // [DP_TILE]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* mycnt = wcnt + wv * RX_DIGITS;
    unsigned long long* wmask = reinterpret_cast<unsigned long long*>(s_rec) + wv * RX_DIGITS;
    int smin = INT32_MAX, smax = INT32_MIN, amin = INT32_MAX, amax = INT32_MIN;
    uint32_t flags = 0;
    const uint32_t tile_lo = blockIdx.x * tpb, tile_hi = (ntiles - tile_lo < tpb) ? ntiles : tile_lo + tpb;
    // a row as the words the path needs: the state id (8 bytes) and {action, reward} (16 bytes); column 1 is never read (S1:73)
    struct Row { uint2 s; uint4 ar; };
    auto tile_count = [&](uint32_t tile) __attribute__((always_inline)) {
        const uint32_t base = tile * (uint32_t)DP_TILE;
        return (n - base < (uint32_t)DP_TILE) ? n - base : (uint32_t)DP_TILE;
    };
    // rows g0 .. g0+3 of the lane (a wave reads 2 KiB contiguous per row group); uniform tile pointer + a 32-bit lane offset
    auto load_rows = [&](uint32_t tile, uint32_t lane_i, int g0, Row (&q)[4]) __attribute__((always_inline)) {
        const uint4* __restrict__ rows = reinterpret_cast<const uint4*>(data) + 2 * (size_t)tile * DP_TILE;
        const size_t tb = (size_t)tile * DP_TILE;
        const uint32_t cnt = tile_count(tile);
        // UNCONDITIONAL loads (round 5): a lane beyond the tile's end re-reads the tile's last row and the caller ignores it (ok[] is
        // decided there).  Behind `if (i < cnt)` every row's loads sat in their own branch and the compiler put their wait right behind
        // them — thirteen dependent round trips to HBM per tile instead of four rows in flight at a time (DCARL_PARTITION_BATCHED=0: that form)
        const uint32_t last = cnt - 1u;                            // (a tile holds at least one record)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i0 = lane_i + (uint32_t)(g0 + u) * WAVE;
#if DCARL_PARTITION_BATCHED
            const uint32_t i = i0 < cnt ? i0 : last;
            if (g0 + u < DP_G) {
#else
            const uint32_t i = i0;
            if (g0 + u < DP_G && i < cnt) {
#endif
                if constexpr (SOA) {                               // (a wave reads 256 contiguous bytes of each array)
                    q[u].s.x = (uint32_t)p_idx[tb + i];
                    q[u].ar.x = (uint32_t)p_act[tb + i];
                    q[u].ar.z = p_rew[tb + i];
                } else if constexpr (PACKED) {                     // (a wave reads 512 contiguous bytes)
                    q[u].s = reinterpret_cast<const uint2*>(data)[tb + i];
                } else {
                    if constexpr ((DCARL_DP_NT & 1) != 0) { q[u].s = nt_load8(rows + 2u * i); q[u].ar = nt_load16(rows + 2u * i + 1u); }
                    else { q[u].s = *reinterpret_cast<const uint2*>(rows + 2u * i); q[u].ar = rows[2u * i + 1u]; }
                }
            }
        }
    };
    // (the same integer tests on the raw words as ingest_compact_kernel: see there)
    uint32_t kept = 0;                                             // SOA: records of this block's tiles that entered the partition
    auto convert = [&](const Row& q, bool& keep) __attribute__((always_inline)) {
        if constexpr (SOA) {
            const int si = (int)q.s.x, ai = (int)q.ar.x;
            keep = si != -1;                                       // DS:50-51: the sampler marks a visit outside [0, state_num) with -1
            if (!keep) return make_uint2(0u, 0u);
            smin = si < smin ? si : smin; smax = si > smax ? si : smax;
            amin = ai < amin ? ai : amin; amax = ai > amax ? ai : amax;
            const uint32_t st = (si >= 0 && si < S) ? (uint32_t)si : 0u, a = (ai >= 0 && ai < A) ? (uint32_t)ai : 0u;
            if ((q.ar.z & 0x7f800000u) == 0x7f800000u) flags |= 1u;      // NaN / Inf reward (an integer test: -fno-honor-nans)
            return make_uint2((st << ACT_BITS) | a, q.ar.z);
        }
        if constexpr (PACKED) {
            keep = true;
            const int si = (int)(q.s.x >> ACT_BITS), ai = (int)(q.s.x & ((1u << ACT_BITS) - 1u));
            smin = si < smin ? si : smin; smax = si > smax ? si : smax;
            amin = ai < amin ? ai : amin; amax = ai > amax ? ai : amax;
            const bool good = si < S && ai < A;
            if ((q.s.y & 0x7f800000u) == 0x7f800000u) flags |= 1u;
            return make_uint2(good ? q.s.x : 0u, q.s.y);
        }
        const bool s_nf = (q.s.y & 0x7ff00000u) == 0x7ff00000u, a_nf = (q.ar.y & 0x7ff00000u) == 0x7ff00000u,
                   w_nf = (q.ar.w & 0x7ff00000u) == 0x7ff00000u;
        const double sd = __hiloint2double((int)q.s.y, (int)q.s.x), ad = __hiloint2double((int)q.ar.y, (int)q.ar.x),
                     wd = __hiloint2double((int)q.ar.w, (int)q.ar.z);
        const int si = s_nf ? INT32_MIN : fabs(sd) < 2.0e9 ? (int)sd : (q.s.y >> 31) ? INT32_MIN : INT32_MAX;
        const int ai = a_nf ? INT32_MIN : fabs(ad) < 2.0e9 ? (int)ad : (q.ar.y >> 31) ? INT32_MIN : INT32_MAX;
        smin = si < smin ? si : smin; smax = si > smax ? si : smax;
        amin = ai < amin ? ai : amin; amax = ai > amax ? ai : amax;
        if (s_nf || a_nf) flags |= 2u;
        const uint32_t st = (si >= 0 && si < S) ? (uint32_t)si : 0u, a = (ai >= 0 && ai < A) ? (uint32_t)ai : 0u;
        if (w_nf || fabs(wd) > 3.4028234663852886e38) flags |= 1u;
        return make_uint2((st << ACT_BITS) | a, __float_as_uint((float)wd));
    };
    Row pre[4];                                                    // the first four rows of the NEXT tile, loaded under this tile's write-out
    uint32_t lane_i = (uint32_t)wv * (DP_G * WAVE) + lane;
    if (tile_lo < tile_hi) load_rows(tile_lo, lane_i, 0, pre);
    for (uint32_t tile = tile_lo; tile < tile_hi; ++tile) {
        const uint32_t base = tile * (uint32_t)DP_TILE;
        const uint32_t cnt = tile_count(tile);
        uint2 r[DP_G];
        bool ok[DP_G];
        asm volatile("" : "+v"(lane_i));                            // (opaque per tile: hoisted out of the tile loop the 13 row offsets
                                                                    // would be precomputed once — and spilled)
#pragma unroll
        for (int g0 = 0; g0 < DP_G; g0 += 4) {
            Row q[4];
            if (g0 == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = pre[u];
            } else {
                load_rows(tile, lane_i, g0, q);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u;
                if (g < DP_G) {
                    ok[g] = lane_i + (uint32_t)g * WAVE < cnt;
                    r[g] = make_uint2(0u, 0u);
                    if (ok[g]) { bool keep = true; r[g] = convert(q[u], keep); ok[g] = keep; }
                }
            }
        }
        uint32_t local[DP_G];
        dp_rank<DP_G>(r, ok, local, mycnt, wmask, lane, [](uint32_t k) { return (k >> (ACT_BITS + DP_BSHIFT)) & 255u; });
        __syncthreads();
        uint32_t c = 0;
        if (tid < RX_DIGITS) {
#pragma unroll
            for (int w = 0; w < DP_NWV; ++w) c += wcnt[w * RX_DIGITS + tid];
        }
        uint32_t total;
        const uint32_t so = block_excl_scan(c, wsum, &total);
        const uint32_t tslot = (tile - tile_lo) % DP_TB;
        if (tid < RX_DIGITS) {
            tabbuf[tslot * RX_DIGITS + tid] = (so << 16) | c;      // so < 6 656, c <= 6 656: 13 + 13 bits
            uint32_t at = so;
#pragma unroll
            for (int w = 0; w < DP_NWV; ++w) { const uint32_t x = wcnt[w * RX_DIGITS + tid]; wcnt[w * RX_DIGITS + tid] = at; at += x; }
            // the table words of the last (up to) 16 tiles leave together: 64 contiguous bytes per bucket (one word per tile and
            // bucket, written tile by tile, reached HBM as 5e7 partial-line writes on the configs[1] table)
            if ((tslot == DP_TB - 1 || tile + 1 == tile_hi) && tid < nb) {
                uint32_t* dst = tab + (size_t)tid * ntiles + (tile - tslot);
                for (uint32_t j = 0; j <= tslot; ++j) dst[j] = tabbuf[j * RX_DIGITS + tid];
            }
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < DP_G; ++g)
            if (ok[g]) s_rec[mycnt[(r[g].x >> (ACT_BITS + DP_BSHIFT)) & 255u] + local[g]] = r[g];
        __syncthreads();
        if (tile + 1 < tile_hi) load_rows(tile + 1, lane_i, 0, pre);
        // the tile leaves in partitioned order, four records per lane: two 16-byte stores + the records' states-in-bucket as four
        // bytes of the side array the count pass reads instead of the records (1 byte per record instead of 8)
        const uint32_t nout = SOA ? total : cnt;                   // (rows: every record of the tile is kept, total == cnt)
        if (SOA && tid == 0) kept += total;
        for (uint32_t i = 4u * tid; i < nout; i += 4u * DP_TH) {
            if (i + 4u <= nout) {
                const uint4 p0 = *reinterpret_cast<const uint4*>(s_rec + i), p1 = *reinterpret_cast<const uint4*>(s_rec + i + 2);
                const uint32_t x4 = ((p0.x >> ACT_BITS) & 255u) | (((p0.z >> ACT_BITS) & 255u) << 8) |
                                    (((p1.x >> ACT_BITS) & 255u) << 16) | (((p1.z >> ACT_BITS) & 255u) << 24);
                if constexpr ((DCARL_DP_NT & 2) != 0) {
                    nt_store16(rec_out + base + i, p0.x, p0.y, p0.z, p0.w);
                    nt_store16(rec_out + base + i + 2, p1.x, p1.y, p1.z, p1.w);
                    nt_store4(xs_out + base + i, x4);
                } else {
                    *reinterpret_cast<uint4*>(rec_out + base + i) = p0;
                    *reinterpret_cast<uint4*>(rec_out + base + i + 2) = p1;
                    *reinterpret_cast<uint32_t*>(xs_out + base + i) = x4;
                }
            } else {
                for (uint32_t k = i; k < nout; ++k) { const uint2 x = s_rec[k]; rec_out[base + k] = x; xs_out[base + k] = (uint8_t)((x.x >> ACT_BITS) & 255u); }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const int a0 = __shfl_xor(smin, off), a1 = __shfl_xor(smax, off), a2 = __shfl_xor(amin, off), a3 = __shfl_xor(amax, off);
        smin = a0 < smin ? a0 : smin; smax = a1 > smax ? a1 : smax; amin = a2 < amin ? a2 : amin; amax = a3 > amax ? a3 : amax;
        flags |= __shfl_xor(flags, off);
    }
    if (lane == 0 && tile_lo < tile_hi) {
        __hip_atomic_fetch_min(&info[I_MINSTATE], (int64_t)smin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&info[I_MAXSTATE], (int64_t)smax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_min(&info[I_MINACT], (int64_t)amin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&info[I_MAXACT], (int64_t)amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flags) __hip_atomic_fetch_or(&info[I_FLAGS], (int64_t)flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (SOA && tid == 0 && kept) __hip_atomic_fetch_add(&info[I_KEPT], (int64_t)kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block b -> (bucket, group).  XCD b % 8 (blocks are dealt to the XCDs round-robin) owns every eighth SET of DP_SET neighbouring
// buckets and walks its sets one after the other, group after group: (set, group, bucket in set).  Two kinds of locality meet in that XCD's
// L2: the pieces of a quad-row written by consecutive groups of one bucket (the 128 blocks in flight per XCD hold 32 consecutive
// groups of each bucket of the set; the pieces of a row lie within a few groups), and the 128-byte lines at the two ends of a
// bucket's run in a tile, which the neighbouring bucket's block of the same group reads too (fetched once instead of twice:
// 1.6x -> ~1.2x the bytes of the 208-byte runs).  Tables of fewer than 32 buckets (8 192 states) have too few buckets for that
// and deal (bucket, group) pairs to all XCDs instead.  Returns false for the padding blocks of the grid.
#ifndef DCARL_DP_SET
#define DCARL_DP_SET 4                                            // (A/B builds: tools/build_variant.sh ... -DDCARL_DP_SET=1)
#endif
constexpr int DP_SET = DCARL_DP_SET;
__device__ __forceinline__ bool dp_bucket_group(uint32_t b, int nb, uint32_t ngroups, int* d, uint32_t* g) {
    if (nb < 8 * DP_SET) {
        *d = (int)(b % (uint32_t)nb);
        *g = b / (uint32_t)nb;
        return *g < ngroups;
    }
    // the sets are DEALT to the XCDs (set s -> XCD s % 8), not cut into eight contiguous ranges: with state popularity falling
    // exponentially in the state id, a contiguous eighth of the buckets held 78 % of the records (that XCD did 78 % of the work)
    const uint32_t x = b & 7u, j = b >> 3;
    const uint32_t per_set = (uint32_t)DP_SET * ngroups;
    const uint32_t ls = j / per_set, within = j - ls * per_set;    // the XCD's ls-th set
    *g = within / DP_SET;
    *d = (int)((ls * 8u + ((x - ls) & 7u)) * DP_SET + within % DP_SET);   // round ls is dealt starting one XCD further on
    return *d < nb;
}
// The two (bucket, group) kernels are PERSISTENT: 8 * DP_PB_* blocks, block b on XCD b % 8, take the items j*8 + x of "their" XCD x
// from that XCD's queue (a counter), in the order above, and when it is empty the other XCDs' items.  With one block per item the
// hardware dispatcher hands out blocks in order and waits for a free slot on the XCD whose turn it is: a table whose state
// popularity falls exponentially in the state id (the heaviest set of a round holds 1.76x the round's mean) kept seven XCDs
// waiting for the eighth — pack 17.7 ms against 12.2 on the uniform table of the same size; neither dealing order changes that.
#ifndef DCARL_PK_PREFETCH
#define DCARL_PK_PREFETCH 1                                        // (0: the pack loads an item's header when it starts the item — A/B builds)
#endif
#ifndef DCARL_DP_PERSISTENT
#define DCARL_DP_PERSISTENT 1                                      // (0: one block per item, for A/B builds — tools/build_variant.sh)
#endif
constexpr int DP_QSTRIDE = 32;                                     // words between the counters (a 128-byte line each)
constexpr size_t DP_QUEUE_BYTES = 8 * DP_QSTRIDE * 4;
constexpr int DP_PB_PACK = 128, DP_PB_COUNT = 256;                 // resident blocks per XCD: 32 CUs x 4 (pack: LDS + 128 VGPRs) or x 8
struct ItemQueue {
    uint32_t* q; uint32_t per, x, ahead;
    __device__ __forceinline__ ItemQueue(uint32_t* q_, uint32_t per_) : q(q_), per(per_), x(blockIdx.x & 7u), ahead(0) {}
    // issue the request for the item after this one (the atomic's round trip hides under the item's work)
    __device__ __forceinline__ void request() {
        if (DCARL_DP_PERSISTENT) ahead = atomicAdd(&q[x * DP_QSTRIDE], 1u);
    }
    __device__ __forceinline__ uint32_t first() { if (!DCARL_DP_PERSISTENT) return blockIdx.x; request(); return take(); }
    // the requested item, or one of another XCD's, or ~0 when every queue is empty (a queue that was empty stays empty and there
    // only ever stays empty, and every block looks at all eight: no item is left behind)
    __device__ __forceinline__ uint32_t take() const {
        if (!DCARL_DP_PERSISTENT) return ~0u;
        if (ahead < per) return ahead * 8u + x;
        for (uint32_t k = 1; k < 8; ++k) {
            const uint32_t y = (x + k) & 7u;
            if (__hip_atomic_load(&q[y * DP_QSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per) continue;
            const uint32_t j = atomicAdd(&q[y * DP_QSTRIDE], 1u);
            if (j < per) return j * 8u + y;
        }
        return ~0u;
    }
};
inline uint32_t dp_grid(int nb, uint32_t ngroups) {
    if (nb < 8 * DP_SET) return (uint32_t)nb * ngroups;
    const uint32_t nsets = ((uint32_t)nb + DP_SET - 1) / DP_SET;
    return 8u * ((nsets + 7u) / 8u) * DP_SET * ngroups;
}
// tiles per group: a bucket's stream of a group should be about one pack chunk (PK_TILE records), and a tile holds 6 656 / nb of
// the bucket's records on a table that spreads its arrivals: nb / 2 tiles, at most DP_GT (the LDS tables' size), at least one
inline uint32_t dp_group_tiles(int nb) { const int gt = nb / 2; return (uint32_t)(gt < 1 ? 1 : gt > DP_GT ? DP_GT : gt); }

__global__ __launch_bounds__(256) void dp_count_kernel(const uint8_t* __restrict__ xs, const uint32_t* __restrict__ tab, uint32_t ntiles,
                                                       int nb, uint32_t ngroups, uint32_t gt, uint32_t* __restrict__ hist2,
                                                       uint32_t* __restrict__ queue, uint32_t per_xcd) {
    __shared__ uint32_t h[DP_BS];
    __shared__ uint32_t s_item;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ItemQueue iq(queue, per_xcd);
    if (tid == 0) s_item = iq.first();
    __syncthreads();
    uint32_t item = s_item;
    __syncthreads();
    while (item != ~0u) {
        bool asked = false;                                        // (block-uniform) the next item is requested
        int d; uint32_t g;
        if (dp_bucket_group(item, nb, ngroups, &d, &g)) {
            h[tid] = 0;
            __syncthreads();
            // wave wv takes tiles g*GT + wv*RPW .. +RPW-1: lane l fetches tile l's table word; one run (~26 records) per load
            // instruction, eight loads in flight before the first is counted
            constexpr int RPW = DP_GT / 4;                         // runs per wave (<= 64: one table word per lane)
            static_assert(RPW <= WAVE && RPW % 8 == 0, "a table word per lane, runs in batches of eight");
            const uint32_t tile0 = g * gt + (uint32_t)wv * RPW;   // (gt <= DP_GT tiles per group: the waves beyond them idle)
            const uint32_t mine = (lane < RPW && (uint32_t)wv * RPW + lane < gt && tile0 + lane < ntiles) ? tab[(size_t)d * ntiles + tile0 + lane] : 0u;
            for (int i0 = 0; i0 < RPW; i0 += 8) {                   // (all 32 runs in flight at once measured 7 % slower than batches of 8)
                uint32_t key[8], cnt[8], off[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t e = __shfl(mine, i0 + j);
                    cnt[j] = e & 0xffffu;
                    off[j] = e >> 16;
                    const uint8_t* src = xs + (size_t)(tile0 + i0 + j) * DP_TILE + off[j];
                    key[j] = (uint32_t)lane < cnt[j] ? src[lane] : 0u;
                }
                // the request for the next item goes out BEHIND the item's last loads (memory operations return in order: ahead of
                // them, every wait for a load would wait for the atomic's longer round trip too)
                if (i0 == RPW - 8) { if (tid == 0) iq.request(); asked = true; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if ((uint32_t)lane < cnt[j]) atomicAdd(&h[key[j]], 1u);
                    if (cnt[j] > (uint32_t)WAVE) {                  // wave-uniform: a long run (skewed arrival orders)
                        const uint8_t* src = xs + (size_t)(tile0 + i0 + j) * DP_TILE + off[j];
                        for (uint32_t k0 = WAVE + lane; k0 < cnt[j] + lane; k0 += 4 * WAVE) {       // four loads in flight
                            uint32_t kk[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) kk[u] = k0 + u * WAVE < cnt[j] ? (uint32_t)src[k0 + u * WAVE] : ~0u;
#pragma unroll
                            for (int u = 0; u < 4; ++u) if (kk[u] != ~0u) atomicAdd(&h[kk[u]], 1u);
                        }
                    }
                }
            }
            __syncthreads();
            hist2[((size_t)g * nb + d) * DP_BS + tid] = h[tid];
        }
        if (!asked && tid == 0) iq.request();
        if (tid == 0) s_item = iq.take();
        __syncthreads();                                           // (also: every thread has read h before the next item clears it)
        item = s_item;
        __syncthreads();
    }
}

// The same counts by a block per (group, 64 consecutive buckets): LANE = bucket.  The byte array of a tile is partitioned by bucket,
// so the 64 runs of the block's buckets are ONE contiguous piece of ~1.7 KB per tile: lane i walks the run of bucket d0 + i in aligned
// 4-byte words (every line of the piece is fetched once and used whole — the item-per-bucket form above pulls a 128-byte line for
// every 26-byte run and keeps 26 of a wave's 64 lanes busy) and counts into ITS 256 counters of the block's 64 KiB table in LDS; the
// group's table words are staged in LDS first (read coalesced: 512 contiguous bytes per bucket).  Runs longer than CW_LONG bytes
// (skewed arrival orders: one bucket receives most of a tile) are walked by the whole wave afterwards.  Output as above.
constexpr int CW_TH = 1024, CW_NWV = CW_TH / WAVE, CW_NB = 64, CW_TS = DP_GT + 1, CW_LONG = 64, CW_PW = 10;
constexpr unsigned dp_count_wide_lds() { return (unsigned)(CW_NB * DP_BS + CW_NB * CW_TS) * 4u; }
static_assert(CW_NB == WAVE, "lane = bucket");
__global__ __launch_bounds__(CW_TH) void dp_count_wide_kernel(const uint8_t* __restrict__ xs, const uint32_t* __restrict__ tab, uint32_t ntiles,
                                                             int nb, uint32_t gt, int nq, uint32_t* __restrict__ hist2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (measured and dropped: the counters transposed, [state in bucket][bucket = lane], so that a wave's 64 atomics land in 64 different
    // banks whatever the bytes are — 0.74 against 0.71 ms on configs[1], and slower on tables of few groups: the pass is bound by the
    // instructions around the atomics, not by their bank conflicts; nor by those instructions: counting every word WHOLE without a test per
    // byte and taking the up to three foreign bytes at either end off again afterwards measured 0.712 ms — what is left is the rate of
    // the ~40 LDS atomic instructions per (tile, wave), 26 of whose 64 x 40 lane slots hold a record)
    uint32_t* h = reinterpret_cast<uint32_t*>(smem);              // [CW_NB][DP_BS]
    uint32_t* tl = h + CW_NB * DP_BS;                              // [CW_NB][CW_TS]: table words of the group's tiles (odd stride: a column read hits 64 banks)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t g = blockIdx.x / (uint32_t)nq;
    const int d0 = (int)(blockIdx.x % (uint32_t)nq) * CW_NB;
    const int nbq = nb - d0 < CW_NB ? nb - d0 : CW_NB;
    const uint32_t tile0 = g * gt, nt = ntiles - tile0 < gt ? ntiles - tile0 : gt;
    for (int i = tid; i < nbq * (DP_BS / 4); i += CW_TH) reinterpret_cast<uint4*>(h)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < nbq * DP_GT; i += CW_TH) {
        const int r = i / DP_GT, c = i % DP_GT;
        tl[r * CW_TS + c] = (uint32_t)c < nt ? tab[(size_t)(d0 + r) * ntiles + tile0 + c] : 0u;
    }
    __syncthreads();
    uint32_t* hrow = h + lane * DP_BS;
    // the first CW_PW words of a tile's run are requested one tile AHEAD (the counting of tile t runs under the loads of tile t + CW_NWV:
    // 0.79 -> 0.71 ms for the pass on configs[1])
    auto fetch = [&](uint32_t t, uint32_t& e, uint32_t (&v)[CW_PW]) __attribute__((always_inline)) {
        e = (lane < nbq && t < nt) ? tl[lane * CW_TS + t] : 0u;
        const uint32_t cnt = e & 0xffffu, off = e >> 16;
        const uint32_t words = cnt > (uint32_t)CW_LONG ? 0u : ((off & 3u) + cnt + 3u) >> 2;
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(xs + (size_t)(tile0 + t) * DP_TILE) + (off >> 2);
#pragma unroll
        for (int u = 0; u < CW_PW; ++u) v[u] = ((uint32_t)u < words) ? src[u] : 0u;
    };
    uint32_t e_n, v_n[CW_PW];
    fetch((uint32_t)wv, e_n, v_n);
    for (uint32_t t = (uint32_t)wv; t < nt; t += CW_NWV) {
        const uint32_t e = e_n;
        uint32_t v[CW_PW];
#pragma unroll
        for (int u = 0; u < CW_PW; ++u) v[u] = v_n[u];
        fetch(t + CW_NWV, e_n, v_n);
        const uint32_t cnt = e & 0xffffu, off = e >> 16;
        const size_t tb = (size_t)(tile0 + t) * DP_TILE;           // (a multiple of 4: the words are aligned)
        const bool lng = cnt > (uint32_t)CW_LONG;
        const uint32_t lead = off & 3u;
        const uint32_t words = lng ? 0u : (lead + cnt + 3u) >> 2;  // aligned words that cover the run (cnt == 0: none)
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(xs + tb) + (off >> 2);
#pragma unroll
        for (int u = 0; u < CW_PW; ++u) {
            if ((uint32_t)u < words) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t pos = (uint32_t)u * 4u + b - lead;               // (wraps for the bytes in front of the run)
                    if (pos < cnt) atomicAdd(&hrow[(v[u] >> (8 * b)) & 255u], 1u);
                }
            }
        }
        for (uint32_t w0 = CW_PW; __any(w0 < words); w0 += 4) {                      // runs of more than ~36 bytes: the rest on demand
            uint32_t x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = (w0 + u < words) ? src[w0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (w0 + u < words) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t pos = (w0 + u) * 4u + b - lead;
                        if (pos < cnt) atomicAdd(&hrow[(x[u] >> (8 * b)) & 255u], 1u);
                    }
                }
            }
        }
        unsigned long long m = __ballot(lng);
        while (m) {                                                // wave-uniform: the long runs of this tile, one after the other, by the whole wave
            const int i = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t ee = __shfl(e, i), lc = ee & 0xffffu;
            const uint8_t* __restrict__ s8 = xs + tb + (ee >> 16);
            uint32_t* hr = h + i * DP_BS;
            for (uint32_t k0 = lane; k0 < lc + lane; k0 += 4 * WAVE) {          // four loads in flight
                uint32_t kk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kk[u] = k0 + u * WAVE < lc ? (uint32_t)s8[k0 + u * WAVE] : ~0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) if (kk[u] != ~0u) atomicAdd(&hr[kk[u]], 1u);
            }
        }
    }
    __syncthreads();
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(hist2 + ((size_t)g * nb + d0) * DP_BS);
    for (int i = tid; i < nbq * (DP_BS / 4); i += CW_TH) dst[i] = reinterpret_cast<const uint4*>(h)[i];
}

// thread = state: exclusive scan of its counts over the groups (in place) -> t0 of every (group, state); total = stream length
__global__ __launch_bounds__(256) void dp_scan_kernel(uint32_t* __restrict__ hist2, int nb, uint32_t ngroups, int S, int32_t* __restrict__ len_state) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;           // = bucket * 256 + state in bucket
    if (i >= (uint32_t)nb * DP_BS) return;
    const uint32_t d = i >> 8, x = i & 255u;
    uint32_t run = 0;
    const size_t stride = (size_t)nb * DP_BS;
    uint32_t* col = hist2 + (size_t)d * DP_BS + x;
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 8) {                  // eight loads in flight (the chain is latency, not bandwidth)
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g0 + j < ngroups ? col[(size_t)(g0 + j) * stride] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (g0 + j < ngroups) col[(size_t)(g0 + j) * stride] = run;
            run += v[j];
        }
    }
    if ((int)i < S) len_state[i] = (int32_t)run;
}

__global__ __launch_bounds__(PK_TH) __attribute__((amdgpu_waves_per_eu(4, 4))) void dp_pack_kernel(
    const uint2* __restrict__ rec, const uint32_t* __restrict__ tab, uint32_t ntiles, int nb, uint32_t ngroups, uint32_t gt,
    const uint32_t* __restrict__ t0tab, const int32_t* __restrict__ state_slot, const int64_t* __restrict__ sro, int S,
    float* __restrict__ R, uint8_t* __restrict__ act, uint32_t* __restrict__ queue, uint32_t per_xcd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t* ebase = reinterpret_cast<int64_t*>(smem);             // [BS] element of the state's record t = 0
    PkState* stx = reinterpret_cast<PkState*>(ebase + DP_BS);      // [BS]
    uint32_t* wcnt = reinterpret_cast<uint32_t*>(stx + DP_BS);     // [NWV][RX_DIGITS]
    uint32_t* wsum = wcnt + PK_NWV * RX_DIGITS;                    // [16]
    uint32_t* P = wsum + 16;                                       // [GT + 1] first record of run i in the group's stream (+ pad)
    uint32_t* roff = P + DP_GT + 2;                                // [GT] offset of run i inside its tile
    uint32_t* misc = roff + DP_GT;                                 // [6]: longest run of the item, next item, most quad rows of a state in the chunk
    uint2* s_rec = reinterpret_cast<uint2*>(misc + 6);             // [PK_TILE], 16-byte aligned
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* mycnt = wcnt + wv * RX_DIGITS;
    unsigned long long* wmask = reinterpret_cast<unsigned long long*>(s_rec) + wv * RX_DIGITS;
    ItemQueue iq(queue, per_xcd);
    if (tid == 0) misc[1] = iq.first();
    __syncthreads();
    uint32_t item = misc[1];
    __syncthreads();
    // The item's header — its table word per tile, every state's t0 and where its stream starts in the layout (two dependent loads) — is
    // three memory round trips before the first record is gathered: the NEXT item (known as soon as the queue's answer is back, behind
    // the gather's loads) has its header requested while this item is ranked and written (DCARL_PK_PREFETCH=0: A/B builds).
    bool have_pf = false;                                          // (block-uniform) pf_* hold the header of `item`
    uint32_t pf_e = 0, pf_t0 = 0;
    int pf_slot = 0;
    int64_t pf_eb = 0;
    while (item != ~0u) {
        bool asked = false;                                        // (block-uniform) the next item is requested
        bool next_known = false;                                   // (block-uniform) ... and taken: next_item, its header on the way
        uint32_t next_item = ~0u;
        int d; uint32_t g;
        if (dp_bucket_group(item, nb, ngroups, &d, &g)) {
            uint32_t c_run = 0;
            if (tid < DP_GT) {
                const uint32_t tile = g * gt + tid;
                const uint32_t e = have_pf ? pf_e : ((uint32_t)tid < gt && tile < ntiles) ? tab[(size_t)d * ntiles + tile] : 0u;
                c_run = e & 0xffffu;
                roff[tid] = e >> 16;
            }
            if (tid == 0) { misc[0] = 0; misc[2] = 0; misc[3] = 1; }
            uint32_t n_g;
            const uint32_t pre = block_excl_scan(c_run, wsum, &n_g);      // (its barriers also publish misc[0] = 0)
            if (tid < DP_GT) P[tid] = pre;
            if (tid == 0) P[DP_GT] = n_g;
            {   // the longest run decides how the runs are gathered (below)
                uint32_t m = c_run;
#pragma unroll
                for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(m, off); m = o > m ? o : m; }
                if (lane == 0 && m) atomicMax(&misc[0], m);
            }
            if (tid < DP_BS) {
                const int state = d * DP_BS + tid;
                uint32_t t0 = 0;
                int64_t eb = 0;
                if (state < S) {
                    int slot;
                    if (have_pf) { t0 = pf_t0; slot = pf_slot; eb = pf_eb; }
                    else {
                        t0 = t0tab[((size_t)g * nb + d) * DP_BS + tid];
                        slot = state_slot ? state_slot[state] : state;
                        eb = sro[slot >> 6] * WAVE + (int64_t)(slot & 63) * 4;
                    }
                    if ((slot >> 6) != (state >> 6)) misc[3] = 0;  // (after the scan's barriers; every writer stores the same value)
                }
                stx[tid] = PkState{0u, t0, 0u, 0u};
                ebase[tid] = eb;
            }
            __syncthreads();
#ifndef PK_LONG
#define PK_LONG 128
#endif
#ifndef PK_EVEN
#define PK_EVEN 8
#endif
            const bool long_runs = misc[0] > (uint32_t)PK_LONG;           // block-uniform
            const bool slices_kept = misc[3] != 0;                         // 64 neighbouring states are the 64 slots of ONE slice (block-uniform)
            const uint2* __restrict__ gsrc = rec + (size_t)g * gt * DP_TILE;      // uniform base + 32-bit lane offsets (a group spans <= 6.8 MB)
            for (uint32_t c0 = 0; c0 < n_g; c0 += PK_TILE) {
                const uint32_t cn = (n_g - c0 < (uint32_t)PK_TILE) ? n_g - c0 : (uint32_t)PK_TILE;
                // the runs (pieces of them) that fall into [c0, c0 + cn) -> dense in LDS, in stream order
                if (!long_runs) {
                    // Wave wv takes runs wv*RPW .. +RPW-1, a HALF-wave one run (~26 records): all 16 load instructions of the wave are in
                    // flight before the first LDS store (one memory round trip per chunk); what a run holds beyond 32 records (one run in
                    // ten on a table that spreads its arrivals) follows in a loop
                    constexpr int RPW = DP_GT / PK_NWV;                     // 32 runs per wave
                    const int half = lane >> 5, l5 = lane & 31;
                    uint2 v[RPW / 2];
#pragma unroll
                    for (int j = 0; j < RPW / 2; ++j) {
                        const int i = wv * RPW + 2 * j + half;
                        const uint32_t pi = P[i], ci = P[i + 1] - pi, p0 = pi - c0;     // (p0 wraps below zero for runs that began in an earlier chunk)
                        v[j] = ((uint32_t)l5 < ci && p0 + l5 < cn) ? gsrc[(uint32_t)i * DP_TILE + roff[i] + l5] : make_uint2(0u, 0u);
                    }
                    // the request for the next item goes out BEHIND the gather's loads (memory operations return in order: ahead
                    // of them, the wait for the records would wait for the atomic's longer round trip too)
                    if (!asked) { if (tid == 0) iq.request(); asked = true; }
                    bool longer = false;
#pragma unroll
                    for (int j = 0; j < RPW / 2; ++j) {
                        const int i = wv * RPW + 2 * j + half;
                        const uint32_t pi = P[i], ci = P[i + 1] - pi, p0 = pi - c0;
                        if ((uint32_t)l5 < ci && p0 + l5 < cn) s_rec[p0 + l5] = v[j];
                        longer |= ci > 32u;
                    }
                    if (__any(longer)) {
#pragma unroll 1
                        for (int j = 0; j < RPW / 2; ++j) {
                            const int i = wv * RPW + 2 * j + half;
                            const uint32_t pi = P[i], ci = P[i + 1] - pi, p0 = pi - c0;
                            const uint32_t at = (uint32_t)i * DP_TILE + roff[i];
                            for (uint32_t k = 32u + l5; k < ci; k += 32u)
                                if (p0 + k < cn) s_rec[p0 + k] = gsrc[at + k];
                        }
                    }
                } else {
                    // LONG runs (a tile sent most of its records to this bucket: skewed popularity, state-major arrival): a thread per
                    // record of the chunk finds its run by bisection over P (7 steps) — every lane loads, whatever the run lengths
                    uint32_t src[DP_G];
#pragma unroll
                    for (int g2 = 0; g2 < DP_G; ++g2) {
                        const uint32_t i = (uint32_t)tid + g2 * PK_TH, pos = c0 + i;
                        int lo = 0;
#pragma unroll
                        for (int step = DP_GT / 2; step; step >>= 1) if (P[lo + step] <= pos) lo += step;    // largest run with P[run] <= pos
                        src[g2] = (uint32_t)lo * DP_TILE + roff[lo] + (pos - P[lo]);
                    }
                    uint2 v[DP_G];
#pragma unroll
                    for (int g2 = 0; g2 < DP_G; ++g2) v[g2] = ((uint32_t)tid + g2 * PK_TH < cn) ? gsrc[src[g2]] : make_uint2(0u, 0u);
                    if (!asked) { if (tid == 0) iq.request(); asked = true; }
#pragma unroll
                    for (int g2 = 0; g2 < DP_G; ++g2) if ((uint32_t)tid + g2 * PK_TH < cn) s_rec[tid + g2 * PK_TH] = v[g2];
                }
                const bool pf_now = DCARL_PK_PREFETCH && DCARL_DP_PERSISTENT && c0 == 0 && asked;      // (block-uniform)
                if (pf_now && tid == 0) misc[4] = iq.take();              // (the queue's answer came back behind the gather's loads)
                __syncthreads();
                int d2 = 0; uint32_t g2 = 0;
                bool pf_valid = false;
                if (pf_now) {
                    next_item = misc[4];
                    next_known = true;
                    pf_valid = next_item != ~0u && dp_bucket_group(next_item, nb, ngroups, &d2, &g2);
                    if (pf_valid) {
                        if (tid < DP_GT) {
                            const uint32_t tile = g2 * gt + tid;
                            pf_e = ((uint32_t)tid < gt && tile < ntiles) ? tab[(size_t)d2 * ntiles + tile] : 0u;
                        }
                        if (tid < DP_BS && d2 * DP_BS + tid < S) {
                            pf_t0 = t0tab[((size_t)g2 * nb + d2) * DP_BS + tid];
                            pf_slot = state_slot ? state_slot[d2 * DP_BS + tid] : d2 * DP_BS + tid;
                        }
                    }
                }
                uint2 r[DP_G];
                bool ok[DP_G];
#pragma unroll
                for (int g2 = 0; g2 < DP_G; ++g2) {
                    const uint32_t i = (uint32_t)wv * (DP_G * WAVE) + g2 * WAVE + lane;
                    ok[g2] = i < cn;
                    r[g2] = ok[g2] ? s_rec[i] : make_uint2(0u, 0u);
                }
                __syncthreads();                                           // the dense copy is in registers: the buffer serves the ranks now
                uint32_t local[DP_G];
                // (ranks from eight ballots per group instead of the LDS masks — 2 LDS operations per group instead of 5, 40 VALU
                // instructions instead of 10 on vector ALUs that idle 90 % of the time — measured 6.94 against 5.93 ms: not kept)
                dp_rank<DP_G>(r, ok, local, mycnt, wmask, lane, [](uint32_t k) { return (k >> ACT_BITS) & (uint32_t)(DP_BS - 1); });
                __syncthreads();
                uint32_t c = 0;
                if (tid < DP_BS) {
#pragma unroll
                    for (int w = 0; w < PK_NWV; ++w) c += wcnt[w * RX_DIGITS + tid];
                }
                uint32_t total;
                const uint32_t so = block_excl_scan(c, wsum, &total);
                if (tid < DP_BS) {
                    uint32_t at = so;
#pragma unroll
                    for (int w = 0; w < PK_NWV; ++w) { const uint32_t x = wcnt[w * RX_DIGITS + tid]; wcnt[w * RX_DIGITS + tid] = at; at += x; }
                    stx[tid].so = so;
                    stx[tid].c = c;
                }
                {   // the most quad rows any state's piece touches in this chunk: which write-out (below)
                    uint32_t nq = 0;
                    if (tid < DP_BS && c) { const uint32_t ta = stx[tid].t; nq = ((ta + c + 3u) >> 2) - (ta >> 2); }
#pragma unroll
                    for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(nq, off); nq = o > nq ? o : nq; }
                    if (lane == 0 && nq) atomicMax(&misc[2], nq);
                }
                __syncthreads();
#pragma unroll
                for (int g2 = 0; g2 < DP_G; ++g2)
                    if (ok[g2]) s_rec[mycnt[(r[g2].x >> ACT_BITS) & (uint32_t)(DP_BS - 1)] + local[g2]] = r[g2];
                __syncthreads();
                // Write-out.  Chunks that hold about the same few records of every state (at most PK_EVEN quad rows per state: tables
                // that spread their arrivals), of a bucket whose states kept their slices, go out a thread per STATE and quad row: the
                // 64 lanes of a wave are the 64 slots of a slice, one store instruction writes a whole 1-KiB row (6.1 ms for the
                // configs[1] table; the per-record form below 7.0: its store instructions carry 16-byte pieces of a dozen rows each;
                // on length-sorted ragged tables, where neighbouring states lie in different slices, it is the other way round:
                // 15.5 against 13.2 ms).  Everything else a thread per STAGED RECORD — the same
                // work for every thread whatever the states' shares (a thread per state took 3x as long on a table with exponentially
                // distributed state popularity and 100x on one with 20 states): record i of the staging buffer belongs to state x, is
                // its (i - so)-th of this chunk and its t-th overall; the thread whose record opens a quad that lies wholly inside the
                // state's piece stores the quad (16 + 4 bytes), records of quads the piece covers only partly go out one by one.
                if (pf_valid && tid < DP_BS && d2 * DP_BS + tid < S) pf_eb = sro[pf_slot >> 6] * WAVE + (int64_t)(pf_slot & 63) * 4;
                if (pf_now) have_pf = pf_valid;                            // (of the NEXT item: read at the top of the next iteration only)
                const uint32_t nqmax = misc[2];
                if (slices_kept && nqmax <= (uint32_t)PK_EVEN) {
                    const uint32_t am = (1u << ACT_BITS) - 1u;
                    for (int x = tid; x < DP_BS; x += PK_TH) {
                        const PkState sx = stx[x];
                        const uint32_t ta = sx.t, tb = sx.t + sx.c, qa = ta >> 2;
                        const uint32_t nqq = sx.c ? ((tb + 3u) >> 2) - qa : 0u;
                        const int64_t eb = ebase[x];
                        for (uint32_t q = 0; q < nqmax; ++q) {
                            if (q >= nqq) continue;
                            const uint32_t t4 = (qa + q) << 2;      // arrival index of the quad's first record
                            const int64_t e = eb + (int64_t)t4 * WAVE;
                            if (t4 >= ta && t4 + 4u <= tb) {
                                const uint32_t i0 = sx.so + (t4 - ta);
                                const uint2 a0 = s_rec[i0], a1 = s_rec[i0 + 1], a2 = s_rec[i0 + 2], a3 = s_rec[i0 + 3];
                                *reinterpret_cast<uint4*>(R + e) = make_uint4(a0.y, a1.y, a2.y, a3.y);
                                *reinterpret_cast<uint32_t*>(act + e) = (a0.x & am) | ((a1.x & am) << 8) | ((a2.x & am) << 16) | ((a3.x & am) << 24);
                            } else {
#pragma unroll
                                for (uint32_t j = 0; j < 4u; ++j) {
                                    const uint32_t t = t4 + j;
                                    if (t >= ta && t < tb) {
                                        const uint2 a = s_rec[sx.so + (t - ta)];
                                        R[e + j] = __uint_as_float(a.y);
                                        act[e + j] = (uint8_t)(a.x & am);
                                    }
                                }
                            }
                        }
                    }
                } else
                for (uint32_t i = tid; i < cn; i += PK_TH) {
                    const uint2 a0 = s_rec[i];
                    const uint32_t x = (a0.x >> ACT_BITS) & (uint32_t)(DP_BS - 1);
                    const PkState sx = stx[x];
                    const uint32_t t = sx.t + (i - sx.so), q0 = t & ~3u;
                    const int64_t e = ebase[x] + (int64_t)q0 * WAVE;       // e(slot, q0) = (sro + q0) * 64 + lane * 4
                    const uint32_t am = (1u << ACT_BITS) - 1u;
                    if (q0 >= sx.t && q0 + 4u <= sx.t + sx.c) {
                        if ((t & 3u) == 0u) {
                            const uint2 a1 = s_rec[i + 1], a2 = s_rec[i + 2], a3 = s_rec[i + 3];
                            *reinterpret_cast<uint4*>(R + e) = make_uint4(a0.y, a1.y, a2.y, a3.y);
                            *reinterpret_cast<uint32_t*>(act + e) = (a0.x & am) | ((a1.x & am) << 8) | ((a2.x & am) << 16) | ((a3.x & am) << 24);
                        }
                    } else {
                        R[e + (t & 3u)] = __uint_as_float(a0.y);
                        act[e + (t & 3u)] = (uint8_t)(a0.x & am);
                    }
                }
                __syncthreads();
                if (tid < DP_BS) stx[tid].t += stx[tid].c;
                if (tid == 0) misc[2] = 0;
                __syncthreads();
            }
        }
        if (next_known) {
            item = next_item;                                      // (have_pf was set with it)
            __syncthreads();                                       // the item's tables in LDS are free
        } else {
            have_pf = false;
            if (!asked && tid == 0) iq.request();
            if (tid == 0) misc[1] = iq.take();
            __syncthreads();                                       // (also: the item's tables in LDS are free)
            item = misc[1];
            __syncthreads();                                       // (an item without work takes thread 0 straight to the next write)
        }
    }
}

// the layout's padding: elements [len, rows of the slice) of every slot (dp_pack_kernel writes records only).  Block (w, y): slice
// w, the y-th of gridDim.y equal pieces of its quad rows; thread = (slot of the slice, one of four row phases) — a slice of 20 long
// streams has 44 padding lanes of millions of rows each (a thread per slot took 36 ms there).
constexpr int PAD_Y = 64;
__global__ __launch_bounds__(256) void dp_pad_kernel(const int32_t* __restrict__ len_slot, const int64_t* __restrict__ sro, int S, int W,
                                                     float* __restrict__ R, uint8_t* __restrict__ act) {
    const int w = blockIdx.x, lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int k = w * WAVE + lane;
    const int64_t row0 = sro[w];
    const uint32_t rows = (uint32_t)(sro[w + 1] - row0);
    const uint32_t len = k < S ? (uint32_t)len_slot[k] : 0u;
    if (len >= rows) return;
    const int64_t eb = row0 * WAVE + (int64_t)lane * 4;
    const uint32_t nq = rows >> 2, per = (nq + gridDim.y - 1) / gridDim.y;
    const uint32_t q_lo = blockIdx.y * per, q_hi = (q_lo + per < nq) ? q_lo + per : nq;
    for (uint32_t q = q_lo + ph; q < q_hi; q += 4) {
        const uint32_t t4 = q << 2;
        if (t4 + 4u <= len) continue;                              // a quad of records
        const int64_t e = eb + (int64_t)t4 * WAVE;
        if (t4 >= len) {
            *reinterpret_cast<uint4*>(R + e) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint32_t*>(act + e) = 0u;
        } else {
            for (uint32_t t = len; t < t4 + 4u; ++t) { R[e + (t & 3u)] = 0.f; act[e + (t & 3u)] = 0; }
        }
    }
}

// ---- host side: the plan (which buffer holds what) and the launch sequences ------------------------------------------------
inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Passes { int n; int shift[8]; int bits[8]; };
// digits covering bits [lo, lo + width) in passes of at most 8 bits, appended to p
inline void add_passes(Passes& p, int lo, int width) {
    if (width <= 0) return;
    const int np = (width + 7) / 8, per = (width + np - 1) / np;
    for (int i = 0; i < np; ++i) {
        const int sh = lo + i * per, b = (sh + per <= lo + width) ? per : lo + width - sh;
        if (b > 0) { p.shift[p.n] = sh; p.bits[p.n] = b; ++p.n; }
    }
}
inline void block_split(int64_t n, uint32_t* blk, int* nblk) {
    constexpr int64_t unit = 2 * RX_TILE;                       // a multiple of the scatter kernel's tile (8 192 or 16 384)
    const int64_t tiles = (n + unit - 1) / unit;
    const int64_t tpb = tiles > RX_MAXBLK ? (tiles + RX_MAXBLK - 1) / RX_MAXBLK : 1;
    *blk = (uint32_t)(tpb * unit);
    *nblk = (int)((n + *blk - 1) / *blk);
    if (*nblk < 1) *nblk = 1;
}

struct IngestPlan {
    int64_t N; int S, A, VB; bool arrival, sort_len, buckets, pairs;
    Passes rec, len;                 // record sort, slot sort (keys = inverted lengths)
    uint32_t blk, lblk; int nblk, lnblk, W;
    int lbits;
    size_t key[2], val[2], idx[2], log[2], hist, tot, start, end1, len_state, lkey[2], lval[2], band_off, unit_slice, tile_sum, total;
};

IngestPlan make_plan(int64_t N, int S, int A, int VB, bool arrival, bool sort_len, bool buckets) {
    IngestPlan p{};
    p.N = N; p.S = S; p.A = A; p.VB = VB; p.arrival = arrival; p.buckets = buckets;
    p.W = slices_of(S);
    p.sort_len = sort_len && S > WAVE && !buckets;
    // f32 values: the records travel as {key, value} pairs through rx_scatter_lines_kernel
    // (pair buffer i = key[i] and val[i], which are adjacent).  DCARL_INGEST_PAIRS=0: the two-array passes (A/B runs, tests).
    {
        const char* e = DCARL_KNOB("DCARL_INGEST_PAIRS");
        p.pairs = VB == 4 && !(e && e[0] == '0');
    }
    p.rec.n = 0;
    if (buckets) add_passes(p.rec, 0, bits_for(A));                // (state, action): the action digit first
    add_passes(p.rec, ACT_BITS, bits_for(S));
    block_split(N, &p.blk, &p.nblk);
    p.len.n = 0;
    p.lbits = bits_for(N + 1);
    if (p.sort_len) add_passes(p.len, 0, p.lbits);
    block_split(S, &p.lblk, &p.lnblk);
    const int64_t groups = buckets ? (int64_t)S * A : S;
    const size_t n = (size_t)(N > 0 ? N : 1);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
    for (int i = 0; i < 2; ++i) { p.key[i] = take(n * 4); p.val[i] = take(n * VB); p.idx[i] = arrival ? take(n * 4) : 0; }
    // pair passes with arrival bookkeeping: one position log per pass (the first two in the idx buffers, which they replace)
    for (int i = 0; i < 2; ++i) p.log[i] = (arrival && p.pairs && !buckets && p.rec.n > 2 + i) ? take(n * 4) : 0;
    const int mb = p.nblk > p.lnblk ? p.nblk : p.lnblk;
    p.hist = take((size_t)RX_DIGITS * mb * 4);
    p.tot = take(RX_DIGITS * 4);
    p.start = take((size_t)groups * 4 + 4);
    p.end1 = take((size_t)groups * 4 + 4);
    p.len_state = take((size_t)S * 4 + 4);
    for (int i = 0; i < 2; ++i) { p.lkey[i] = take((size_t)S * 4 + 4); p.lval[i] = take((size_t)S * 4 + 4); }
    p.band_off = take((size_t)(p.W + 1) * 4);
    p.unit_slice = take((size_t)(N / PACK_ROWS + 2 * (int64_t)p.W + 2) * 4);
    p.tile_sum = take((size_t)((groups + CS_TILE - 1) / CS_TILE + 1) * 8);
    p.total = o;
    return p;
}

template <int VB, bool IDX>
void launch_scatter(const uint32_t* ki, const void* vi, const uint32_t* ii, uint32_t* ko, void* vo, uint32_t* io, uint32_t n, int shift,
                    int bits, uint32_t blk, const uint32_t* hist, int nblk, const uint32_t* tot, hipStream_t st) {
    // Two instances (blk is a multiple of every tile size): 512 threads / tiles of 8 192 is what ships; the 256-thread one (tiles
    // of 4 096, four blocks per CU) is kept for A/B runs.  What bounds the pass is the write-out: runs of ~32 records per digit
    // at random alignment (tools/experiments/ubench_scatter.hip: the same kernel writing sequentially runs at 4.6 TB/s, the real one at
    // 2.6-3.4 on random keys; without its stores 5.2; a plain copy with this grid 5.6).  Shorter tiles halve the runs (256
    // threads: 2.2-2.8 TB/s on random keys — they only won on the too regular arrival order of this repo's first end-to-end
    // bench table); a 16 384-record tile has one block per CU left and loses to its own latency; {key, value} as one 8-byte
    // record changes nothing.
    const char* force = DCARL_KNOB("DCARL_INGEST_SCATTER_THREADS");          // "256" / "512": tests and A/B runs
    if (force && atoi(force) == 256) {
        constexpr int TH = 256;
        constexpr unsigned lds = rx_scatter_lds<VB, IDX, TH>();
        DCARL_RAISE_LDS_LIMIT(((int)lds), rx_scatter_kernel<VB, IDX, TH>);
        hipLaunchKernelGGL((rx_scatter_kernel<VB, IDX, TH>), dim3(nblk), dim3(TH), lds, st, ki, vi, ii, ko, vo, io, n, shift, bits, blk,
                           hist, nblk, tot);
        return;
    }
    constexpr int TH = RX_THREADS;
    constexpr unsigned lds = rx_scatter_lds<VB, IDX, TH>();
    DCARL_RAISE_LDS_LIMIT(((int)lds), rx_scatter_kernel<VB, IDX, TH>);
    hipLaunchKernelGGL((rx_scatter_kernel<VB, IDX, TH>), dim3(nblk), dim3(TH), lds, st, ki, vi, ii, ko, vo, io, n, shift, bits, blk,
                       hist, nblk, tot);
}

// all passes of one sort; buffers ping-pong between index 0 and 1 starting at 0; hist_ready: the first pass's histogram exists.
// last_val (nullable): where the LAST pass writes its values instead of the ping-pong buffer.  Returns the index holding the result.
template <int VB, bool IDX>
int run_sort(const Passes& ps, uint32_t n, uint32_t blk, int nblk, uint32_t* const key[2], void* const val[2], uint32_t* const idx[2],
             uint32_t* hist, uint32_t* tot, bool hist_ready, void* last_val, hipStream_t st) {
    int cur = 0;
    for (int i = 0; i < ps.n; ++i) {
        if (!(i == 0 && hist_ready))
            hipLaunchKernelGGL(rx_hist_kernel, dim3(nblk), dim3(RX_THREADS), 0, st, key[cur], n, ps.shift[i], ps.bits[i], blk, hist, nblk);
        hipLaunchKernelGGL(rx_scan_kernel, dim3(1 << ps.bits[i]), dim3(256), 0, st, hist, nblk, tot);
        void* vo = (i == ps.n - 1 && last_val) ? last_val : val[cur ^ 1];
        launch_scatter<VB, IDX>(key[cur], val[cur], idx[cur], key[cur ^ 1], vo, idx[cur ^ 1], n, ps.shift[i], ps.bits[i], blk, hist, nblk,
                                tot, st);
        cur ^= 1;
    }
    return cur;
}

// the same for {key, f32 value} pairs: every pass through rx_scatter_lines_kernel (512 threads, stores in 64-byte units, tiles
// of 6 656 records; 6 144 in the last pass, whose run reports need 3 KiB more LDS).  start / end1 (nullable): the LAST pass
// reports every state's first / one-past-last position (start must hold ~0 and end1 0 on entry).
constexpr int LN_TH = 512, LN_G = 13, LN_G_LAST = 12, LN_UNIT = 8, LN_G_VAL = 8, LN_UNIT_VAL = 16;
// values (nullable, with start / end1): the last pass writes only the values, there (bucket mode: A = the action count).
template <int G, int UNIT, bool BOUNDS, bool VAL_ONLY, bool DST>
void launch_lines(int nblk, hipStream_t st, const uint2* in, uint2* out, uint32_t n, int shift, int bits, uint32_t blk, const uint32_t* hist,
                  const uint32_t* tot, uint32_t* start, uint32_t* end1, int A, uint32_t* log) {
    constexpr unsigned lds = rx_lines_lds<LN_TH, G, UNIT, BOUNDS>();
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    DCARL_RAISE_LDS_LIMIT(((int)lds), rx_scatter_lines_kernel<LN_TH, G, UNIT, BOUNDS, VAL_ONLY, DST>);
    hipLaunchKernelGGL((rx_scatter_lines_kernel<LN_TH, G, UNIT, BOUNDS, VAL_ONLY, DST>), dim3(nblk), dim3(LN_TH), lds, st, in, out, n, shift, bits,
                       blk, hist, nblk, tot, start, end1, A, log);
}
// logs (nullable): logs[i] receives pass i's position log (arrival bookkeeping).
int run_sort_pairs(const Passes& ps, uint32_t n, uint32_t blk, int nblk, uint2* const rec[2], uint32_t* hist, uint32_t* tot, bool hist_ready,
                   uint32_t* start, uint32_t* end1, float* values, int A, uint32_t* const* logs, hipStream_t st) {
    int cur = 0;
    for (int i = 0; i < ps.n; ++i) {
        if (!(i == 0 && hist_ready))
            hipLaunchKernelGGL(rx_hist_kernel, dim3(nblk), dim3(RX_THREADS), 0, st, reinterpret_cast<const uint32_t*>(rec[cur]), n, ps.shift[i],
                               ps.bits[i], blk, hist, nblk, 2);
        hipLaunchKernelGGL(rx_scan_kernel, dim3(1 << ps.bits[i]), dim3(256), 0, st, hist, nblk, tot);
        const bool last = i == ps.n - 1 && start;
        const int sh = ps.shift[i], bi = ps.bits[i];
        uint32_t* lg = logs ? logs[i] : nullptr;
        if (last && values) launch_lines<LN_G_VAL, LN_UNIT_VAL, true, true, false>(nblk, st, rec[cur], reinterpret_cast<uint2*>(values), n, sh, bi, blk, hist, tot, start, end1, A, nullptr);
        else if (last && lg) launch_lines<LN_G_LAST, LN_UNIT, true, false, true>(nblk, st, rec[cur], rec[cur ^ 1], n, sh, bi, blk, hist, tot, start, end1, A, lg);
        else if (last) launch_lines<LN_G_LAST, LN_UNIT, true, false, false>(nblk, st, rec[cur], rec[cur ^ 1], n, sh, bi, blk, hist, tot, start, end1, A, nullptr);
        else if (lg) launch_lines<LN_G, LN_UNIT, false, false, true>(nblk, st, rec[cur], rec[cur ^ 1], n, sh, bi, blk, hist, tot, nullptr, nullptr, 0, lg);
        else launch_lines<LN_G, LN_UNIT, false, false, false>(nblk, st, rec[cur], rec[cur ^ 1], n, sh, bi, blk, hist, tot, nullptr, nullptr, 0, nullptr);
        cur ^= 1;
    }
    return cur;
}

struct Bufs { uint32_t* key[2]; void* val[2]; uint32_t* idx[2]; uint32_t* lkey[2]; void* lval[2]; uint32_t* none[2]; };
Bufs bufs_of(const IngestPlan& p, void* ws) {
    unsigned char* b = static_cast<unsigned char*>(ws);
    Bufs r{};
    for (int i = 0; i < 2; ++i) {
        r.key[i] = reinterpret_cast<uint32_t*>(b + p.key[i]);
        r.val[i] = b + p.val[i];
        r.idx[i] = p.arrival ? reinterpret_cast<uint32_t*>(b + p.idx[i]) : nullptr;
        r.lkey[i] = reinterpret_cast<uint32_t*>(b + p.lkey[i]);
        r.lval[i] = b + p.lval[i];
        r.none[i] = nullptr;
    }
    return r;
}

template <typename T, bool ARR, bool PAIRS = false>
void launch_compact(const IngestPlan& p, const double* data, const Bufs& b, int32_t* rec_state, uint32_t* hist, int64_t* info, hipStream_t st) {
    const int sh = p.rec.n ? p.rec.shift[0] : 0, bi = p.rec.n ? p.rec.bits[0] : 0;
    hipLaunchKernelGGL((ingest_compact_kernel<T, ARR, PAIRS>), dim3(p.nblk), dim3(RX_THREADS), 0, st, data, (uint32_t)p.N, p.S,
                       p.A, p.blk, sh, bi, b.key[0], static_cast<T*>(b.val[0]), b.idx[0], rec_state, hist, p.nblk, info);
}

// ---- the direct path's plan -------------------------------------------------------------------------------------------------
struct DirectPlan {
    int64_t N; int S, nb, W; uint32_t ntiles, ngroups, gt, tpb; int nblk; uint32_t lblk; int lnblk, lbits; Passes len;
    size_t rec, xs, tab, hist2, len_state, state_slot, lkey[2], lval[2], band_off, hist, tot, total;
    size_t queue;                                                   // the work queues of the two persistent kernels
};
// mode (the caller's flags, the SAME in the workspace-size, group and pack calls of one table): -1 automatic = eligible tables of
// >= 2^20 records (below that the launch count, not the traffic, is what an ingest costs) and >= 2 048 states; 0 never (DCARL_INGEST_NO_DIRECT);
// 1 whenever the table is eligible (DCARL_INGEST_FORCE_DIRECT: tests run it at every size)
bool use_direct(int64_t N, int S, int VB, bool arrival, bool buckets, int mode) {
    if (VB != 4 || arrival || buckets || N <= 0 || S > 65536 || mode == 0) return false;
    return mode == 1 || (N >= ((int64_t)1 << 20) && S >= 2048);    // (fewer than 8 buckets: the sort path is 5-40 % faster, tools/experiments/ab_ingest_paths.py)
}
DirectPlan make_direct_plan(int64_t N, int S, bool sort_len) {
    DirectPlan p{};
    p.N = N; p.S = S;
    p.nb = (S + DP_BS - 1) / DP_BS;
    p.W = slices_of(S);
    p.ntiles = (uint32_t)((N + DP_TILE - 1) / DP_TILE);
    p.gt = dp_group_tiles(p.nb);
    p.ngroups = (p.ntiles + p.gt - 1) / p.gt;
    // blocks of the partition pass: whole tiles, at least 32 per block when the table allows (a block's table words of
    // consecutive tiles are neighbours in memory), at most RX_MAXBLK * 2 blocks
    uint32_t tpb = (p.ntiles + 4095) / 4096;
    if (tpb < 32) tpb = p.ntiles < 32u * 512u ? (p.ntiles + 511) / 512 : 32;
    if (tpb < 1) tpb = 1;
    p.tpb = tpb;
    p.nblk = (int)((p.ntiles + tpb - 1) / tpb);
    p.len.n = 0;
    p.lbits = bits_for(N + 1);
    if (sort_len && S > WAVE) add_passes(p.len, 0, p.lbits);
    block_split(S, &p.lblk, &p.lnblk);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
    p.rec = take((size_t)N * 8);
    p.xs = take((size_t)N + 4);
    p.tab = take((size_t)p.nb * p.ntiles * 4);
    p.hist2 = take((size_t)p.ngroups * p.nb * DP_BS * 4);
    p.len_state = take((size_t)p.nb * DP_BS * 4 + 4);
    p.state_slot = take((size_t)S * 4 + 4);
    for (int i = 0; i < 2; ++i) { p.lkey[i] = take((size_t)S * 4 + 4); p.lval[i] = take((size_t)S * 4 + 4); }
    p.band_off = take((size_t)(p.W + 1) * 4);
    p.hist = take((size_t)RX_DIGITS * p.lnblk * 4);
    p.tot = take(RX_DIGITS * 4);
    p.queue = take(2 * DP_QUEUE_BYTES);
    p.total = o;
    return p;
}

}  // namespace

int64_t ingest_workspace_bytes(int64_t N, int S, int A, int value_bytes, bool arrival, bool buckets, int direct_mode) {
    // (the flags decide the path, so the figure is the chosen path's own: the direct path keeps ONE record buffer, 9.3 bytes per
    // record against the sort's 16)
    if (use_direct(N, S, value_bytes, arrival, buckets, direct_mode)) return (int64_t)make_direct_plan(N, S, true).total;
    return (int64_t)make_plan(N, S, A, value_bytes, arrival, true, buckets).total;
}

// phase 1 of the table ingest: everything up to the slice row offsets (the caller then knows how many rows to allocate)
// the direct path up to the slice rows: partition (in tiles) -> count -> scan -> slots / slice rows; dcarl_ingest_pack writes the
// layout.  SOA: the records come as the sampler's {idx, act, R} arrays (launch_ingest_group_pairs) instead of (N,4) f64 rows.
template <int MODE>
int launch_direct_group(const double* data, const int32_t* p_idx, const int32_t* p_act, const float* p_rew, int64_t N, int S, int A,
                        bool sort_len, void* ws, int32_t* len_slot, int32_t* slot_state, int32_t* state_slot, int64_t* sro, int64_t* info,
                        hipStream_t st) {
    {
        const DirectPlan dp = make_direct_plan(N, S, sort_len);
        unsigned char* base = static_cast<unsigned char*>(ws);
        uint2* rec = reinterpret_cast<uint2*>(base + dp.rec);
        uint8_t* xs = base + dp.xs;
        uint32_t* tab = reinterpret_cast<uint32_t*>(base + dp.tab);
        uint32_t* hist2 = reinterpret_cast<uint32_t*>(base + dp.hist2);
        int32_t* len_state = reinterpret_cast<int32_t*>(base + dp.len_state);
        uint32_t* lkey[2] = {reinterpret_cast<uint32_t*>(base + dp.lkey[0]), reinterpret_cast<uint32_t*>(base + dp.lkey[1])};
        void* lval[2] = {base + dp.lval[0], base + dp.lval[1]};
        uint32_t* none[2] = {nullptr, nullptr};
        uint32_t* band_off = reinterpret_cast<uint32_t*>(base + dp.band_off);
        uint32_t* hist = reinterpret_cast<uint32_t*>(base + dp.hist);
        uint32_t* tot = reinterpret_cast<uint32_t*>(base + dp.tot);
        hipLaunchKernelGGL(ingest_init_info_kernel, dim3(1), dim3(64), 0, st, info, N);
        constexpr unsigned lds = dp_partition_lds();
        DCARL_RAISE_LDS_LIMIT(((int)lds), dp_partition_kernel<MODE>);
        hipLaunchKernelGGL(dp_partition_kernel<MODE>, dim3(dp.nblk), dim3(DP_TH), lds, st, data, p_idx, p_act,
                           reinterpret_cast<const uint32_t*>(p_rew), (uint32_t)N, S, A, dp.ntiles, dp.tpb, rec, xs, tab, dp.nb, info);
        {
            uint32_t* queue = reinterpret_cast<uint32_t*>(base + dp.queue);
            const uint32_t items = dp_grid(dp.nb, dp.ngroups), per_xcd = (items + 7u) / 8u;
            const uint32_t pb = (!DCARL_DP_PERSISTENT || per_xcd < (uint32_t)DP_PB_COUNT) ? per_xcd : (uint32_t)DP_PB_COUNT;
            // tables of >= 32 buckets: a block per (group, 64 buckets) reads the byte array in whole contiguous pieces (1.5 -> ? ms on
            // configs[1]); fewer buckets leave most of its lanes (= buckets) idle.  DCARL_DP_COUNT=queue / wide: A/B runs.
            bool wide = dp.nb >= 32;
            if (const char* e = DCARL_KNOB("DCARL_DP_COUNT")) wide = e[0] == 'w' ? true : e[0] == 'q' ? false : wide;
            if (wide) {
                constexpr unsigned cl = dp_count_wide_lds();
                DCARL_RAISE_LDS_LIMIT(((int)cl), dp_count_wide_kernel);
                const int nq = (dp.nb + CW_NB - 1) / CW_NB;
                hipLaunchKernelGGL(dp_count_wide_kernel, dim3(dp.ngroups * (uint32_t)nq), dim3(CW_TH), cl, st, xs, tab, dp.ntiles, dp.nb, dp.gt, nq, hist2);
            } else {
                (void)hipMemsetAsync(queue, 0, DP_QUEUE_BYTES, st);
                hipLaunchKernelGGL(dp_count_kernel, dim3(8u * pb), dim3(256), 0, st, xs, tab, dp.ntiles, dp.nb, dp.ngroups, dp.gt, hist2, queue, per_xcd);
            }
        }
        hipLaunchKernelGGL(dp_scan_kernel, dim3((unsigned)dp.nb), dim3(256), 0, st, hist2, dp.nb, dp.ngroups, S, len_state);
        const unsigned sb = (unsigned)ceil_div64(S, 256);
        const uint32_t lmask = dp.lbits >= 32 ? 0xffffffffu : ((1u << dp.lbits) - 1u);
        const bool sorted = dp.len.n > 0;
        hipLaunchKernelGGL(lengths_given_kernel, dim3(sb), dim3(256), 0, st, len_state, S, lmask, sorted ? lkey[0] : nullptr,
                           sorted ? static_cast<uint32_t*>(lval[0]) : nullptr, info);
        const uint32_t* order = nullptr;
        if (sorted) {
            const int lc = run_sort<4, false>(dp.len, (uint32_t)S, dp.lblk, dp.lnblk, lkey, lval, none, hist, tot, false, nullptr, st);
            order = static_cast<const uint32_t*>(lval[lc]);
        }
        hipLaunchKernelGGL(slots_kernel, dim3(sb), dim3(256), 0, st, order, len_state, S, len_slot, slot_state, state_slot);
        (void)hipMemcpyAsync(base + dp.state_slot, state_slot, (size_t)S * 4, hipMemcpyDeviceToDevice, st);   // the pack call needs it
        hipLaunchKernelGGL(slice_rows_kernel, dim3((unsigned)((dp.W + 3) / 4)), dim3(256), 0, st, len_slot, S, dp.W, sro, band_off);
        hipLaunchKernelGGL(slice_scan_kernel, dim3(1), dim3(1024), 0, st, sro, band_off, dp.W, info);
        return 0;
    }
}
// {idx, act, R} arrays -> the direct path (the caller has checked that the table is eligible: use_direct with mode 1)
int launch_ingest_group_pairs(const int32_t* idx, const int32_t* act, const float* R, int64_t N, int S, int A, bool sort_len, void* ws,
                              int32_t* len_slot, int32_t* slot_state, int32_t* state_slot, int64_t* sro, int64_t* info, hipStream_t st) {
    return launch_direct_group<1>(nullptr, idx, act, R, N, S, A, sort_len, ws, len_slot, slot_state, state_slot, sro, info, st);
}
// host-compacted 8-byte records {state << 5 | action, reward f32} -> the direct path (dcarl_ingest_group_packed_f32)
int launch_ingest_group_packed(const uint64_t* rec, int64_t N, int S, int A, bool sort_len, void* ws, int32_t* len_slot, int32_t* slot_state,
                               int32_t* state_slot, int64_t* sro, int64_t* info, hipStream_t st) {
    return launch_direct_group<2>(reinterpret_cast<const double*>(rec), nullptr, nullptr, nullptr, N, S, A, sort_len, ws, len_slot, slot_state,
                                  state_slot, sro, info, st);
}

template <typename T>
int launch_ingest_group(const double* data, int64_t N, int S, int A, bool sort_len, bool arrival, void* ws, int32_t* len_slot,
                        int32_t* slot_state, int32_t* state_slot, int64_t* sro, int32_t* rec_state, int64_t* info, hipStream_t st,
                        int direct_mode) {
    constexpr int VB = sizeof(T);
    if (use_direct(N, S, VB, arrival, false, direct_mode))
        return launch_direct_group<0>(data, nullptr, nullptr, nullptr, N, S, A, sort_len, ws, len_slot, slot_state, state_slot, sro, info, st);
    const IngestPlan p = make_plan(N, S, A, VB, arrival, sort_len, false);
    const Bufs b = bufs_of(p, ws);
    unsigned char* base = static_cast<unsigned char*>(ws);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + p.hist);
    uint32_t* tot = reinterpret_cast<uint32_t*>(base + p.tot);
    uint32_t* start = reinterpret_cast<uint32_t*>(base + p.start);
    uint32_t* end1 = reinterpret_cast<uint32_t*>(base + p.end1);
    int32_t* len_state = reinterpret_cast<int32_t*>(base + p.len_state);
    uint32_t* band_off = reinterpret_cast<uint32_t*>(base + p.band_off);
    hipLaunchKernelGGL(ingest_init_info_kernel, dim3(1), dim3(64), 0, st, info, N);
    (void)hipMemsetAsync(start, 0, (size_t)S * 4 + 4, st);
    (void)hipMemsetAsync(end1, 0, (size_t)S * 4 + 4, st);
    int cur = 0;
    if (N > 0) {
        bool done = false;
        if constexpr (VB == 4) {
            if (p.pairs) {
                uint2* const rec[2] = {reinterpret_cast<uint2*>(b.key[0]), reinterpret_cast<uint2*>(b.key[1])};
                if (arrival) launch_compact<T, true, true>(p, data, b, rec_state, hist, info, st);
                else launch_compact<T, false, true>(p, data, b, rec_state, hist, info, st);
                if (p.rec.n > 0) {                                 // the last pass reports the runs itself (start: minima from ~0)
                    (void)hipMemsetAsync(start, 0xff, (size_t)S * 4 + 4, st);
                    uint32_t* const logs[4] = {b.idx[0], b.idx[1], reinterpret_cast<uint32_t*>(base + p.log[0]), reinterpret_cast<uint32_t*>(base + p.log[1])};
                    cur = run_sort_pairs(p.rec, (uint32_t)N, p.blk, p.nblk, rec, hist, tot, true, start, end1, nullptr, 0, arrival ? logs : nullptr, st);
                } else {
                    hipLaunchKernelGGL(run_bounds_kernel<true>, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, b.key[cur], (uint32_t)N, 0, start, end1);
                }
                done = true;
            }
        }
        if (!done) {
            if (arrival) launch_compact<T, true>(p, data, b, rec_state, hist, info, st);
            else launch_compact<T, false>(p, data, b, rec_state, hist, info, st);
            cur = arrival ? run_sort<VB, true>(p.rec, (uint32_t)N, p.blk, p.nblk, b.key, b.val, b.idx, hist, tot, true, nullptr, st)
                          : run_sort<VB, false>(p.rec, (uint32_t)N, p.blk, p.nblk, b.key, b.val, b.idx, hist, tot, true, nullptr, st);
            hipLaunchKernelGGL(run_bounds_kernel<false>, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, b.key[cur], (uint32_t)N, 0, start, end1);
        }
    }
    (void)cur;
    const unsigned sb = (unsigned)ceil_div64(S, 256);
    const uint32_t lmask = p.lbits >= 32 ? 0xffffffffu : ((1u << p.lbits) - 1u);
    hipLaunchKernelGGL(lengths_kernel, dim3(sb), dim3(256), 0, st, start, end1, S, len_state, lmask, p.sort_len ? b.lkey[0] : nullptr,
                       p.sort_len ? static_cast<uint32_t*>(b.lval[0]) : nullptr, info);
    const uint32_t* order = nullptr;
    if (p.sort_len) {
        const int lc = run_sort<4, false>(p.len, (uint32_t)S, p.lblk, p.lnblk, b.lkey, b.lval, b.none, hist, tot, false, nullptr, st);
        order = static_cast<const uint32_t*>(b.lval[lc]);
    }
    hipLaunchKernelGGL(slots_kernel, dim3(sb), dim3(256), 0, st, order, len_state, S, len_slot, slot_state, state_slot);
    hipLaunchKernelGGL(slice_rows_kernel, dim3((unsigned)((p.W + 3) / 4)), dim3(256), 0, st, len_slot, S, p.W, sro, band_off);
    hipLaunchKernelGGL(slice_scan_kernel, dim3(1), dim3(1024), 0, st, sro, band_off, p.W, info);
    if constexpr (VB == 4) {
        if (p.pairs && arrival && N > 0) {
            // arrival bookkeeping of the pair passes: rec_t over the first log (in place), rec_elem in the pair buffer the sort
            // left free; launch_ingest_pack copies them out
            const int c = p.rec.n & 1;
            hipLaunchKernelGGL(arrival_positions_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, b.idx[0], b.idx[1],
                               reinterpret_cast<const uint32_t*>(base + p.log[0]), reinterpret_cast<const uint32_t*>(base + p.log[1]), p.rec.n,
                               rec_state, start, state_slot, sro, (uint32_t)N, reinterpret_cast<int64_t*>(b.key[c ^ 1]),
                               reinterpret_cast<int32_t*>(b.idx[0]));
        }
    }
    return 0;
}

// The slot order alone, for tables built some other way (sampler, state-major arrays): states by descending stream length
// (stable) -> len per slot, slot <-> state maps, slice row offsets.  max_len bounds the lengths (it sizes the sort keys).
int64_t slot_order_workspace_bytes(int S) {
    uint32_t blk; int nblk;
    block_split(S, &blk, &nblk);
    return (int64_t)(align_up((size_t)RX_DIGITS * nblk * 4) + align_up(RX_DIGITS * 4) + 4 * align_up((size_t)S * 4 + 4) +
                     align_up((size_t)(slices_of(S) + 1) * 4) + 256);
}
int launch_slot_order(const int32_t* len_state, int S, int64_t max_len, bool sort_len, void* ws, int32_t* len_slot, int32_t* slot_state,
                      int32_t* state_slot, int64_t* sro, int64_t* info, hipStream_t st) {
    unsigned char* base = static_cast<unsigned char*>(ws);
    uint32_t blk; int nblk;
    block_split(S, &blk, &nblk);
    size_t o = 0;
    auto take = [&](size_t bytes) { unsigned char* p = base + o; o += align_up(bytes); return p; };
    uint32_t* hist = reinterpret_cast<uint32_t*>(take((size_t)RX_DIGITS * nblk * 4));
    uint32_t* tot = reinterpret_cast<uint32_t*>(take(RX_DIGITS * 4));
    uint32_t* lkey[2]; void* lval[2]; uint32_t* none[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) { lkey[i] = reinterpret_cast<uint32_t*>(take((size_t)S * 4 + 4)); lval[i] = take((size_t)S * 4 + 4); }
    uint32_t* band_off = reinterpret_cast<uint32_t*>(take((size_t)(slices_of(S) + 1) * 4));
    const int W = slices_of(S);
    const bool sorted = sort_len && S > WAVE;
    const int lbits = bits_for(max_len + 1);
    const uint32_t lmask = lbits >= 32 ? 0xffffffffu : ((1u << lbits) - 1u);
    Passes ps{};
    if (sorted) add_passes(ps, 0, lbits);
    hipLaunchKernelGGL(ingest_init_info_kernel, dim3(1), dim3(64), 0, st, info, (int64_t)0);
    const unsigned sb = (unsigned)ceil_div64(S, 256);
    hipLaunchKernelGGL(lengths_given_kernel, dim3(sb), dim3(256), 0, st, len_state, S, lmask, sorted ? lkey[0] : nullptr,
                       sorted ? static_cast<uint32_t*>(lval[0]) : nullptr, info);
    const uint32_t* order = nullptr;
    if (sorted) {
        const int lc = run_sort<4, false>(ps, (uint32_t)S, blk, nblk, lkey, lval, none, hist, tot, false, nullptr, st);
        order = static_cast<const uint32_t*>(lval[lc]);
    }
    hipLaunchKernelGGL(slots_kernel, dim3(sb), dim3(256), 0, st, order, len_state, S, len_slot, slot_state, state_slot);
    hipLaunchKernelGGL(slice_rows_kernel, dim3((unsigned)((W + 3) / 4)), dim3(256), 0, st, len_slot, S, W, sro, band_off);
    hipLaunchKernelGGL(slice_scan_kernel, dim3(1), dim3(1024), 0, st, sro, band_off, W, info);
    return 0;
}

// phase 2: the sorted stream (still in the workspace) -> R / act in the sliced layout (+ arrival bookkeeping)
template <typename T>
int launch_ingest_pack(int64_t N, int S, int A, bool sort_len, bool arrival, const void* ws, const int32_t* len_slot,
                       const int32_t* slot_state, const int64_t* sro, int64_t total_bands, T* R, uint8_t* act, int64_t* rec_elem,
                       int32_t* rec_t, hipStream_t st, int direct_mode) {
    constexpr int VB = sizeof(T);
    if constexpr (VB == 4) {
        if (use_direct(N, S, VB, arrival, false, direct_mode)) {
            if (total_bands <= 0) return 0;
            const DirectPlan dp = make_direct_plan(N, S, sort_len);
            const unsigned char* base = static_cast<const unsigned char*>(ws);
            const uint2* rec = reinterpret_cast<const uint2*>(base + dp.rec);
            const uint32_t* tab = reinterpret_cast<const uint32_t*>(base + dp.tab);
            const uint32_t* t0tab = reinterpret_cast<const uint32_t*>(base + dp.hist2);
            const int32_t* state_slot = slot_state ? reinterpret_cast<const int32_t*>(base + dp.state_slot) : nullptr;
            hipLaunchKernelGGL(dp_pad_kernel, dim3((unsigned)dp.W, PAD_Y), dim3(256), 0, st, len_slot, sro, S, dp.W, R, act);
            constexpr unsigned lds = dp_pack_lds();
            static_assert(lds <= (PK_TH == 256 ? 40 : 80) * 1024, "four (two) blocks per CU");
            DCARL_RAISE_LDS_LIMIT(((int)lds), dp_pack_kernel);
            uint32_t* queue = reinterpret_cast<uint32_t*>(const_cast<unsigned char*>(base) + dp.queue + DP_QUEUE_BYTES);
            const uint32_t items = dp_grid(dp.nb, dp.ngroups), per_xcd = (items + 7u) / 8u;
            const uint32_t pb = (!DCARL_DP_PERSISTENT || per_xcd < (uint32_t)DP_PB_PACK) ? per_xcd : (uint32_t)DP_PB_PACK;
            (void)hipMemsetAsync(queue, 0, DP_QUEUE_BYTES, st);
            hipLaunchKernelGGL(dp_pack_kernel, dim3(8u * pb), dim3(PK_TH), lds, st, rec, tab, dp.ntiles, dp.nb, dp.ngroups, dp.gt,
                               t0tab, state_slot, sro, S, R, act, queue, per_xcd);
            return 0;
        }
    }
    const IngestPlan p = make_plan(N, S, A, VB, arrival, sort_len, false);
    const Bufs b = bufs_of(p, const_cast<void*>(ws));
    unsigned char* base = static_cast<unsigned char*>(const_cast<void*>(ws));
    const uint32_t* start = reinterpret_cast<const uint32_t*>(base + p.start);
    const uint32_t* band_off = reinterpret_cast<const uint32_t*>(base + p.band_off);
    uint32_t* unit_slice = reinterpret_cast<uint32_t*>(base + p.unit_slice);
    if (total_bands <= 0) return 0;
    const int cur = (N > 0 ? p.rec.n : 0) & 1;
    const uint32_t units = (uint32_t)total_bands;
    hipLaunchKernelGGL(unit_slice_kernel, dim3((units + 255) / 256), dim3(256), 0, st, band_off, p.W, units, unit_slice);
    constexpr unsigned lds = pack_lds<VB>();
    constexpr int WPB = PACK_WAVES;
    const dim3 grid((units + WPB - 1) / WPB), block(WPB * WAVE);
    if constexpr (VB == 4) {
        if (p.pairs) {
            DCARL_RAISE_LDS_LIMIT(((int)lds), ingest_pack_kernel<VB, false, true>);
            hipLaunchKernelGGL((ingest_pack_kernel<VB, false, true>), grid, block, lds, st, b.key[cur], b.val[cur], b.idx[cur], start, len_slot,
                               slot_state, sro, band_off, unit_slice, units, S, R, act, rec_elem, rec_t);
            if (arrival && rec_elem && rec_t) {                    // made by the group call (arrival_positions_kernel)
                (void)hipMemcpyAsync(rec_elem, b.key[cur ^ 1], (size_t)N * 8, hipMemcpyDeviceToDevice, st);
                (void)hipMemcpyAsync(rec_t, b.idx[0], (size_t)N * 4, hipMemcpyDeviceToDevice, st);
            }
            return 0;
        }
    }
    if (arrival && rec_elem && rec_t) {
        DCARL_RAISE_LDS_LIMIT(((int)lds), ingest_pack_kernel<VB, true>);
        hipLaunchKernelGGL((ingest_pack_kernel<VB, true>), grid, block, lds, st, b.key[cur], b.val[cur], b.idx[cur], start, len_slot,
                           slot_state, sro, band_off, unit_slice, units, S, R, act, rec_elem, rec_t);
    } else {
        DCARL_RAISE_LDS_LIMIT(((int)lds), ingest_pack_kernel<VB, false>);
        hipLaunchKernelGGL((ingest_pack_kernel<VB, false>), grid, block, lds, st, b.key[cur], b.val[cur], b.idx[cur], start, len_slot,
                           slot_state, sro, band_off, unit_slice, units, S, R, act, rec_elem, rec_t);
    }
    return 0;
}

// the final-state layout straight from the arrival-ordered table: values sorted by (state, action), arrival order inside a
// bucket, and seg_off [S*A+1]
template <typename T>
int launch_ingest_buckets(const double* data, int64_t N, int S, int A, void* ws, T* values, int64_t* seg_off, int64_t* info,
                          hipStream_t st) {
    constexpr int VB = sizeof(T);
    const IngestPlan p = make_plan(N, S, A, VB, false, false, true);
    const Bufs b = bufs_of(p, ws);
    unsigned char* base = static_cast<unsigned char*>(ws);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + p.hist);
    uint32_t* tot = reinterpret_cast<uint32_t*>(base + p.tot);
    uint32_t* start = reinterpret_cast<uint32_t*>(base + p.start);
    uint32_t* end1 = reinterpret_cast<uint32_t*>(base + p.end1);
    int64_t* tile_sum = reinterpret_cast<int64_t*>(base + p.tile_sum);
    const int64_t M = (int64_t)S * A;
    hipLaunchKernelGGL(ingest_init_info_kernel, dim3(1), dim3(64), 0, st, info, N);
    (void)hipMemsetAsync(start, 0, (size_t)M * 4 + 4, st);
    (void)hipMemsetAsync(end1, 0, (size_t)M * 4 + 4, st);
    if (N > 0) {
        bool done = false;
        if constexpr (VB == 4) {
            if (p.pairs && p.rec.n > 0) {                          // (S = A = 1 has no pass to report the runs: the two-array path)
                uint2* const rec[2] = {reinterpret_cast<uint2*>(b.key[0]), reinterpret_cast<uint2*>(b.key[1])};
                launch_compact<T, false, true>(p, data, b, nullptr, hist, info, st);
                (void)hipMemsetAsync(start, 0xff, (size_t)M * 4 + 4, st);
                run_sort_pairs(p.rec, (uint32_t)N, p.blk, p.nblk, rec, hist, tot, true, start, end1, values, A, nullptr, st);
                done = true;
            }
        }
        if (!done) {
            launch_compact<T, false>(p, data, b, nullptr, hist, info, st);
            const int cur = run_sort<VB, false>(p.rec, (uint32_t)N, p.blk, p.nblk, b.key, b.val, b.idx, hist, tot, true, values, st);
            if (p.rec.n == 0) (void)hipMemcpyAsync(values, b.val[0], (size_t)N * VB, hipMemcpyDeviceToDevice, st);
            hipLaunchKernelGGL(run_bounds_kernel<false>, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, st, b.key[cur], (uint32_t)N, A, start, end1);
        }
    }
    const int64_t ntiles = (M + CS_TILE - 1) / CS_TILE;
    hipLaunchKernelGGL(counts_tile_kernel, dim3((unsigned)ntiles), dim3(CS_THREADS), 0, st, start, end1, M, tile_sum);
    hipLaunchKernelGGL(tile_sums_scan_kernel, dim3(1), dim3(1024), 0, st, tile_sum, ntiles);
    hipLaunchKernelGGL(counts_offsets_kernel, dim3((unsigned)ntiles), dim3(CS_THREADS), 0, st, start, end1, M, tile_sum, seg_off);
    return 0;
}

template <typename T>
int launch_export_records(const T* R, const uint8_t* act, const int64_t* sro, const int32_t* state_slot, const double* state_value, int S,
                          int64_t mult, const int32_t* rec_state, const int64_t* rec_elem, int64_t N, double* out, hipStream_t st) {
    if (N == 0) return 0;
    hipLaunchKernelGGL((export_records_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, R, act, sro, state_slot,
                       state_value, S, mult, rec_state, rec_elem, N, out);
    return 0;
}
template int launch_export_records<float>(const float*, const uint8_t*, const int64_t*, const int32_t*, const double*, int, int64_t,
                                          const int32_t*, const int64_t*, int64_t, double*, hipStream_t);
template int launch_export_records<double>(const double*, const uint8_t*, const int64_t*, const int32_t*, const double*, int, int64_t,
                                           const int32_t*, const int64_t*, int64_t, double*, hipStream_t);

template int launch_ingest_group<float>(const double*, int64_t, int, int, bool, bool, void*, int32_t*, int32_t*, int32_t*, int64_t*, int32_t*,
                                        int64_t*, hipStream_t, int);
template int launch_ingest_group<double>(const double*, int64_t, int, int, bool, bool, void*, int32_t*, int32_t*, int32_t*, int64_t*, int32_t*,
                                         int64_t*, hipStream_t, int);
template int launch_ingest_pack<float>(int64_t, int, int, bool, bool, const void*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                       float*, uint8_t*, int64_t*, int32_t*, hipStream_t, int);
template int launch_ingest_pack<double>(int64_t, int, int, bool, bool, const void*, const int32_t*, const int32_t*, const int64_t*, int64_t,
                                        double*, uint8_t*, int64_t*, int32_t*, hipStream_t, int);
template int launch_ingest_buckets<float>(const double*, int64_t, int, int, void*, float*, int64_t*, int64_t*, hipStream_t);
template int launch_ingest_buckets<double>(const double*, int64_t, int, int, void*, double*, int64_t*, int64_t*, hipStream_t);

}  // namespace dcarl
