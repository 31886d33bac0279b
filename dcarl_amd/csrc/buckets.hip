// The reference's buckets themselves: `data_state_act[idx][act].append(R)` (S1:80).  The online kernels never
// materialise them (they keep sufficient statistics); the final-state kernels (bounds.hip) stream them.  This file
// turns an online record table (sliced time-major layout, include/dcarl.h) into the (state, action) bucket layout:
//   count_records_kernel  : n[s][a] = len(data_state_act[s][a]) after the whole table
//   group_records_kernel  : values[seg_off[s*A+a] + k] = k-th reward appended to bucket (s,a) (arrival order kept)
// and draws samples straight into the bucket layout (sample_buckets_kernel: add_an_act_data, DS:5-9, n times per
// bucket).  Mapping: lane = state, wavefront = slice, the lane walks its state's quads exactly like the online
// kernels; per-lane per-action cursors live in LDS [action][lane] (the bank depends on the lane only: conflict-free).
#include "common.h"
#include <type_traits>
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


// ---- group_records, the write-combining form (round 5) -------------------------------------------------------------------------
// The scatter above writes one 4-byte element per lane and record: 64 lanes = 64 different lines per store instruction, S*A output
// streams in all (720 896 on configs[1]: 46 MB of lines in progress against 32 MB of L2), so the lines leave for memory partly
// written, many times over (5 % of the roofline: 29 ms for the 1.3e9 records of configs[1]).  Here a wavefront still owns a slice and a
// lane its state's stream, in order — but a record goes into a RING of one 64-byte line per (lane, action) in LDS, and a line leaves as
// four 16-byte stores when its last element arrives: every global write is a whole aligned 64-byte line, except the head of a
// bucket that starts inside a line and the tails at the end of the stream (element by element: neighbouring buckets share those
// lines).  The position of a record inside its bucket is one returning LDS add on the (lane, action) cursor — atomics of one lane
// execute in program order, so arrival order is kept (S1:80's append).  LDS per wavefront: A x (4 096 + 512) bytes.
template <typename T, int LINE> constexpr unsigned regroup_lds(int A) { return (unsigned)A * (WAVE * LINE + 2 * WAVE * 4); }

typedef float ld_f4 __attribute__((ext_vector_type(4)));
// the record loads of one bank as inline asm (see regroup_kernel: their waits are placed by hand)
template <typename T, int PF, class Q4>
__device__ __forceinline__ void rg_load(ld_f4 (&rlo)[PF], ld_f4 (&rhi)[PF], unsigned (&ab)[PF], const Q4* Rq, const uchar4* Aq, int q0, int nq) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int q = q0 + i < nq ? q0 + i : nq - 1;            // (beyond the slice: a valid address, the step is skipped)
        const Q4* pr = Rq + (int64_t)q * WAVE;
        const uchar4* pa = Aq + (int64_t)q * WAVE;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(rlo[i]) : "v"(pr) : "memory");
        if constexpr (sizeof(T) == 8) asm volatile("global_load_dwordx4 %0, %1, off offset:16 nt" : "=v"(rhi[i]) : "v"(pr) : "memory");
        asm volatile("global_load_dword %0, %1, off nt" : "=v"(ab[i]) : "v"(pa) : "memory");
    }
}
// "at most N vector-memory operations outstanding"; every register of the bank passes through the wait, so that no use of it can be
// scheduled ahead of it
template <typename T, int N>
__device__ __forceinline__ void rg_arrived(ld_f4 (&rlo)[8], ld_f4 (&)[8], unsigned (&ab)[8]) {
    static_assert(sizeof(T) == 4, "eight quads per bank: f32");
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(rlo[0]), "+v"(rlo[1]), "+v"(rlo[2]), "+v"(rlo[3]), "+v"(rlo[4]), "+v"(rlo[5]), "+v"(rlo[6]), "+v"(rlo[7]),
                   "+v"(ab[0]), "+v"(ab[1]), "+v"(ab[2]), "+v"(ab[3]), "+v"(ab[4]), "+v"(ab[5]), "+v"(ab[6]), "+v"(ab[7])
                 : "n"(N) : "memory");
}
template <typename T, int N>
__device__ __forceinline__ void rg_arrived(ld_f4 (&rlo)[4], ld_f4 (&rhi)[4], unsigned (&ab)[4]) {
    asm volatile("s_waitcnt vmcnt(%12)"
                 : "+v"(rlo[0]), "+v"(rlo[1]), "+v"(rlo[2]), "+v"(rlo[3]), "+v"(rhi[0]), "+v"(rhi[1]), "+v"(rhi[2]), "+v"(rhi[3]),
                   "+v"(ab[0]), "+v"(ab[1]), "+v"(ab[2]), "+v"(ab[3])
                 : "n"(N) : "memory");
}

// LINE: bytes of the line in progress per (lane, action) — 64 (a whole HBM line; LDS lets three wavefronts share a CU at 11 actions)
// or 32 (twice the wavefronts per CU, the L2 has to merge two halves).  NQ: quads per step (1 or 2: the cursor adds of 4 * NQ records
// of the lane travel together).
template <typename T, int LINE, int NQ>
__global__ __launch_bounds__(WAVE) void regroup_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, int S, int A, const int64_t* __restrict__ seg_off,
    T* __restrict__ values) {
    using Q4 = typename std::conditional<sizeof(T) == 4, float4, double4>::type;
    constexpr int E = LINE / (int)sizeof(T);                     // elements per line
    constexpr int NV = LINE / 16;                                // 16-byte stores per line
    constexpr int NR = 4 * NQ;                                   // records per step
    constexpr int PF = (sizeof(T) == 4 ? 8 : 4);                 // quads per batch (two batches live: one consumed, one in flight)
    static_assert(PF % NQ == 0, "whole steps per batch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* ring = reinterpret_cast<T*>(smem);                        // [A][WAVE][E]: the line in progress of (action, lane)
    uint32_t* pos = reinterpret_cast<uint32_t*>(smem + (size_t)A * WAVE * LINE);   // [A][WAVE]: next element of the bucket, relative to
    uint32_t* start = pos + A * WAVE;                            // [A][WAVE]: ... and its first one             the state's aligned base
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const int s = w * WAVE + lane;
    const int n = s < S ? len[s] : 0;
    const int so = (s < S && slot_state) ? slot_state[s] : s;
    int64_t base_al = 0;
    if (s < S) base_al = seg_off[(int64_t)so * A] & ~(int64_t)(E - 1);
    for (int a = 0; a < A; ++a) {
        const uint32_t rel = s < S ? (uint32_t)(seg_off[(int64_t)so * A + a] - base_al) : 0u;
        pos[a * WAVE + lane] = rel;
        start[a * WAVE + lane] = rel;
    }
    T* __restrict__ vb = values + base_al;
    const int64_t row0 = slice_row_off[w];
    const int nq = (int)((slice_row_off[w + 1] - row0) >> 2);    // quad rows of the slice (wave-uniform)
    int nw = n;                                                  // the longest stream of the slice (wave-uniform): the loop's bound
#pragma unroll
    for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(nw, off); nw = o > nw ? o : nw; }
    nw = __builtin_amdgcn_readfirstlane(nw);
    const int nqw = (nw + 3) >> 2 < nq ? (nw + 3) >> 2 : nq;
    const Q4* __restrict__ Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
    const uchar4* __restrict__ Aq = reinterpret_cast<const uchar4*>(act) + row0 / 4 * WAVE + lane;

    // A STEP = NR records of the lane, without a branch until a line completes.  A record beyond the lane's stream adds 0 to its
    // cursor and parks its value in the NEXT FREE element of that bucket's ring (which the bucket's next real record overwrites).
    // Phase 1: the NR cursor adds and the NR reads of the buckets' first elements go out together (one LDS round trip).  Phase 2: the
    // ring writes — except a record that follows, in the same step, a record of its own bucket which completed the line: it would
    // overwrite the line's first elements before they are read; it is written in phase 4.  Phase 3: completed lines leave — not one
    // branch per record (64 lanes x 1/16: each of the NR branches would run for ~4 lanes) but one per ROUND: every lane that has a
    // completed line left takes its oldest one (NR = 8: ~25 lanes in the first round, ~6 in the second, a third is rare).
    auto step = [&](const uchar4 (&a4)[NQ], const Q4 (&r4)[NQ], int t) __attribute__((always_inline)) {
        int av[NR];
        T rv[NR];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            av[4 * q] = a4[q].x; av[4 * q + 1] = a4[q].y; av[4 * q + 2] = a4[q].z; av[4 * q + 3] = a4[q].w;
            rv[4 * q] = r4[q].x; rv[4 * q + 1] = r4[q].y; rv[4 * q + 2] = r4[q].z; rv[4 * q + 3] = r4[q].w;
        }
        uint32_t p[NR], st0[NR];
        bool valid[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            valid[j] = t + j < n;
            av[j] = av[j] < A ? av[j] : A - 1;                   // (ids are validated where the table is built; never index out of range)
            p[j] = __hip_atomic_fetch_add(&pos[av[j] * WAVE + lane], valid[j] ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            st0[j] = start[av[j] * WAVE + lane];
        }
        uint32_t fmask = 0u, hmask = 0u, lmask = 0u;             // per lane: records that completed a whole line / a bucket's first (partial) line / late ones
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const bool done = valid[j] && ((p[j] + 1u) & (uint32_t)(E - 1)) == 0u;
            const bool whole = p[j] + 1u >= st0[j] + (uint32_t)E;
            bool late = false;
#pragma unroll
            for (int i = 0; i < j; ++i) late = late || (((fmask | hmask) >> i) & 1u && av[i] == av[j]);
            fmask |= (done && whole ? 1u : 0u) << j;
            hmask |= (done && !whole ? 1u : 0u) << j;
            lmask |= (late ? 1u : 0u) << j;
            if (!late) ring[((size_t)av[j] * WAVE + lane) * E + (p[j] & (uint32_t)(E - 1))] = rv[j];
        }
        while (__any(fmask != 0u)) {                             // phase 3, a round: the lane's oldest completed line
            if (fmask != 0u) {
                const int j0 = __ffs((int)fmask) - 1;
                fmask &= fmask - 1u;
                int aj = av[0];
                uint32_t pj = p[0];
#pragma unroll
                for (int j = 1; j < NR; ++j) { aj = j0 == j ? av[j] : aj; pj = j0 == j ? p[j] : pj; }
                const uint4* line = reinterpret_cast<const uint4*>(ring + ((size_t)aj * WAVE + lane) * E);
                uint4 x[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) x[k] = line[k];
                uint4* dst = reinterpret_cast<uint4*>(vb + (pj + 1u - (uint32_t)E));
#pragma unroll
                for (int k = 0; k < NV; ++k) nt_store16(dst + k, x[k].x, x[k].y, x[k].z, x[k].w);
            }
        }
        if (__any(hmask != 0u)) {                                // a bucket that began inside its first line: that line's own elements only
#pragma unroll 1
            for (int j = 0; j < NR; ++j) {
                if ((hmask >> j) & 1u) {
                    const T* line = ring + ((size_t)av[j] * WAVE + lane) * E;
                    for (uint32_t e = st0[j]; e <= p[j]; ++e) vb[e] = line[e & (uint32_t)(E - 1)];
                }
            }
        }
        if (__any(lmask != 0u)) {                                // phase 4: the records that had to wait for their bucket's line to leave
#pragma unroll
            for (int j = 0; j < NR; ++j)
                if ((lmask >> j) & 1u) ring[((size_t)av[j] * WAVE + lane) * E + (p[j] & (uint32_t)(E - 1))] = rv[j];
        }
    };
    constexpr int LD = PF * (sizeof(T) == 4 ? 2 : 3);            // load instructions per bank
    ld_f4 rlo[2][PF], rhi[2][PF];                                // (f64: a quad is two 16-byte loads)
    unsigned ab[2][PF];
    auto consume = [&](auto bank, int q0) __attribute__((always_inline)) {
        constexpr int b = decltype(bank)::value;
#pragma unroll
        for (int i = 0; i < PF; i += NQ) {
            const int t = (q0 + i) * 4;
            if (q0 + i < nqw) {                                  // wave-uniform (a lane whose stream has ended files its records in the trash row)
                uchar4 a4[NQ];
                Q4 r4[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const unsigned x = ab[b][i + q];
                    a4[q] = make_uchar4((unsigned char)(x & 255u), (unsigned char)((x >> 8) & 255u), (unsigned char)((x >> 16) & 255u), (unsigned char)(x >> 24));
                    if constexpr (sizeof(T) == 4) {
                        r4[q].x = rlo[b][i + q].x; r4[q].y = rlo[b][i + q].y; r4[q].z = rlo[b][i + q].z; r4[q].w = rlo[b][i + q].w;
                    } else {
                        r4[q].x = __hiloint2double(__float_as_int(rlo[b][i + q].y), __float_as_int(rlo[b][i + q].x));
                        r4[q].y = __hiloint2double(__float_as_int(rlo[b][i + q].w), __float_as_int(rlo[b][i + q].z));
                        r4[q].z = __hiloint2double(__float_as_int(rhi[b][i + q].y), __float_as_int(rhi[b][i + q].x));
                        r4[q].w = __hiloint2double(__float_as_int(rhi[b][i + q].w), __float_as_int(rhi[b][i + q].z));
                    }
                }
                step(a4, r4, t);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if (nqw > 0) {
        // (the set-up's own loads — len, seg_off — were waited for by the compiler where it used them: nothing older is outstanding)
        rg_load<T, PF>(rlo[0], rhi[0], ab[0], Rq, Aq, 0, nq);
        for (int q0 = 0; q0 < nqw; q0 += 2 * PF) {
            rg_load<T, PF>(rlo[1], rhi[1], ab[1], Rq, Aq, q0 + PF, nq);
            rg_arrived<T, LD>(rlo[0], rhi[0], ab[0]);
            consume(B0{}, q0);
            rg_load<T, PF>(rlo[0], rhi[0], ab[0], Rq, Aq, q0 + 2 * PF, nq);
            rg_arrived<T, LD>(rlo[1], rhi[1], ab[1]);
            consume(B1{}, q0 + PF);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the last bank requested is never consumed: nothing of it may land later
    }
    if (s < S) {                                                 // the lines still in progress: their elements one by one
        for (int a = 0; a < A; ++a) {
            const uint32_t hi = pos[a * WAVE + lane], st0 = start[a * WAVE + lane];
            uint32_t lo = hi & ~(uint32_t)(E - 1);
            lo = lo > st0 ? lo : st0;
            const T* line = ring + ((size_t)a * WAVE + lane) * E;
            for (uint32_t e = lo; e < hi; ++e) vb[e] = line[e & (uint32_t)(E - 1)];
        }
    }
}

template <typename T, int LINE, int NQ>
static void launch_regroup(int W, hipStream_t st, const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                           const int32_t* slot_state, int S, int A, const int64_t* seg_off, T* values) {
    DCARL_RAISE_LDS_LIMIT(((int)regroup_lds<T, LINE>(DCARL_MAX_ACTIONS)), regroup_kernel<T, LINE, NQ>);
    const unsigned lds = regroup_lds<T, LINE>(A);
    hipLaunchKernelGGL((regroup_kernel<T, LINE, NQ>), dim3((unsigned)W), dim3(WAVE), lds, st, R, act, slice_row_off, len,
                       slot_state, S, A, seg_off, values);
}

// ---- group_records, the chunk-sort form (round 5, second cut) -------------------------------------------------------------------
// The write-combining kernel above is bound by ONE wavefront's latency chain (cursor add -> ring write -> line read -> stores, ~450
// cycles per quad) with only three wavefronts per CU (its rings fill the LDS): 4.2 ms per 20 000-record stream however many slices run.
// Here a 256-thread block owns a slice and walks it in CHUNKS of C = 128 records per state; wave k takes the k-th quarter of the
// chunk's time range for all 64 states (lane = state), so a (state, action) bucket's records stay in arrival order as long as the
// quarters are ranked one after the other:
//   count   fire-and-forget LDS adds on [wave][action][lane]                                        (barrier)
//   offsets thread (wave, lane): prefix over the actions and the earlier waves -> where this wave's records of each action go in the
//           state's chunk sorted by action; wave 0 also leaves the chunk's action offsets
//   place   one returning add per record on the wave's own cursor -> value and action tag into the staging row of the state   (barrier)
//   write   wave k, states k, k+4, ...: lane = element of the sorted chunk, its tag says which bucket, consecutive lanes of a run store
//           consecutive addresses                                                                    (barrier)
//   advance the buckets' positions by the chunk's counts, zero the counters                          (barrier)
// A bucket's piece of a chunk is ~12 records: the lines it shares with the previous and the next chunk are completed within a few
// microseconds by the same block, in the L2.  Two blocks (8 wavefronts) per CU at 11 actions.
template <typename T, int QW, int RS_WAVES> constexpr unsigned regroup_sort_lds(int A) {
    constexpr int C = RS_WAVES * QW * 4;
    return (unsigned)(WAVE * (C + 1) * sizeof(T) + WAVE * C + ((2 * RS_WAVES * A + A + A + 1) * WAVE) * 4 + WAVE * 8);
}
template <typename T, int QW, int RS_WAVES>
__global__ __launch_bounds__(RS_WAVES* WAVE) void regroup_sort_kernel(
    const T* __restrict__ R, const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off,
    const int32_t* __restrict__ len, const int32_t* __restrict__ slot_state, int S, int A, const int64_t* __restrict__ seg_off,
    T* __restrict__ values) {
    using Q4 = typename std::conditional<sizeof(T) == 4, float4, double4>::type;
    constexpr int C = RS_WAVES * QW * 4;                         // records per state and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* stage = reinterpret_cast<T*>(smem);                       // [WAVE][C + 1]: the state's chunk sorted by action (odd stride: lanes spread over the banks)
    int64_t* vbase = reinterpret_cast<int64_t*>(smem + (size_t)WAVE * (C + 1) * sizeof(T) + ((WAVE * (C + 1) * sizeof(T)) & 7));
    uint32_t* cnt = reinterpret_cast<uint32_t*>(vbase + WAVE);   // [RS_WAVES][A][WAVE]
    uint32_t* lb = cnt + RS_WAVES * A * WAVE;                    // [RS_WAVES][A][WAVE]: wave k's next place of action a in the state's sorted chunk
    uint32_t* pos = lb + RS_WAVES * A * WAVE;                    // [A][WAVE]: next element of the bucket, relative to the state's first
    uint32_t* off = pos + A * WAVE;                              // [A + 1][WAVE]: the chunk's action offsets
    uint8_t* tag = reinterpret_cast<uint8_t*>(off + (A + 1) * WAVE);   // [WAVE][C]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), k = tid >> 6;
    const int w = blockIdx.x;
    const int s = w * WAVE + lane;
    const int n = s < S ? len[s] : 0;
    const int so = (s < S && slot_state) ? slot_state[s] : s;
    if (k == 0) {
        const int64_t b0 = s < S ? seg_off[(int64_t)so * A] : 0;
        vbase[lane] = b0;
        for (int a = 0; a < A; ++a) pos[a * WAVE + lane] = s < S ? (uint32_t)(seg_off[(int64_t)so * A + a] - b0) : 0u;
    }
    for (int a = 0; a < A; ++a) cnt[(k * A + a) * WAVE + lane] = 0u;
    const int64_t row0 = slice_row_off[w];
    const int nq = (int)((slice_row_off[w + 1] - row0) >> 2);
    const Q4* __restrict__ Rq = reinterpret_cast<const Q4*>(R) + row0 / 4 * WAVE + lane;
    const uchar4* __restrict__ Aq = reinterpret_cast<const uchar4*>(act) + row0 / 4 * WAVE + lane;
    const int nchunks = (nq * 4 + C - 1) / C;
    Q4 rq[QW];
    uchar4 aq[QW];
    auto load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const int q = c * (C / 4) + k * QW + i;
            rq[i] = Q4{}; aq[i] = make_uchar4(0, 0, 0, 0);
            if (q < nq) { rq[i] = Rq[(int64_t)q * WAVE]; aq[i] = Aq[(int64_t)q * WAVE]; }
        }
    };
    if (nchunks > 0) load(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int t0 = c * C + k * QW * 4;                       // this wave's first record of the chunk
        int av[QW * 4];
        T rv[QW * 4];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            av[4 * i] = aq[i].x; av[4 * i + 1] = aq[i].y; av[4 * i + 2] = aq[i].z; av[4 * i + 3] = aq[i].w;
            rv[4 * i] = rq[i].x; rv[4 * i + 1] = rq[i].y; rv[4 * i + 2] = rq[i].z; rv[4 * i + 3] = rq[i].w;
        }
#pragma unroll
        for (int j = 0; j < QW * 4; ++j) {
            av[j] = av[j] < A ? av[j] : A - 1;                   // (ids are validated where the table is built; never index out of range)
            if (t0 + j < n) atomicAdd(&cnt[(k * A + av[j]) * WAVE + lane], 1u);
        }
        if (c + 1 < nchunks) load(c + 1);                        // the next chunk's records travel under this chunk's ranking and write-out
        __syncthreads();
        {
            uint32_t run = 0;
            for (int a = 0; a < A; ++a) {
                uint32_t ck[RS_WAVES];
#pragma unroll
                for (int kk = 0; kk < RS_WAVES; ++kk) ck[kk] = cnt[(kk * A + a) * WAVE + lane];
                uint32_t mine = run;
#pragma unroll
                for (int kk = 0; kk < RS_WAVES; ++kk) mine += kk < k ? ck[kk] : 0u;
                lb[(k * A + a) * WAVE + lane] = mine;
                if (k == 0) off[a * WAVE + lane] = run;
#pragma unroll
                for (int kk = 0; kk < RS_WAVES; ++kk) run += ck[kk];
            }
            if (k == 0) off[A * WAVE + lane] = run;
        }
#pragma unroll
        for (int j = 0; j < QW * 4; ++j) {
            if (t0 + j < n) {
                const uint32_t pl = __hip_atomic_fetch_add(&lb[(k * A + av[j]) * WAVE + lane], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                stage[lane * (C + 1) + pl] = rv[j];
                tag[lane * C + pl] = (uint8_t)av[j];
            }
        }
        __syncthreads();
        for (int sp = k; sp < WAVE; sp += RS_WAVES) {            // wave-uniform state
            const uint32_t tot = off[A * WAVE + sp];
            T* __restrict__ vb = values + vbase[sp];
            for (uint32_t e = lane; e < tot; e += WAVE) {
                const int a = tag[sp * C + e];
                const uint32_t dst = pos[a * WAVE + sp] + (e - off[a * WAVE + sp]);
                vb[dst] = stage[sp * (C + 1) + e];
            }
        }
        __syncthreads();
        for (int i = tid; i < A * WAVE; i += RS_WAVES * WAVE) {
            pos[i] += off[i + WAVE] - off[i];
#pragma unroll
            for (int kk = 0; kk < RS_WAVES; ++kk) cnt[kk * A * WAVE + i] = 0u;
        }
        __syncthreads();
    }
}

// ---- count_records: n[s][a] alone (reads the action bytes only).  Lane = state, fire-and-forget LDS adds on [action][lane] counters
// (bank = lane), sixteen quad rows of actions requested before the first is counted; four slices per block, five blocks per CU.
constexpr int COUNT_PF = 16;
__global__ __launch_bounds__(GROUP_WAVES* WAVE) void count_records_kernel(
    const uint8_t* __restrict__ act, const int64_t* __restrict__ slice_row_off, const int32_t* __restrict__ len,
    const int32_t* __restrict__ slot_state, int S, int A, int32_t* __restrict__ n_out) {
    __shared__ uint32_t cnt[GROUP_WAVES][DCARL_MAX_ACTIONS][WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const int w = blockIdx.x * GROUP_WAVES + wv;
    if (w * WAVE >= S) return;                                   // wave-uniform
    const int s = w * WAVE + lane;
    const int n = s < S ? len[s] : 0;
    for (int a = 0; a < A; ++a) cnt[wv][a][lane] = 0u;
    const int64_t row0 = slice_row_off[w];
    const int nq = (int)((slice_row_off[w + 1] - row0) >> 2);
    const uint32_t* __restrict__ Aq = reinterpret_cast<const uint32_t*>(act) + row0 / 4 * WAVE + lane;
    for (int q0 = 0; q0 < nq; q0 += COUNT_PF) {
        uint32_t x[COUNT_PF];
#pragma unroll
        for (int i = 0; i < COUNT_PF; ++i) x[i] = (q0 + i < nq) ? __builtin_nontemporal_load(Aq + (int64_t)(q0 + i) * WAVE) : 0u;
#pragma unroll
        for (int i = 0; i < COUNT_PF; ++i) {
            const int t = (q0 + i) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (q0 + i < nq && t + j < n) {
                    const int a0 = (int)((x[i] >> (8 * j)) & 255u);
                    atomicAdd(&cnt[wv][a0 < A ? a0 : A - 1][lane], 1u);           // (result unused: a ds_add without return)
                }
            }
        }
    }
    if (s < S) {
        const int so = slot_state ? slot_state[s] : s;
        for (int a = 0; a < A; ++a) n_out[(int64_t)so * A + a] = (int32_t)cnt[wv][a][lane];
    }
}

template <typename T, int QW, int RS_WAVES = 4>
static void launch_regroup_sort(int W, hipStream_t st, const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len,
                                const int32_t* slot_state, int S, int A, const int64_t* seg_off, T* values) {
    DCARL_RAISE_LDS_LIMIT((160 * 1024), regroup_sort_kernel<T, QW, RS_WAVES>);
    const unsigned lds = regroup_sort_lds<T, QW, RS_WAVES>(A);
    hipLaunchKernelGGL((regroup_sort_kernel<T, QW, RS_WAVES>), dim3((unsigned)W), dim3(RS_WAVES * WAVE), lds, st, R, act, slice_row_off, len, slot_state, S, A,
                       seg_off, values);
}

template <typename T>
int launch_group_records(const T* R, const uint8_t* act, const int64_t* slice_row_off, const int32_t* len, const int32_t* slot_state, int S, int A,
                         const int64_t* seg_off, T* values, int32_t* n_out, hipStream_t st) {
    if (S == 0) return 0;
    const int W = slices_of(S);
    dim3 grid((W + GROUP_WAVES - 1) / GROUP_WAVES), block(GROUP_WAVES * WAVE);
    if (values) {
#ifdef DCARL_AB_BUILD
        // the two forms this one replaced, for same-box comparisons (tools/bench_regroup.py) — A/B variant of the library only:
        // DCARL_GROUP_RECORDS=scatter (element-wise scatter: 27-33 ms on configs[1]), =wc (a 64-byte line per (lane, action) in LDS,
        // whole-line stores, one wavefront per slice: 7.9 ms), =q4 (this kernel with chunks of 64 records per state: 6.0 ms), =4 (four wavefronts
        // per slice whatever A: 4.8 ms)
        if (const char* e = DCARL_KNOB("DCARL_GROUP_RECORDS")) {
            if (e[0] == 's') {
                hipLaunchKernelGGL((group_records_kernel<T, true>), grid, block, 0, st, R, act, slice_row_off, len, slot_state, S, A, seg_off,
                                   values, n_out);
                return 0;
            }
            if (e[0] == 'w') { launch_regroup<T, 64, 1>(W, st, R, act, slice_row_off, len, slot_state, S, A, seg_off, values); return 0; }
            if (e[0] == 'q') { launch_regroup_sort<T, 4>(W, st, R, act, slice_row_off, len, slot_state, S, A, seg_off, values); return 0; }
            if (e[0] == '4') { launch_regroup_sort<T, (sizeof(T) == 4 ? 8 : 4), 4>(W, st, R, act, slice_row_off, len, slot_state, S, A, seg_off, values); return 0; }
        }
#endif
        // eight wavefronts per slice (chunks of 256 / 128 records per state: a bucket's piece of a chunk is twice as long, half as many
        // lines are shared between chunks) while their tables fit the LDS: 4.76 -> 4.04 ms on configs[1], 2.28 -> 1.53 on the configs[4] shard
        if (A <= 16) launch_regroup_sort<T, (sizeof(T) == 4 ? 8 : 4), 8>(W, st, R, act, slice_row_off, len, slot_state, S, A, seg_off, values);
        else launch_regroup_sort<T, (sizeof(T) == 4 ? 8 : 4), 4>(W, st, R, act, slice_row_off, len, slot_state, S, A, seg_off, values);
    } else
        hipLaunchKernelGGL(count_records_kernel, grid, block, 0, st, act, slice_row_off, len, slot_state, S, A, n_out);
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
