// Micro-benchmark: what read bandwidth does a box sustain for (a) a fully coalesced float4 stream, (b) the final-state
// kernel's pattern — 4-lane clusters each reading 64 B pieces of their own ~364 B segment — and what do (c) a few
// percent of interleaved writes, (d) f64 arithmetic per sample, (e) one short-lived wave per 64 KB cost on top of it.
// hipcc --offload-arch=gfx950 -O3 -w tools/experiments/ubench_stream.hip -o tools/experiments/ubench_stream.bin && tools/experiments/ubench_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: lane l of wave w reads float4 index (w*NV + i)*64 + l           (each instruction = 1 KiB contiguous)
// MODE G (4,8,16): cluster c = l/G of wave w owns segment (w*(64/G) + c) of SEG float4s; lane reads seg*SEG + i*G + l%G
// WR: the wave writes 128 + 64 bytes per iteration (like V_out / n_out of a pass).  FL: f64 work per sample like acc16.
// ONESHOT: a wave does 11 iterations (one "task") and exits; the grid covers the buffer (no grid-stride loop).
template <int MODE, int NV, int SEG, int WR, bool FL, bool ONESHOT>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ in, double* __restrict__ out, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = 0.0, q = 0.0;
    const long wb = ONESHOT ? w0 * 11 : w0, we = ONESHOT ? (w0 * 11 + 11 < nwaves ? w0 * 11 + 11 : nwaves) : nwaves;
    const long ws = ONESHOT ? 1 : (long)gridDim.x * 4;
    for (long w = wb; w < we; w += ws) {
        float4 x[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            long idx;
            if (MODE == 0) idx = (w * NV + i) * 64 + lane;
            else idx = (w * (64 / MODE) + lane / MODE) * (long)SEG + i * MODE + lane % MODE;
            x[i] = in[idx];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (FL) {
                const double K = 1.5;
                double a = x[i].x - K, b = x[i].y - K, c = x[i].z - K, d = x[i].w - K;
                acc += (a + b) + (c + d);
                q = fma(a, a, q); q = fma(b, b, q); q = fma(c, c, q); q = fma(d, d, q);
            } else acc += x[i].x + x[i].y + x[i].z + x[i].w;
        }
        if (WR == 1) {
            if ((lane & 3) == 0) out[w * 16 + (lane >> 2)] = acc + q;                      // 128 B per wave-iteration
            if ((lane & 3) == 0) reinterpret_cast<int*>(out + nwaves * 16)[w * 16 + (lane >> 2)] = (int)q;   // 64 B
        }
        if (WR == 2) {                                                                     // the same, non-temporal
            if ((lane & 3) == 0) __builtin_nontemporal_store(acc + q, &out[w * 16 + (lane >> 2)]);
            if ((lane & 3) == 0) __builtin_nontemporal_store((int)q, &reinterpret_cast<int*>(out + nwaves * 16)[w * 16 + (lane >> 2)]);
        }
    }
    if (WR >= 3) {                                                                         // once per task: 1408 + 704 B, coalesced
        double* o1 = out + wb * 16;
        int* o2 = reinterpret_cast<int*>(out + nwaves * 16) + wb * 16;
        for (int i = lane; i < 176; i += 64) {
            if (WR == 3) { o1[i] = acc + q; o2[i] = (int)q; }
            else { __builtin_nontemporal_store(acc + q, &o1[i]); __builtin_nontemporal_store((int)q, &o2[i]); }
        }
    }
    if (!WR && acc + q == 12345.678) out[0] = acc;
}

template <int MODE, int NV, int SEG, int WR, bool FL, bool ONESHOT>
void run(const char* name, const float4* d, double* o, size_t bytes, int blocks) {
    const long per_wave = (MODE == 0) ? (long)NV * 64 * 16 : (long)(64 / MODE) * SEG * 16;
    const long nwaves = bytes / per_wave;
    const double moved = (MODE == 0) ? (double)nwaves * per_wave : (double)nwaves * (64 / MODE) * NV * MODE * 16;
    if (ONESHOT) blocks = (int)((nwaves + 43) / 44);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV, SEG, WR, FL, ONESHOT>), dim3(blocks), dim3(256), 0, 0, d, o, nwaves);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, NV, SEG, WR, FL, ONESHOT>), dim3(blocks), dim3(256), 0, 0, d, o, nwaves);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s blocks %6d  %8.1f GB/s read\n", name, blocks, moved * 5 / (ms * 1e-3) / 1e9);
}

// the other two traffic mixes of the path: write-only (the samplers) and read + write in equal parts with the online
// kernel's element sizes (16 B + 4 B read, 16 B + 4 B written per lane and step)
template <int MODE>
__global__ __launch_bounds__(256) void kw(const float4* __restrict__ in, const unsigned* __restrict__ in2, float4* __restrict__ out,
                                          unsigned* __restrict__ out2, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (MODE == 0) { out[i] = make_float4((float)i, 1.f, 2.f, 3.f); }                          // write-only, 16 B
        if (MODE == 1) { out[i] = make_float4((float)i, 1.f, 2.f, 3.f); out2[i] = (unsigned)i; }   // write-only, 16 + 4 B
        if (MODE == 2) { float4 v = in[i]; unsigned a = in2[i]; v.x += 1.f; out[i] = v; out2[i] = a + 1u; }   // copy, 20 B each way
        if (MODE == 3) { float4 v = in[i]; v.x += 1.f; out[i] = v; }                                // copy, 16 B each way
    }
}
template <int MODE>
void runw(const char* name, float4* a, unsigned* a2, float4* b, unsigned* b2, long n, double bytes_per_elem) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kw<MODE>), dim3(8192), dim3(256), 0, 0, a, a2, b, b2, n);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kw<MODE>), dim3(8192), dim3(256), 0, 0, a, a2, b, b2, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.1f GB/s (read + written)\n", name, bytes_per_elem * n * 5 / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = 4ull << 30;
    float4* d; double* o;
    hipMalloc(&d, bytes + (1 << 20)); hipMalloc(&o, 512ull << 20);
    hipMemset(d, 0, bytes);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 6, 24, 0, true, true>("4-lane clusters x 24 vectors, f64, short-lived waves, no writes", d, o, bytes, 0);
        run<4, 6, 24, 1, true, true>("  + 128 B + 64 B written per pass (3 %)", d, o, bytes, 0);
        run<4, 6, 24, 2, true, true>("  + the same, non-temporal stores", d, o, bytes, 0);
        run<4, 6, 24, 3, true, true>("  + written once per task (1408 B + 704 B, coalesced)", d, o, bytes, 0);
        run<4, 6, 24, 4, true, true>("  + once per task, non-temporal", d, o, bytes, 0);
    }
    {
        const long n = 1l << 28;                                  // 4 GiB of float4 + 1 GiB of dwords, twice
        float4 *a, *b; unsigned *a2, *b2;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&a2, n * 4); hipMalloc(&b2, n * 4);
        hipMemset(a, 0, n * 16); hipMemset(a2, 0, n * 4);
        for (int rep = 0; rep < 2; ++rep) {
            runw<0>("write-only, 16 B per lane", a, a2, b, b2, n, 16);
            runw<1>("write-only, 16 + 4 B per lane (sampler: R + act)", a, a2, b, b2, n, 20);
            runw<3>("copy, 16 B read + 16 B written", a, a2, b, b2, n, 32);
            runw<2>("copy, 16 + 4 B read and written (the online kernel's mix)", a, a2, b, b2, n, 40);
        }
    }
    return 0;
}
