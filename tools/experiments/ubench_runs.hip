// What does the DESTINATION pattern of a radix-scatter pass cost on MI355X?  Every block owns ST output streams and, per turn,
// appends a run of RUN 4-byte words to each of them (consecutive lanes -> consecutive words of a run, then the next stream:
// the write-out of ingest.hip's rx_scatter_kernel), nothing else: no reads, no ranking.
//   ubench_runs <run words> <streams> <misalign 0|1> <lds bytes per block (occupancy)> <nontemporal 0|1> <read too 0|1>
// misalign = 2 starts every stream at a random 64-byte boundary;
// misalign = 1 starts every stream at a random word inside a 128-byte line (what a counting sort's offsets are), 0 at a line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <bool NT, bool RD>
__global__ __launch_bounds__(512) void runs_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, const uint32_t* __restrict__ start,
                                                    int run, int streams, int turns, uint64_t stream_words) {
    extern __shared__ uint32_t pad[];
    if (threadIdx.x == 9999) pad[0] = 1;
    const int per_turn = streams * run;                       // words a block writes per turn
    const uint64_t b0 = (uint64_t)blockIdx.x * streams;
    const uint32_t* src = in + (uint64_t)blockIdx.x * turns * per_turn;
    for (int t = 0; t < turns; ++t) {
        for (int i = threadIdx.x; i < per_turn; i += 512) {
            const int s = i / run, w = i - s * run;
            uint32_t v = i;
            if (RD) v = src[(uint64_t)t * per_turn + i];
            uint32_t* p = out + (b0 + s) * stream_words + start[b0 + s] + (uint64_t)t * run + w;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

int main(int argc, char** argv) {
    const int run = argc > 1 ? atoi(argv[1]) : 32, streams = argc > 2 ? atoi(argv[2]) : 256, mis = argc > 3 ? atoi(argv[3]) : 1;
    const int lds = argc > 4 ? atoi(argv[4]) : 65536, nt = argc > 5 ? atoi(argv[5]) : 0, rd = argc > 6 ? atoi(argv[6]) : 0;
    const int nblk = argc > 7 ? atoi(argv[7]) : 2048;
    const uint64_t total_words = 1ull << 30;                  // 4 GiB written
    const int turns = (int)(total_words / ((uint64_t)nblk * streams * run));
    const uint64_t stream_words = (uint64_t)turns * run + 64;
    uint32_t *out, *in, *start;
    hipMalloc(&out, (uint64_t)nblk * streams * stream_words * 4 + 4096);
    hipMalloc(&in, total_words * 4);
    hipMemset(in, 1, total_words * 4);
    hipMalloc(&start, (size_t)nblk * streams * 4);
    uint32_t* h = (uint32_t*)malloc((size_t)nblk * streams * 4);
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < nblk * streams; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        // stream_words is not a multiple of 32: cancel the base's misalignment first so that mis = 0 is line-aligned
        const uint64_t base = (uint64_t)i * stream_words;
        const uint32_t fix = (uint32_t)((32 - base % 32) % 32);
        h[i] = mis == 1 ? (uint32_t)(x >> 33) % 32 : mis == 2 ? (fix + 16 * ((uint32_t)(x >> 33) % 2)) % 32 + (fix + 16 >= 32 && ((x >> 33) % 2) ? 0 : 0) : fix;
    }
    hipMemcpy(start, h, (size_t)nblk * streams * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&runs_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&runs_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&runs_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&runs_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (nt && rd) hipLaunchKernelGGL((runs_kernel<true, true>), dim3(nblk), dim3(512), lds, 0, out, in, start, run, streams, turns, stream_words);
        else if (nt) hipLaunchKernelGGL((runs_kernel<true, false>), dim3(nblk), dim3(512), lds, 0, out, in, start, run, streams, turns, stream_words);
        else if (rd) hipLaunchKernelGGL((runs_kernel<false, true>), dim3(nblk), dim3(512), lds, 0, out, in, start, run, streams, turns, stream_words);
        else hipLaunchKernelGGL((runs_kernel<false, false>), dim3(nblk), dim3(512), lds, 0, out, in, start, run, streams, turns, stream_words);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nblk * streams * run * turns * 4.0;
        if (rep) printf("run %4d B  streams %4d  misaligned %d  lds %6d  nt %d  read %d  blocks %d: %.3f ms  write %.2f TB/s%s\n", run * 4, streams, mis, lds, nt, rd, nblk,
                        ms, bytes / ms / 1e9, rd ? " (+ the same read)" : "");
    }
    return 0;
}
