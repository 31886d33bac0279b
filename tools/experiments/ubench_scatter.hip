// Where does a scatter pass of the ingest spend its time?  Includes the kernels themselves (ingest.hip) and times ONE pass on
// random keys with parts of the kernel compiled out:   hipcc -DRX_EXP=<0|1|3|4> tools/experiments/ubench_scatter.hip
//   0 the shipped pass   1 no global stores   3 copy only (no ranking, no staging)   4 staged but written sequentially
//   5 {key, value} as one 8-byte record (array of structures) instead of two arrays
#include "../dcarl_amd/csrc/ingest.hip"
#include <cstdio>
#include <vector>
namespace dcarl { void note_kernel(const char*, ...) {} }
using namespace dcarl;
int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1ll << 28);
    const int th = argc > 2 ? atoi(argv[2]) : 0;
    if (th) setenv("DCARL_INGEST_SCATTER_THREADS", th == 256 ? "256" : "512", 1);
    uint32_t blk; int nblk;
    block_split(N, &blk, &nblk);
    uint32_t *k0, *k1, *hist, *tot; float *v0, *v1;
    hipMalloc(&k0, N * 8); hipMalloc(&k1, N * 8); hipMalloc(&v0, N * 4); hipMalloc(&v1, N * 4);
    hipMalloc(&hist, (size_t)RX_DIGITS * nblk * 4); hipMalloc(&tot, RX_DIGITS * 4);
    std::vector<uint32_t> h(N);
    uint64_t x = 88172645463325252ull;
    for (int64_t i = 0; i < N; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)((x >> 20) & 0xffff) << 5 | (uint32_t)(x & 15) % 11; }
#if defined(RX_EXP) && RX_EXP == 5
    { std::vector<uint32_t> h2(2 * N); for (int64_t i = 0; i < N; ++i) { h2[2 * i] = h[i]; h2[2 * i + 1] = (uint32_t)i; } hipMemcpy(k0, h2.data(), N * 8, hipMemcpyHostToDevice); }
#else
    hipMemcpy(k0, h.data(), N * 4, hipMemcpyHostToDevice);
#endif
    hipMemset(v0, 0, N * 4);
    uint32_t* ks; hipMalloc(&ks, N * 4); hipMemcpy(ks, h.data(), N * 4, hipMemcpyHostToDevice);   // the keys alone, for the histogram
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(rx_hist_kernel, dim3(nblk), dim3(RX_THREADS), 0, 0, ks, (uint32_t)N, 5, 8, blk, hist, nblk);
        hipLaunchKernelGGL(rx_scan_kernel, dim3(256), dim3(256), 0, 0, hist, nblk, tot);
        hipEventRecord(e0);
        launch_scatter<4, false>(k0, v0, nullptr, k1, v1, nullptr, (uint32_t)N, 5, 8, blk, hist, nblk, tot, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("RX_EXP=%d N=%lld blk=%u nblk=%d threads=%s: %.3f ms = %.2f TB/s of 16 B/record\n",
#ifdef RX_EXP
                        RX_EXP,
#else
                        0,
#endif
                        (long long)N, blk, nblk, th ? argv[2] : "auto", ms, N * 16.0 / ms / 1e9);
    }
    return 0;
}
