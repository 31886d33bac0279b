// One radix pass on random keys: the shipped two-array pass (rx_scatter_kernel) against the whole-line pass on 8-byte records
// (rx_scatter_lines_kernel), results compared record for record.   ubench_scatter_lines [N] [shift] [bits]
#include "../dcarl_amd/csrc/ingest.hip"
#include <cstdio>
#include <vector>
namespace dcarl { void note_kernel(const char*, ...) {} }
using namespace dcarl;

uint32_t *g_start, *g_end;
template <int TH, int G, int LR, bool BD>
float run_lines(const uint2* in, uint2* out, uint32_t n, int shift, int bits, uint32_t blk, const uint32_t* hist, int nblk, const uint32_t* tot,
                hipEvent_t e0, hipEvent_t e1) {
    constexpr unsigned lds = rx_lines_lds<TH, G, LR, BD>();
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rx_scatter_lines_kernel<TH, G, LR, BD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rx_scatter_lines_kernel<TH, G, LR, BD>), dim3(nblk), dim3(TH), lds, 0, in, out, n, shift, bits, blk, hist, nblk, tot, g_start, g_end, 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("launch error: %s\n", hipGetErrorString(err));
    return best;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1ll << 28);
    const int shift = argc > 2 ? atoi(argv[2]) : 5, bits = argc > 3 ? atoi(argv[3]) : 8;
    uint32_t blk; int nblk;
    block_split(N, &blk, &nblk);
    uint32_t *k0, *k1, *hist, *tot; float *v0, *v1; uint2 *r0, *r1;
    hipMalloc(&k0, N * 4); hipMalloc(&k1, N * 4); hipMalloc(&v0, N * 4); hipMalloc(&v1, N * 4); hipMalloc(&r0, N * 8); hipMalloc(&r1, N * 8);
    hipMalloc(&hist, (size_t)RX_DIGITS * nblk * 4); hipMalloc(&tot, RX_DIGITS * 4);
    std::vector<uint32_t> h(N), hv(N);
    std::vector<uint2> hr(N);
    uint64_t x = 88172645463325252ull;
    for (int64_t i = 0; i < N; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (uint32_t)((x >> 20) & 0xffff) << 5 | (uint32_t)(x & 15) % 11; hv[i] = (uint32_t)i;
        hr[i] = make_uint2(h[i], hv[i]);
    }
    hipMemcpy(k0, h.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(r0, hr.data(), N * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rx_hist_kernel, dim3(nblk), dim3(RX_THREADS), 0, 0, k0, (uint32_t)N, shift, bits, blk, hist, nblk);
    hipLaunchKernelGGL(rx_scan_kernel, dim3(1 << bits), dim3(256), 0, 0, hist, nblk, tot);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        launch_scatter<4, false>(k0, v0, nullptr, k1, v1, nullptr, (uint32_t)N, shift, bits, blk, hist, nblk, tot, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("N=%lld shift %d bits %d blk=%u nblk=%d\n  two arrays, runs as they fall: %.3f ms = %.2f TB/s of 16 B/record\n", (long long)N, shift, bits, blk, nblk, best, N * 16.0 / best / 1e9);
    hipMemcpy(h.data(), k1, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hv.data(), v1, N * 4, hipMemcpyDeviceToHost);
    auto check = [&](const char* name, float ms) {
        hipMemcpy(hr.data(), r1, N * 8, hipMemcpyDeviceToHost);
        int64_t bad = 0, first = -1;
        for (int64_t i = 0; i < N; ++i) if (hr[i].x != h[i] || hr[i].y != hv[i]) { if (!bad) first = i; ++bad; }
        printf("  whole lines, %s: %.3f ms = %.2f TB/s   %s", name, ms, N * 16.0 / ms / 1e9, bad ? "MISMATCH" : "identical result\n");
        if (bad) printf(" %lld records differ, first at %lld\n", (long long)bad, (long long)first);
        hipMemset(r1, 0xff, N * 8);
    };
    hipMemset(r1, 0xff, N * 8);
#define RUN(TH, G, LR, BD) check(#TH " threads x " #G ", " #LR " records per store unit, run reports " #BD, run_lines<TH, G, LR, BD>(r0, r1, (uint32_t)N, shift, bits, blk, hist, nblk, tot, e0, e1))
    hipMalloc(&g_start, (1 << 22) * 4); hipMalloc(&g_end, (1 << 22) * 4); hipMemset(g_start, 0xff, (1 << 22) * 4); hipMemset(g_end, 0, (1 << 22) * 4);
    RUN(512, 13, 8, false); RUN(512, 12, 8, true); RUN(512, 9, 16, false); RUN(1024, 13, 16, false);
    return 0;
}
