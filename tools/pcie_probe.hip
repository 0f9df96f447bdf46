// pcie_probe.hip -- what a chain read-back can cost on this box: pinned allocation, host registration, D2H into pinned /
// registered / pageable memory, and a multi-threaded unpack of a pinned staging buffer.  hipcc -O2 -o pcie_probe pcie_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t GB = 1ull << 30, n = 2 * GB;
    void* d; CK(hipMalloc(&d, n)); CK(hipMemset(d, 1, n));
    hipStream_t s; CK(hipStreamCreate(&s));
    double t = now(); void* hp; CK(hipHostMalloc(&hp, n, hipHostMallocDefault)); printf("hipHostMalloc 2 GiB: %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    for (int r = 0; r < 2; ++r) { t = now(); CK(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); printf("D2H pinned: %.1f GB/s\n", n / (now() - t) / 1e9); }
    char* pg = (char*)malloc(n);
    t = now(); CK(hipMemcpy(pg, d, n, hipMemcpyDeviceToHost)); printf("D2H pageable (first touch): %.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpy(pg, d, n, hipMemcpyDeviceToHost)); printf("D2H pageable (touched): %.1f GB/s\n", n / (now() - t) / 1e9);
    char* pg2 = (char*)malloc(n);
    t = now(); CK(hipHostRegister(pg2, n, hipHostRegisterDefault)); printf("hipHostRegister 2 GiB untouched: %.3f s (%.1f GB/s)\n", now() - t, n / (now() - t) / 1e9);
    t = now(); CK(hipMemcpyAsync(pg2, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); printf("D2H registered: %.1f GB/s\n", n / (now() - t) / 1e9);
    t = now(); CK(hipHostUnregister(pg2)); printf("hipHostUnregister: %.3f s\n", now() - t);
    // threaded unpack: T threads, each copies its slices pinned -> fresh pageable, chunk by chunk behind its own DMA
    for (int T : {1, 2, 4, 8, 12}) {
        char* dst = (char*)malloc(n);
        const size_t chunk = 32ull << 20;
        std::vector<hipStream_t> st(T); for (auto& x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        t = now();
        std::vector<std::thread> th;
        const size_t nchunks = n / chunk;
        for (int k = 0; k < T; ++k) th.emplace_back([&, k] {
            (void)hipSetDevice(0);
            char* stage = (char*)hp + (size_t)k * 2 * chunk;              // two staging chunks per thread inside the pinned block
            size_t mine = 0; std::vector<size_t> ids; for (size_t c = k; c < nchunks; c += T) ids.push_back(c);
            if (ids.empty()) return;
            (void)hipMemcpyAsync(stage, (char*)d + ids[0] * chunk, chunk, hipMemcpyDeviceToHost, st[k]);
            for (size_t j = 0; j < ids.size(); ++j) {
                (void)hipStreamSynchronize(st[k]);
                if (j + 1 < ids.size()) (void)hipMemcpyAsync(stage + ((j + 1) & 1) * chunk, (char*)d + ids[j + 1] * chunk, chunk, hipMemcpyDeviceToHost, st[k]);
                memcpy(dst + ids[j] * chunk, stage + (j & 1) * chunk, chunk);
                ++mine;
            }
        });
        for (auto& x : th) x.join();
        printf("threaded staged D2H into fresh pageable, %2d threads: %.1f GB/s\n", T, n / (now() - t) / 1e9);
        bool ok = dst[0] == 1 && dst[n - 1] == 1 && dst[n / 2 + 12345] == 1;
        if (!ok) printf("  DATA MISMATCH\n");
        free(dst);
        for (auto& x : st) (void)hipStreamDestroy(x);
    }
    return 0;
}
