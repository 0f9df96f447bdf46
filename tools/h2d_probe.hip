// H2D paths for a large host batch: pinned staging (memcpy + copy), straight from pageable memory, threaded staging.
// hipcc -O2 --offload-arch=gfx950 tools/h2d_probe.hip -o tools/h2d_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = (size_t)400 << 20;                       // 400 MB = 4 M rows of 12 doubles
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void* d; CK(hipMalloc(&d, n));
    char* pg = (char*)malloc(n); memset(pg, 1, n);            // the caller's array: pageable, initialised
    void* hp; CK(hipHostMalloc(&hp, n, hipHostMallocDefault));
    for (int r = 0; r < 3; ++r) {
        double t = now(); memcpy(hp, pg, n); double t1 = now();
        CK(hipMemcpyAsync(d, hp, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        printf("memcpy into pinned %.1f GB/s, then H2D pinned %.1f GB/s: together %.1f GB/s\n", n / (t1 - t) / 1e9, n / (now() - t1) / 1e9, n / (now() - t) / 1e9);
    }
    for (int r = 0; r < 3; ++r) {
        double t = now(); CK(hipMemcpyAsync(d, pg, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        printf("H2D straight from pageable: %.1f GB/s\n", n / (now() - t) / 1e9);
    }
    for (int T : {2, 4, 8}) {
        const size_t chunk = (size_t)16 << 20;
        double t = now();
        std::vector<std::thread> th;
        // T threads, each staging its own interleaved chunks through its own slot of the pinned buffer, copies on one stream
        for (int k = 0; k < T; ++k) th.emplace_back([&, k]() {
            hipStream_t sk; (void)hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
            char* slot = (char*)hp + (size_t)k * chunk;
            for (size_t off = (size_t)k * chunk; off < n; off += (size_t)T * chunk) {
                const size_t len = off + chunk <= n ? chunk : n - off;
                memcpy(slot, pg + off, len);
                (void)hipMemcpyAsync((char*)d + off, slot, len, hipMemcpyHostToDevice, sk);
                (void)hipStreamSynchronize(sk);
            }
            (void)hipStreamDestroy(sk);
        });
        for (auto& x : th) x.join();
        printf("threaded staging, %d threads x 16 MB slots: %.1f GB/s\n", T, n / (now() - t) / 1e9);
    }
    return 0;
}
