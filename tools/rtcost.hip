// Host-side cost of the runtime calls a model needs (stream, small allocation, small upload).
// hipcc -O2 --offload-arch=gfx950 tools/rtcost.hip -o tools/rtcost && ./tools/rtcost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main()
{
    const int n = 128;
    hipFree(0);
    std::vector<hipStream_t> st(n);
    std::vector<void*> blk(n);
    char host[4096] = {0};
    double t = now();
    for (int i = 0; i < n; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    printf("hipStreamCreateWithFlags: %.1f us each\n", (now() - t) / n * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st[i], nullptr);
    hipDeviceSynchronize();
    printf("first launch on each new stream: %.1f us each\n", (now() - t) / n * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st[i], nullptr);
    hipDeviceSynchronize();
    printf("second launch on each stream: %.1f us each\n", (now() - t) / n * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipMalloc(&blk[i], 3200);
    printf("hipMalloc(3200 B): %.1f us each\n", (now() - t) / n * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipMemcpy(blk[i], host, 3200, hipMemcpyHostToDevice);
    printf("hipMemcpy H2D 3200 B: %.1f us each\n", (now() - t) / n * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipMemcpyAsync(blk[i], host, 3200, hipMemcpyHostToDevice, st[0]);
    hipStreamSynchronize(st[0]);
    printf("hipMemcpyAsync H2D 3200 B (one stream): %.1f us each\n", (now() - t) / n * 1e6);
    hipDeviceProp_t prop;
    t = now();
    for (int i = 0; i < 16; ++i) hipGetDeviceProperties(&prop, 0);
    printf("hipGetDeviceProperties: %.1f us each\n", (now() - t) / 16 * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipFree(blk[i]);
    printf("hipFree: %.1f us each\n", (now() - t) / n * 1e6);
    t = now();
    for (int i = 0; i < n; ++i) hipStreamDestroy(st[i]);
    printf("hipStreamDestroy: %.1f us each\n", (now() - t) / n * 1e6);
    void* big; 
    t = now(); hipMalloc(&big, 3200 * 256); printf("hipMalloc(800 KB slab): %.1f us\n", (now() - t) * 1e6);
    void* pin;
    t = now(); hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault); printf("hipHostMalloc(1 MB): %.1f us\n", (now() - t) * 1e6);
    return 0;
}
