// d2h_kernel_probe -- device memory to REGISTERED host memory: the runtime's DMA (hipMemcpy: fast from a process's first large allocation,
// 30 GB/s from later ones -- tools/vram_realloc_probe.py) against a copy KERNEL that stores through the host memory's device alias
// (hipHostGetDevicePointer), by grid size.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/d2h_kernel_probe.hip -o tools/d2h_kernel_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double v2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_copy(v2* __restrict__ dst, const v2* __restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const v2 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const size_t n = (size_t)((argc > 1 ? atof(argv[1]) : 9.4) * 1e9) / 4096 * 4096;
    void* h = mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (h == MAP_FAILED) return 2;
    char* hp = (char*)(((uintptr_t)h + (2 << 20) - 1) / (2 << 20) * (2 << 20));
    madvise(hp, n, MADV_HUGEPAGE);
    for (size_t i = 0; i < n; i += 4096) hp[i] = 0;
    CK(hipHostRegister(hp, n, hipHostRegisterPortable | hipHostRegisterMapped));
    void* hdev = nullptr;
    CK(hipHostGetDevicePointer(&hdev, hp, 0));
    printf("host %p device alias %p\n", (void*)hp, hdev);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int it = 0; it < 4; ++it) {
        void* d = nullptr;
        CK(hipMalloc(&d, n));
        CK(hipMemset(d, it + 1, n));
        CK(hipDeviceSynchronize());
        double t0 = now();
        CK(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        const double dma = n / (now() - t0) / 1e9;
        printf("allocation %d: hipMemcpyAsync %.1f GB/s; kernel", it, dma);
        for (int blocks : {32, 64, 128, 256, 1024}) {
            memset(hp, 0, 4096);
            t0 = now();
            hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (v2*)hdev, (const v2*)d, n / 16);
            CK(hipStreamSynchronize(st));
            printf("  %d blocks %.1f", blocks, n / (now() - t0) / 1e9);
            if ((unsigned char)hp[0] != (unsigned char)(it + 1) || (unsigned char)hp[n - 1] != (unsigned char)(it + 1)) printf(" (WRONG)");
        }
        printf(" GB/s\n");
        CK(hipFree(d));
    }
    return 0;
}
