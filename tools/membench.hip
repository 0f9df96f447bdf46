// Ceiling probe for the lnprob traffic mix: read R bytes/elem, write W bytes/elem, streaming.
// hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench && ./tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

// variant A: every lane reads three 16-B vectors (lane-contiguous, 3 KiB per wave = a 64x6 fp64 tile),
// reduces them to one double, writes 8 B per lane.  NT selects nontemporal loads/stores.
template <bool NT>
__global__ __launch_bounds__(256) void k_tile(const d2* __restrict__ in, double* __restrict__ out, long ntiles)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long stride = (long)gridDim.x * 4;
    for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += stride) {
        const d2* src = in + t * 192;
        d2 a, b, c;
        if (NT) { a = __builtin_nontemporal_load(src + lane); b = __builtin_nontemporal_load(src + 64 + lane); c = __builtin_nontemporal_load(src + 128 + lane); }
        else { a = src[lane]; b = src[64 + lane]; c = src[128 + lane]; }
        const double v = (a.x + a.y) + (b.x + b.y) + (c.x + c.y);
        if (NT) __builtin_nontemporal_store(v, out + t * 64 + lane); else out[t * 64 + lane] = v;
    }
}
// variant B: plain copy, 16 B per lane in and out
__global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ in, d2* __restrict__ out, long n)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
// variant C: read-only sum (16 B per lane), one store per block
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ in, double* __restrict__ out, long n)
{
    const long stride = (long)gridDim.x * 256;
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { d2 v = in[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[blockIdx.x] = acc;
}
template <class F> float timeit(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main()
{
    const long nw = 4096L * 4096L;           // walkers
    const long ntiles = nw / 64;
    d2* in; double* out; d2* out2;
    hipMalloc(&in, nw * 48); hipMalloc(&out, nw * 8); hipMalloc(&out2, nw * 48);
    hipMemset(in, 0, nw * 48);
    for (int bpc : {4, 8, 16, 32}) {
        const int grid = 256 * bpc;
        float a = timeit([&] { hipLaunchKernelGGL(k_tile<false>, dim3(grid), dim3(256), 0, 0, in, out, ntiles); }, 50);
        float b = timeit([&] { hipLaunchKernelGGL(k_tile<true>, dim3(grid), dim3(256), 0, 0, in, out, ntiles); }, 50);
        printf("tile  blocks/CU %2d: plain %.1f us %.0f GB/s | nt %.1f us %.0f GB/s\n", bpc, a * 1e3, nw * 56 / a / 1e6, b * 1e3, nw * 56 / b / 1e6);
    }
    for (int bpc : {8, 16, 32}) {
        const int grid = 256 * bpc;
        float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, in, out2, nw * 3); }, 20);
        float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, in, out, nw * 3); }, 20);
        printf("copy  blocks/CU %2d: %.1f us %.0f GB/s (r+w) | read-only %.1f us %.0f GB/s\n", bpc, c * 1e3, nw * 96 / c / 1e6, r * 1e3, nw * 48 / r / 1e6);
    }
    return 0;
}
