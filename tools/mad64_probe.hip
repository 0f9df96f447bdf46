// Issue cost of the integer instructions Philox4x32 is made of, on gfx950: one wave-instruction of each kind costs how many SIMD cycles?
// Every wave runs ITER x 8 independent instructions of one kind; 8 waves per SIMD; cycles from s_memtime around the loop.
//   hipcc -O3 --offload-arch=gfx950 tools/mad64_probe.hip -o tools/mad64_probe && tools/mad64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 4096;
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed)
{
    uint32_t x[8];
    uint64_t acc[8];
    for (int j = 0; j < 8; ++j) { x[j] = seed + threadIdx.x * 8 + j; acc[j] = x[j]; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) { acc[j] = (uint64_t)0xD2511F53u * (uint32_t)acc[j] + (acc[j] >> 32); }            // v_mad_u64_u32
            else if (KIND == 1) { x[j] = __builtin_amdgcn_bitop3_b32(x[j], (uint32_t)i, seed, 0x96); }        // v_bitop3_b32
            else if (KIND == 2) { x[j] = __umulhi(x[j], 0xCD9E8D57u) ^ (uint32_t)i; }                          // v_mul_hi_u32 (+ xor)
            else { x[j] = x[j] * 0xCD9E8D57u + (uint32_t)i; }                                                 // v_mul_lo_u32 / v_mad_u32
        }
    }
    uint32_t r = 0;
    for (int j = 0; j < 8; ++j) r ^= x[j] ^ (uint32_t)acc[j] ^ (uint32_t)(acc[j] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND> void run(const char* name, uint32_t* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;                  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double wave_instr = (double)blocks * 4 * ITER * 8;           // of the kind under test
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;               // at 2.4 GHz, 1024 SIMDs
    printf("%-28s %.3f ms  -> %.2f SIMD cycles per wave-instruction (upper bound: loop overhead and the xor / add that ride along included)\n",
           name, ms, simd_cycles / wave_instr);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_mad_u64_u32", d); run<1>("v_bitop3_b32", d); run<2>("v_mul_hi_u32 + v_xor", d); run<3>("v_mul_lo_u32 + add (v_mad)", d);
    return 0;
}
