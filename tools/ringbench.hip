// Loader / consumer ring skeleton for the lnprob stream (experiment).
//   one loader wave per block streams 64-walker tiles (3 KiB) HBM -> LDS ring with LDS-DMA
//   (global_load_lds_dwordx4, 3 per tile), NC consumer waves take tiles round-robin, do `work` dependent
//   fp64 FMAs per lane (stand-in for the 174-instruction walker evaluation) and write 8 B per lane.
// Every spin is bounded: a broken handshake sets err[0] and the kernel still terminates.
// hipcc -O3 --offload-arch=gfx950 tools/ringbench.hip -o tools/ringbench && ./tools/ringbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

constexpr int SLOTS = 24;                 // ring slots per block
constexpr int TILE_D = 384;               // doubles per tile (64 walkers x 6)
constexpr int TILE_B = TILE_D * 8;        // 3072 bytes
constexpr int DEPTH = 8;                  // tiles the loader keeps in flight
constexpr int SPIN_MAX = 1 << 22;

typedef __attribute__((address_space(3))) char* lds_cptr;
typedef volatile __attribute__((address_space(3))) int* lds_iptr;   // flags: real ds_read/ds_write, never flat
typedef const __attribute__((address_space(3))) double* lds_dptr;

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_ring(const double* __restrict__ theta, double* __restrict__ out,
                                                       long ntiles, int work, int* __restrict__ err)
{
    constexpr int NC = NWAVES - 1;
    // ONE shared object: ring first (LDS offset 0), then the slot flags
    __shared__ __attribute__((aligned(16))) char smem[SLOTS * TILE_B + SLOTS * 4];
    lds_iptr full = (lds_iptr)((lds_cptr)smem + SLOTS * TILE_B);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < SLOTS) full[threadIdx.x] = 0;
    __syncthreads();
    const long nb = gridDim.x;
    const long kb = (ntiles - blockIdx.x + nb - 1) / nb;          // tiles of this block: t = blockIdx.x + k*nb
    const unsigned ring_base = (unsigned)(size_t)(lds_cptr)smem;  // LDS byte address of the ring

    if (wave == 0) {
        // ---------------- loader ----------------
        for (long k = 0; k < kb; ++k) {
            const int slot = (int)(k % SLOTS);
            int spins = 0;
            while (full[slot] != 0) {                              // slot still owned by a consumer
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX) { if (lane == 0) err[0] = 1; return; }
            }
            const long t = blockIdx.x + k * nb;
            const char* src = reinterpret_cast<const char*>(theta + t * TILE_D) + lane * 16;
            const unsigned dst = ring_base + slot * TILE_B;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + j * 1024), "s"(dst + j * 1024) : "memory");
            }
            if (k >= DEPTH - 1) {
                // all but the youngest DEPTH-1 tiles (3 DMAs each) have landed -> publish tile k-(DEPTH-1)
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (DEPTH - 1)) : "memory");
                const long kp = k - (DEPTH - 1);
                if (lane == 0) full[(int)(kp % SLOTS)] = (int)(kp + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (long kp = (kb > DEPTH - 1 ? kb - (DEPTH - 1) : 0); kp < kb; ++kp)
            if (lane == 0) full[(int)(kp % SLOTS)] = (int)(kp + 1);
    } else {
        // ---------------- consumers ----------------
        const int c = wave - 1;
        for (long k = c; k < kb; k += NC) {
            const int slot = (int)(k % SLOTS);
            int spins = 0;
            while (full[slot] != (int)(k + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_MAX) { if (lane == 0) err[0] = 2; return; }
            }
            lds_dptr row = (lds_dptr)((lds_cptr)smem + slot * TILE_B) + lane * 6;
            double x0 = row[0], x1 = row[1], x2 = row[2], x3 = row[3], x4 = row[4], x5 = row[5];
            // every lane has its row in registers: hand the slot back before the long compute
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) full[slot] = 0;
            double acc = ((x0 + x1) + (x2 + x3)) + (x4 + x5);
            double y = acc * 1e-3;
            for (int i = 0; i < work; ++i) y = fma(y, 0.999999, 1e-9);
            const long t = blockIdx.x + k * nb;
            __builtin_nontemporal_store(acc + (y - y), out + t * 64 + lane);
        }
    }
}

template <class F> float timeit(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const long nw = 4096L * 4096L, ntiles = nw / 64;
    double *in, *out; int* err;
    hipMalloc(&in, nw * 48); hipMalloc(&out, nw * 8); hipMalloc(&err, 16);
    hipMemset(err, 0, 16);
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)(i % 1000) * 1e-3;
    for (long off = 0; off < nw * 6; off += (long)h.size()) hipMemcpy(in + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    // correctness of the handshake: out[i] = sum of row i
    hipLaunchKernelGGL(k_ring<8>, dim3(512), dim3(512), 0, 0, in, out, ntiles, 0, err);
    hipDeviceSynchronize();
    std::vector<double> ho(1 << 16);
    int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long base : {0L, nw / 2, nw - (long)ho.size()}) {
        hipMemcpy(ho.data(), out + base, ho.size() * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < ho.size(); ++i) {
            double want = 0; for (int d = 0; d < 6; ++d) want += h[((base + (long)i) * 6 + d) % (long)h.size()];
            if (std::fabs(ho[i] - want) > 1e-12) ++bad;
        }
    }
    printf("handshake check: err=%d bad=%ld\n", herr, bad);
    if (herr || bad) return 1;
    for (int work : {0, 100, 150, 200}) {
        for (int grid : {256, 512, 768}) {
            float a = timeit([&] { hipLaunchKernelGGL(k_ring<8>, dim3(grid), dim3(512), 0, 0, in, out, ntiles, work, err); }, 20);
            float b = timeit([&] { hipLaunchKernelGGL(k_ring<4>, dim3(grid * 2), dim3(256), 0, 0, in, out, ntiles, work, err); }, 20);
            printf("work %3d grid %4d: 8 waves (1+7) %.1f us %.0f GB/s | 4 waves (1+3) x2 blocks %.1f us %.0f GB/s\n", work, grid,
                   a * 1e3, nw * 56 / a / 1e6, b * 1e3, nw * 56 / b / 1e6);
        }
    }
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("final err=%d\n", herr);
    return 0;
}
