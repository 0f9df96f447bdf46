// gf_sm_experiments.hpp -- two alternative shapes of the hot lnprob kernel that were built, verified against the
// parity suite and measured on MI355X, and that lost to k_lnprob_sm_fast (gf_kernels.hip).  Compiled only on request
// (tools/build_variants.sh "pipe:-DGF_ASM_PIPE" / "ring:-DGF_EXPERIMENTAL_RING"); included by gf_kernels.hip inside
// its anonymous namespace, after k_lnprob_sm_fast.
//   GF_ASM_PIPE          two tiles per wave in flight with hand-written s_waitcnt   profiles/r01/ab_asm_pipe_depth2.txt
//   GF_EXPERIMENTAL_RING loader wave + LDS-DMA ring + consumer waves                profiles/r01/ring_ab.log
#pragma once

#ifdef GF_ASM_PIPE
// EXPERIMENT (-DGF_ASM_PIPE): the same kernel with TWO tiles of a wave in flight.  In the compiler-scheduled
// loop above a pending lnprob store makes every wait an s_waitcnt vmcnt(0) (loads and stores share the counter
// and stores may retire out of order), so a second prefetched tile is always waited for as well and depth 2
// degenerates to depth 1 (ISA of -DGF_PREFETCH_DEPTH=2).  Here the loads are issued by inline assembly, the
// compiler does not track them, and the wait is written by hand: loads retire in order among themselves, so
// with (PD-1)*VPL loads younger than the tile wanted, vmcnt((PD-1)*VPL) is safe whatever the stores do.
template <int NDIM, int MODE, int SAMPLED>
__global__ __launch_bounds__(GF_BLOCK, GF_SM_WAVES_PER_EU) void k_lnprob_sm_pipe(const GfCommon c, const double* __restrict__ ptab,
                                                              const double* __restrict__ theta, int64_t nfull,
                                                              double* __restrict__ lnprob, int32_t* __restrict__ status)
{
    constexpr int NV = GF_WAVE * NDIM / 2;
    static_assert(NV % GF_WAVE == 0 && NV / GF_WAVE <= 4, "whole 16-B vectors per lane, immediate offsets <= 3072");
    constexpr int VPL = NV / GF_WAVE;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double tiles[GF_WAVES_PER_BLOCK][GF_WAVE * NDIM];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4];
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / GF_WAVE);
    double* tile = tiles[wave];
    const int stride = gridDim.x * GF_WAVES_PER_BLOCK;
    int64_t t = (int64_t)blockIdx.x * GF_WAVES_PER_BLOCK + wave;
    if (t >= nfull) return;

    d2_t A[VPL], B[VPL];
#define GF_PIPE_LOAD(BUF, TILE)                                                                                   \
    do {                                                                                                          \
        const d2_t* src_ = reinterpret_cast<const d2_t*>(theta + (TILE) * (GF_WAVE * NDIM)) + lane;               \
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(BUF[0]) : "v"(src_));                          \
        if (VPL > 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=&v"(BUF[VPL > 1 ? 1 : 0]) : "v"(src_)); \
        if (VPL > 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048 nt" : "=&v"(BUF[VPL > 2 ? 2 : 0]) : "v"(src_)); \
        if (VPL > 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072 nt" : "=&v"(BUF[VPL > 3 ? 3 : 0]) : "v"(src_)); \
    } while (0)
#define GF_PIPE_WAIT(BUF)                                                                                         \
    do {                                                                                                          \
        if (VPL == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(BUF[0]));                                          \
        if (VPL == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(BUF[0]), "+v"(BUF[VPL > 1 ? 1 : 0]));             \
        if (VPL == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(BUF[0]), "+v"(BUF[VPL > 1 ? 1 : 0]), "+v"(BUF[VPL > 2 ? 2 : 0])); \
        if (VPL == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(BUF[0]), "+v"(BUF[VPL > 1 ? 1 : 0]), "+v"(BUF[VPL > 2 ? 2 : 0]), "+v"(BUF[VPL > 3 ? 3 : 0])); \
    } while (0)
    double val_prev = 0.0;
    int64_t i_prev = -1;
#define GF_PIPE_PHASE(BUF)                                                                                        \
    do {                                                                                                          \
        GF_PIPE_WAIT(BUF);                                                                                        \
        _Pragma("unroll") for (int j = 0; j < VPL; ++j) reinterpret_cast<d2_t*>(tile)[j * GF_WAVE + lane] = BUF[j]; \
        if (i_prev >= 0) GF_STORE_OUT(lnprob + i_prev, val_prev);                                                 \
        {                                                                                                         \
            const int64_t ta_ = t + 2 * (int64_t)stride;                                                          \
            const int64_t tn_ = ta_ < nfull ? ta_ : t;                                                            \
            GF_PIPE_LOAD(BUF, tn_);                                                                               \
        }                                                                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                    \
        __builtin_amdgcn_wave_barrier();                                                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                    \
        double val_, fr_[3];                                                                                      \
        int st_;                                                                                                  \
        eval_walker<NDIM, MODE, SAMPLED, false>(c, ctab, tile + lane * NDIM, NDIM, val_, fr_, st_);               \
        val_prev = val_;                                                                                          \
        i_prev = t * GF_WAVE + lane;                                                                              \
        if (status) status[i_prev] = st_;                                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                    \
        __builtin_amdgcn_wave_barrier();                                                                          \
        t += stride;                                                                                              \
    } while (0)

    GF_PIPE_LOAD(A, t);
    {
        const int64_t t1 = t + stride < nfull ? t + stride : t;
        GF_PIPE_LOAD(B, t1);
    }
    while (true) {
        GF_PIPE_PHASE(A);
        if (t >= nfull) break;
        GF_PIPE_PHASE(B);
        if (t >= nfull) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (i_prev >= 0) GF_STORE_OUT(lnprob + i_prev, val_prev);
#undef GF_PIPE_LOAD
#undef GF_PIPE_WAIT
#undef GF_PIPE_PHASE
}
#endif  // GF_ASM_PIPE

#ifdef GF_EXPERIMENTAL_RING
// EXPERIMENT, not built by default (-DGF_EXPERIMENTAL_RING, then GF_SM_RING=1 at run time; see
// profiles/r01/ring_ab.log: parity-green but 230 us vs 183 us for k_lnprob_sm_fast, so it is not the product path).
// Loader / consumer variant of the hot kernel.
// One loader wave per 512-thread block streams tiles HBM -> LDS with LDS-DMA (global_load_lds_dwordx4,
// no VGPR staging, RING_DEPTH tiles in flight) into a RING_SLOTS-deep ring; seven consumer waves take
// tiles round-robin, evaluate them straight out of the slot and hand it back.  Slot ownership travels in
// one LDS word per slot (0 = free, k+1 = holds the block's k-th tile).  Every spin is bounded: a broken
// handshake sets *err and the kernel still terminates.
constexpr int RING_SLOTS = 24;
constexpr int RING_DEPTH = 8;
constexpr int RING_SPIN_MAX = 1 << 22;
typedef __attribute__((address_space(3))) char* lds_cptr;
typedef volatile __attribute__((address_space(3))) int* lds_iptr;

template <int NDIM, int MODE, int SAMPLED, bool WANT_FR>
__global__ __launch_bounds__(512, 2) void k_lnprob_sm_ring(const GfCommon c, const double* __restrict__ ptab,
                                                           const double* __restrict__ theta, int64_t nfull,
                                                           double* __restrict__ lnprob, double* __restrict__ fr_out,
                                                           int32_t* __restrict__ status, int* __restrict__ err)
{
    static_assert(NDIM > 0 && (NDIM % 2) == 0, "ring path: whole 1-KiB DMA pieces per tile");
    constexpr int TILE_B = GF_WAVE * NDIM * 8;
    constexpr int NDMA = TILE_B / 1024;
    constexpr int NC = 7;
    // ONE shared object (LDS offset 0): ring | slot flags | prior table
    __shared__ __attribute__((aligned(16))) char smem[RING_SLOTS * TILE_B + 128 + GF_MAX_DIM * 4 * 8];
    lds_iptr full = (lds_iptr)((lds_cptr)smem + RING_SLOTS * TILE_B);
    double* ctab = reinterpret_cast<double*>(smem + RING_SLOTS * TILE_B + 128);
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    if (threadIdx.x < RING_SLOTS) full[threadIdx.x] = 0;
    __syncthreads();

    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / GF_WAVE);
    const int64_t nb = gridDim.x;
    const int64_t kb = (nfull - blockIdx.x + nb - 1) / nb;           // this block's tiles: t = blockIdx.x + k nb
    const unsigned ring_base = (unsigned)(size_t)(lds_cptr)smem;

    if (wave == 0) {
        for (int64_t k = 0; k < kb; ++k) {
            const int slot = (int)(k % RING_SLOTS);
            int spins = 0;
            while (full[slot] != 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > RING_SPIN_MAX) { if (lane == 0) *err = 1; return; }
            }
            const int64_t t = blockIdx.x + k * nb;
            const char* src = reinterpret_cast<const char*>(theta + t * (GF_WAVE * NDIM)) + lane * 16;
            const unsigned dst = ring_base + slot * TILE_B;
#pragma unroll
            for (int j = 0; j < NDMA; ++j) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + j * 1024), "s"(dst + j * 1024) : "memory");
            }
            if (k >= RING_DEPTH - 1) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA * (RING_DEPTH - 1)) : "memory");
                const int64_t kp = k - (RING_DEPTH - 1);
                if (lane == 0) full[(int)(kp % RING_SLOTS)] = (int)(kp + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int64_t kp = (kb > RING_DEPTH - 1 ? kb - (RING_DEPTH - 1) : 0); kp < kb; ++kp)
            if (lane == 0) full[(int)(kp % RING_SLOTS)] = (int)(kp + 1);
    } else {
        const int cidx = wave - 1;
        for (int64_t k = cidx; k < kb; k += NC) {
            const int slot = (int)(k % RING_SLOTS);
            int spins = 0;
            while (full[slot] != (int)(k + 1)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > RING_SPIN_MAX) { if (lane == 0) *err = 2; return; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const double* row = reinterpret_cast<const double*>(smem + slot * TILE_B) + lane * NDIM;
            double val, fr[3];
            int st;
            eval_walker<NDIM, MODE, SAMPLED, WANT_FR>(c, ctab, row, NDIM, val, fr, st);
            // all lanes are done with the slot (val depends on every read): hand it back
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) full[slot] = 0;
            const int64_t i = (blockIdx.x + k * nb) * GF_WAVE + lane;
            GF_STORE_OUT(lnprob + i, val);
            if (WANT_FR) { fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2]; }
            if (status) status[i] = st;
        }
    }
}
#endif  // GF_EXPERIMENTAL_RING
