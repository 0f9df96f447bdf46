// gf_bsm.hip -- the BSM (texture) branch: golemflavor/fr.py:403-458 flux_averaged_BSMu, i.e. for each
// walker, for each of the nbins energy bins: build H = U_SM diag(0,m21,m3x) U_SM^+ / (2E) +
// E^(d-3) U~ diag(0,sc1,sc2) U~^+ (fr.py:380-395), diagonalise it, propagate the source composition
// with the eigenvector moduli (fr.py:502-536), and average over the bins with linear widths
// (fr.py:453-457); then (lnprob) priors + Gaussian likelihood (llh.py:94-130 with the Gaussian
// substitute for gf.get_llh).
//
// Numerics.  The reference diagonalises with a closed form (fr.py:170-237 cardano_eqn) in 80-bit
// arithmetic: eigenvalues from the trigonometric cubic solution, eigenvector k proportional to
// (conj(B)C, AC, AB) with A, B, C differences of products of H entries.  That product form loses
// *relative* accuracy in A, B or C whenever an eigenvector has tiny components -- which every fixed
// texture has by construction (z = 1e-9, fr.py:370) -- and fp64 would miss the 1e-10 parity bar in
// the high-scale corner.  Only |U_ai|^2 enters the propagation, so the values are computed here from
// the eigenvalues alone with the eigenvector-eigenvalue identity
//     |U_ai|^2 = [ (h_aa - l_j)(h_aa - l_k) + sum_{b != a} |h_ab|^2 ] / ((l_i - l_j)(l_i - l_k)),
// (the (a,a) entry of the spectral projector prod_{j != i} (H - l_j)/(l_i - l_j)), which needs only
// absolute accuracy eps*|H| in the eigenvalues.  Measured against 60-digit mpmath over the C4/C5 scan
// ranges this fp64 path is within 1.1e-12 of the exact |U|^2 everywhere, including where the
// reference's own float128 output is off by 3e-9 ("BSM numerics").
//
// Error behaviour.  The reference raises AssertionError when its computed eigenvector matrix X fails
// |tr|XX^+| - 3| < 1e-7 and |sum|XX^+| - 3| < 1e-7 (fr.py:461-499).  In exact arithmetic X is exactly
// unitary: the failure *is* the rounding noise of the (conj(B)C, AC, AB) form in the x87 format.  When a status
// array is requested the kernel evaluates that same form in fp64 -- the same noise, 2^11 times louder -- as an
// estimate: far below the threshold the walker is unitary, far above it GF_ST_NON_UNITARY, and the (walker, bin)
// pairs in between are queued for gf_unitarity.hip, which replays the reference's operations in emulated x87
// arithmetic and decides (gf_x87.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "gf_consts.h"

extern "C" const char* gf_internal_env(const char* name, int affects_results);   // gf_capi.hip: getenv with a record
#include "gf_launch.h"
#include "gf_device.hpp"
#include "gf_bsm_device.hpp"

namespace {
using namespace gfdev;

// waves per SIMD the evaluation kernel is compiled for, by unitarity mode: without the estimate in the loop body the kernel
// needs ~155 VGPRs (three waves fit in the 512-entry file), with tiers 1-2 inline ~210 (two)
#ifndef GF_BSM_WAVES_NONE
#define GF_BSM_WAVES_NONE 3
#endif
#ifndef GF_BSM_WAVES_DEFER
#define GF_BSM_WAVES_DEFER 3
#endif
#ifndef GF_BSM_WAVES_INLINE
#define GF_BSM_WAVES_INLINE 2
#endif
// (the generic-width instances, NDIM = 0, index the row at run time and need ~180: two waves)
#define GF_BSM_WAVES(UM, ND) ((ND) == 0 ? 2 : (UM) == UNI_NONE ? GF_BSM_WAVES_NONE : (UM) == UNI_DEFER ? GF_BSM_WAVES_DEFER : GF_BSM_WAVES_INLINE)
// Queue walker `i` for the arbitration with its undecided bins (bit k of `amb` = energy bin k)
__device__ __forceinline__ void queue_walker(GfArbQueue* __restrict__ uq, int64_t i, unsigned long long amb)
{
    const unsigned int at = atomicAdd(&uq->count, 1u);
    if (at < uq->cap) {
        GfArbItem it;
        it.walker = (unsigned long long)i;
        it.mask = amb;
        uq->items[at] = it;
    } else {
        uq->overflow = 1u;                 // cannot happen while the host cuts batches to fit (gf_launch_bsm); if it ever does,
    }                                      // the dropped walker's verdict is missing and the host call fails (GF_ERR_QUEUE_OVERFLOW)
}

// LPW > 1 (small batches: a host-driven emcee half-ensemble is a few hundred walkers, i.e. a few waves on 1024
// SIMDs): LPW adjacent lanes share a walker and split its energy bins (flux_average); a wave then covers 64 / LPW
// walkers per tile.  Results are bitwise those of LPW = 1.
// UNI_MODE (gf_bsm_device.hpp): UNI_NONE no status; UNI_INLINE tiers 1-2 inside the kernel, as a second phase for the walkers tier 1
// does not clear (small batches);
// UNI_DEFER the evaluation only notes the walkers tier 1 does not clear (`wq`) and k_bsm_tier2 runs tier 2 on those.
template <int NDIM, bool WITH_LLH, int UNI_MODE, int LPW>
__global__ __launch_bounds__(GF_BLOCK, GF_BSM_WAVES(UNI_MODE, NDIM)) void k_bsm(const GfCommon* __restrict__ cp, const GfBsm* __restrict__ tb,
                                                      const double* __restrict__ ptab,
                                                      const double* __restrict__ theta, int layout, int64_t n,
                                                      double* __restrict__ lnprob, double* __restrict__ fr_out,
                                                      int32_t* __restrict__ status, GfArbQueue* __restrict__ uq, GfUniQueue* __restrict__ wq,
                                                      double* __restrict__ t2sn)
{
    constexpr bool CHECK_UNI = UNI_MODE == UNI_INLINE;
    // the constants by pointer (the model's device block), not by value: as a 848-B kernel argument the compiler loads
    // every field up front and spills ~100 scalar registers to VGPR lanes around the tile loop (356 v_readlane /
    // v_writelane per tile in the by-value build)
    const GfCommon& c = *cp;
    constexpr int WPT = GF_WAVE / LPW;                                  // walkers per wave tile
    extern __shared__ __attribute__((aligned(16))) double fdyn[];        // LPW > 1: per lane group [nbins][3] + [LPW]
    // PREFETCH (one lane per walker, compile-time row width): the next tile of theta is copied HBM -> LDS by the LDS-DMA path
    // (global_load_lds_dwordx4: no VGPR destination, nothing held across the bin loop) into the other of two tile buffers
    // while this tile is evaluated.  At 12 columns the exposed load latency was 6 % of the kernel (every tile re-reading
    // the first one: 0.577 ms against 0.612), at 7 columns 1 %; the copy recovers 2-3.5 % (0.594 ms; 0.647 against 0.670
    // with status), profiles/r02/ab_bsm_variants.txt.
#ifdef GF_BSM_NO_PREFETCH
    constexpr bool PREFETCH = false;
#else
    constexpr bool PREFETCH = NDIM > 0 && LPW == 1;
#endif
    __shared__ __attribute__((aligned(16))) double tiles[PREFETCH ? 2 : 1][GF_WAVES_PER_BLOCK][GF_WAVE * (NDIM ? NDIM : GF_MAX_DIM)];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4 + 20];
    double* ttab = ctab + GF_MAX_DIM * 4;       // texture projector entries, see flux_average
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 18) {
        // {t1, t2} pairs: re[0], re[4], re[8], re[1], im[1], re[2], im[2], re[5], im[5]
        const int k = threadIdx.x - 64, e = k >> 1;
        const int idx = e == 0 ? 0 : e == 1 ? 4 : e == 2 ? 8 : e <= 4 ? 1 : e <= 6 ? 2 : 5;
        const bool im = e == 4 || e == 6 || e == 8;
        const double* srcp = (k & 1) ? (im ? tb->t2_im : tb->t2_re) : (im ? tb->t1_im : tb->t1_re);
        ttab[k] = srcp[idx];
    }
    __syncthreads();
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / GF_WAVE);   // uniform, and the compiler may know: tile indices and
                                                                              // addresses stay in scalar registers
    const int ndim = NDIM ? NDIM : c.ndim;
    const int sub = LPW > 1 ? lane % LPW : 0;
    double* fgrp = LPW > 1 ? fdyn + (threadIdx.x / LPW) * GF_FGRP_DOUBLES(tb->nbins, LPW) : nullptr;
    double* tile = tiles[0][wave];
    // UNI_DEFER: walkers waiting to be queued for k_bsm_tier2, per wave (a tile adds at most 64 to fewer than 64); indices
    // within this launch's piece of the batch, which holds fewer than 2^32 walkers
    __shared__ unsigned int pend[UNI_MODE == UNI_DEFER ? GF_WAVES_PER_BLOCK : 1][UNI_MODE == UNI_DEFER ? 2 * GF_WAVE : 1];
    int npend = 0;
    const int64_t ntiles = (n + WPT - 1) / WPT;
    const int64_t stride = (int64_t)gridDim.x * GF_WAVES_PER_BLOCK;
    bool staged = false;                                                 // PREFETCH: `tile` already holds this tile
    for (int64_t t = (int64_t)blockIdx.x * GF_WAVES_PER_BLOCK + wave; t < ntiles; t += stride) {
        const int64_t w0 = t * WPT;
        const int64_t nend = (LPW > 1 && w0 + WPT < n) ? w0 + WPT : n;     // stage this tile's WPT rows only
        // PREFETCH: does this wave have a next tile, and is it a whole one of an AoS batch?  (64 x NDIM doubles, contiguous in
        // memory and in LDS: wave-instruction j moves the 16-B vectors j * 64 + lane, LDS address = uniform base + 16 * lane)
        bool prefetched = false;
        double* tile_next = tile;
        // (the lane through an empty asm: otherwise per-lane addresses and indices -- lnprob + lane, the staging offsets ... --
        // are formed once ahead of the tile loop and held, or spilled, through all of it)
        int lane_t = lane;
        if (PREFETCH) asm volatile("" : "+v"(lane_t));
        if (PREFETCH) {
            if (!staged) stage_theta<NDIM>(theta, layout, n, w0, ndim, tile, lane_t, nend);
            tile_next = tiles[tile == tiles[0][wave] ? 1 : 0][wave];
            prefetched = layout == 0 && (t + stride + 1) * GF_WAVE <= n;
        } else {
            stage_theta<NDIM>(theta, layout, n, w0, ndim, tile, lane, nend);
        }
        auto start_next_tile = [&]() {
            if (PREFETCH && prefetched) {
                // (the offset through an empty asm: otherwise the address is formed at the top of the tile and carried, or
                // spilled, through the prologue, which is where the kernel's register peak is)
                int zero = 0, ln = lane;
                asm volatile("" : "+s"(zero), "+v"(ln));
                const double* src = theta + (t + stride + zero) * GF_WAVE * (NDIM ? NDIM : 1);
                constexpr int NVEC = GF_WAVE * (NDIM ? NDIM : 2) / 2;
#pragma unroll
                for (int j = 0; j < (NVEC + GF_WAVE - 1) / GF_WAVE; ++j) {
                    const int v = j * GF_WAVE + ln;
                    if (v < NVEC)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * v),
                                                         (__attribute__((address_space(3))) void*)(tile_next + 2 * j * GF_WAVE), 16, 0, 0);
                }
            }
        };
        const int64_t i = w0 + lane_t / LPW;
        bool defer = false;
        if (i < n) {
            const double* row = tile + (lane / LPW) * ndim;
            double lp = 0.0;
            bool inbox = true;
            if (WITH_LLH) {
                inbox = lnprior_tab<NDIM>(ctab, row, ndim, c.prior_const, lp);
#ifndef GF_PRIOR_SINK
                // pin the prior sum here: `lp` is next needed after the bin loop, and left alone the compiler sinks its
                // arithmetic there, carrying x, loc and 1/sigma of every column through the loop (72 VGPRs at 12 columns:
                // 225 instead of 153, the difference between two and three waves per SIMD)
                asm volatile("" : "+v"(lp));
#endif
            }
            double fr[3] = {gf_nan(), gf_nan(), gf_nan()};
            double val = -gf_inf();
            int st = ST_OUT_OF_PRIOR;
            // a tile that prefetches takes every lane through the evaluation (the copy is started from inside it and must be
            // started by the whole wave); the results of walkers outside the prior box are discarded below
            if (inbox || (PREFETCH && prefetched)) {
                UniAcc acc = {0.0, 0.0, 0ull, 2.0};
                flux_average<UNI_MODE, LPW>(c, tb, ttab, row, fr, acc, sub, fgrp, UNI_MODE == UNI_DEFER ? t2sn + i * GF_SN_DOUBLES : nullptr,
                                            start_next_tile);
                st = ST_OK;
                if (CHECK_UNI) {
                    if (tb->uni_lo < 0.0) fr[0] = acc.est_max * (1.0 / UNI_EST_SCALE);  // diagnostics (GF_UNI_DUMP): the estimate itself
                    if (!(acc.clear_max < tb->uni_hi)) st = ST_NON_UNITARY;
                    // undecided bins: the x87-faithful evaluation settles them (gf_unitarity.hip); until then the
                    // walker counts as unitary
                    else if (inbox && acc.amb != 0 && sub == 0 && uq) queue_walker(uq, i, uni_arbitration_mask(acc.amb, tb));
                }
                if (UNI_MODE == UNI_DEFER) defer = inbox && acc.a_min < tb->uni_a_ok;      // tier 1 does not clear this walker
                if (WITH_LLH) {
                    // llh.py:109-112: fr -> fr_to_angles -> (Gaussian substitute) angles_to_fr is the
                    // identity on a normalised composition up to rounding
                    val = lp + gauss_llh(c, fr);
                    if (val != val && st == ST_OK) st = ST_NAN;
                } else if (fr[0] != fr[0] || fr[1] != fr[1] || fr[2] != fr[2]) {
                    if (st == ST_OK) st = ST_NAN;
                }
                if (st == ST_NON_UNITARY) val = gf_nan();              // the reference raises here
                if (!inbox) {                                          // evaluated for the wave's sake only
                    st = ST_OUT_OF_PRIOR;
                    val = -gf_inf();
                    fr[0] = fr[1] = fr[2] = gf_nan();
                }
            }
            // the next tile's DMA has had the whole bin loop to land: retire it here, AHEAD of this tile's result stores (vmcnt
            // counts those too, and they are then left to drain behind the next tile's arithmetic).  Every tile has a walker
            // with i < n, so every wave passes here.
            if (PREFETCH && prefetched) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (sub == 0) {
                if (WITH_LLH) lnprob[i] = val;
                if (fr_out) { fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2]; }
                if (status) status[i] = st;
            }
        }
        if (UNI_MODE == UNI_DEFER) {
            // The walkers to queue collect in the wave's LDS list and leave for the global queue 64 or more at a time:
            // one atomic per ~64 queued walkers.  (One per wave and tile -- 65 536 returning atomics on one address for
            // 4 M walkers -- cost the evaluation kernel 18 %, profiles/r02/bsm_defer_atomics.txt.)
            const unsigned long long m = __ballot(defer);
            if (m != 0) {
                if (defer) pend[wave][npend + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)i;
                npend += __popcll(m);                                    // wave-uniform
                if (npend >= GF_WAVE) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    unsigned int base = 0;
                    if (lane == 0) base = atomicAdd(&wq->count, (unsigned int)npend);
                    base = (unsigned int)__shfl((int)base, 0);
                    for (int j = lane; j < npend; j += GF_WAVE) {
                        if (base + j < wq->cap) wq->items[base + j] = (unsigned long long)pend[wave][j];
                        else wq->overflow = 1u;
                    }
                    npend = 0;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (PREFETCH) {
            staged = prefetched;
            tile = tile_next;
        }
    }
    if (UNI_MODE == UNI_DEFER) {
        // what is left in the four lists goes out with one atomic per block
        __shared__ unsigned int blk_off[GF_WAVES_PER_BLOCK + 1];
        __shared__ unsigned int blk_base;
        if (lane == 0) blk_off[wave + 1] = (unsigned int)npend;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            blk_off[0] = 0;
            for (int w = 0; w < GF_WAVES_PER_BLOCK; ++w) { tot += blk_off[w + 1]; blk_off[w + 1] = tot; }
            blk_base = tot ? atomicAdd(&wq->count, tot) : 0u;
        }
        __syncthreads();
        const unsigned int base = blk_base + blk_off[wave];
        for (int j = lane; j < npend; j += GF_WAVE) {
            if (base + j < wq->cap) wq->items[base + j] = (unsigned long long)pend[wave][j];
            else wq->overflow = 1u;
        }
    }
}

// Tier 2 for the walkers the evaluation kernel queued (UNI_DEFER): the fp64 estimate of the bins tier 1 does not clear,
// one lane per walker, full waves of them -- instead of every wave of the evaluation running tier 2 because one of its
// 64 walkers needs it.  The walker's Hamiltonian terms come from `t2sn` (left there by the evaluation kernel), so there
// is no prologue to repeat.  Same classification as the inline path: condemned walkers get NON_UNITARY (and a NaN
// value), undecided (walker, bin) pairs go on to the arbitration queue.
#ifndef GF_T2_WAVES
#define GF_T2_WAVES 2
#endif
__global__ __launch_bounds__(GF_BLOCK, GF_T2_WAVES) void k_bsm_tier2(const GfBsm* __restrict__ tb, const double* __restrict__ t2sn, int64_t n,
                                                                     double* __restrict__ lnprob, int32_t* __restrict__ status,
                                                                     GfArbQueue* __restrict__ uq, GfUniQueue* __restrict__ wq)
{
    const unsigned int count = wq->count < wq->cap ? wq->count : wq->cap;
    for (unsigned int q = blockIdx.x * GF_BLOCK + threadIdx.x; q < count; q += gridDim.x * GF_BLOCK) {
        const int64_t i = (int64_t)wq->items[q];
        if (i >= n) continue;
        Herm3 S, N;
        load_sn(t2sn + i * GF_SN_DOUBLES, S, N);
        UniAcc acc = {0.0, 0.0, 0ull, 2.0};
        tier2_from_sn(tb, S, N, acc);
        if (!(acc.clear_max < tb->uni_hi)) {
            status[i] = ST_NON_UNITARY;
            if (lnprob) lnprob[i] = gf_nan();
        } else if (acc.amb != 0) {
            queue_walker(uq, i, uni_arbitration_mask(acc.amb, tb));
        }
    }
    // the walker queue is re-armed by k_uni_resolve, which follows in stream order (one store there instead of a fence
    // and an atomic per block here)
}

inline int grid_for(int64_t work_items, int per_block, int cus)
{
    int64_t blocks = (work_items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)cus * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// Lanes per walker for a batch of n walkers, from a sweep on MI355X (profiles/r01/bsm_lanes_per_walker_sweep.txt:
// 12-column posterior, 20 bins; 16 lanes win up to 4096 walkers, 4 lanes up to 32768, one lane beyond): the widest split
// whose waves still fit one (16 lanes) or two (4 lanes) per SIMD -- past that the repeated per-walker prologue costs
// more than the shorter critical path saves.
inline int lanes_for(int64_t n, int nbins, int cus, bool check)
{
    (void)check;
    const char* force = gf_internal_env("GF_BSM_LPW", 0);                       // diagnostics / A-B
    if (force) { const int f = std::atoi(force); if (f == 1 || f == 4 || f == 16) return f; }
    if (nbins < 2) return 1;
    const int64_t simds = 4 * (int64_t)(cus > 0 ? cus : 256);
    const int64_t waves1 = (n + GF_WAVE - 1) / GF_WAVE;
    auto fits = [&](int lpw) { return (size_t)(GF_BLOCK / lpw) * GF_FGRP_DOUBLES(nbins, lpw) * sizeof(double) <= 32 * 1024; };
    if (waves1 * 16 <= simds && fits(16)) return 16;
    if (waves1 * 4 <= 2 * simds && fits(4)) return 4;
    return 1;
}

template <int NDIM, int LPW>
hipError_t launch_nl(const GfCommon& c, const GfCommon* d_common, const GfBsm* d_bsm, int nbins, const double* ptab, const double* theta, int layout,
                     int64_t n, int with_llh, double* lnprob, double* fr, int32_t* status, GfArbQueue* uq, GfUniQueue* wq, double* t2sn, int cus, hipStream_t s)
{
    const int grid = grid_for(n * LPW, GF_BLOCK, cus);
    const size_t lds = LPW > 1 ? (size_t)(GF_BLOCK / LPW) * GF_FGRP_DOUBLES(nbins, LPW) * sizeof(double) : 0;
    // status requested: tiers 1-2 inline (wq == NULL) or deferred to k_bsm_tier2 (large batches, one lane per walker)
    const int mode = status == nullptr ? UNI_NONE : (wq != nullptr && LPW == 1 ? UNI_DEFER : UNI_INLINE);
#define GF_GO(WL, UM) hipLaunchKernelGGL((k_bsm<NDIM, WL, UM, LPW>), dim3(grid), dim3(GF_BLOCK), lds, s, d_common, d_bsm, ptab, theta, layout, n, lnprob, fr, status, uq, wq, t2sn)
    if (with_llh) { if (mode == UNI_NONE) GF_GO(true, UNI_NONE); else if (mode == UNI_INLINE) GF_GO(true, UNI_INLINE); else { if constexpr (LPW == 1) GF_GO(true, UNI_DEFER); } }
    else          { if (mode == UNI_NONE) GF_GO(false, UNI_NONE); else if (mode == UNI_INLINE) GF_GO(false, UNI_INLINE); else { if constexpr (LPW == 1) GF_GO(false, UNI_DEFER); } }
#undef GF_GO
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && mode == UNI_DEFER) {
        const int g2 = grid_for(n / 4 + 1, GF_BLOCK, cus);
        hipLaunchKernelGGL(k_bsm_tier2, dim3(g2), dim3(GF_BLOCK), 0, s, d_bsm, t2sn, n, with_llh ? lnprob : nullptr, status, uq, wq);
        e = hipGetLastError();
    }
    return e;
}

template <int NDIM>
hipError_t launch_n(const GfCommon& c, const GfCommon* d_common, const GfBsm* d_bsm, int nbins, const double* ptab, const double* theta, int layout, int64_t n,
                    int with_llh, double* lnprob, double* fr, int32_t* status, GfArbQueue* uq, GfUniQueue* wq, double* t2sn, int cus, hipStream_t s)
{
    switch (lanes_for(n, nbins, cus, status != nullptr)) {
    case 4: return launch_nl<NDIM, 4>(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, uq, nullptr, nullptr, cus, s);
    case 16: return launch_nl<NDIM, 16>(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, uq, nullptr, nullptr, cus, s);
    default: return launch_nl<NDIM, 1>(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, uq, wq, t2sn, cus, s);
    }
}

}  // namespace

// One launch of the evaluation kernel over walkers [0, n) of the block at `theta` (AoS: rows; SoA: the caller passes the
// whole batch and `n` = its size).  `uq` (status requested): where undecided (walker, bin) pairs go; `qbase` is added to
// the walker index in the queue.
static hipError_t launch_eval(const GfCommon& c, const GfCommon* d_common, const GfBsm* d_bsm, int nbins, const double* ptab, const double* theta,
                              int layout, int64_t n, int with_llh, double* lnprob, double* fr, int32_t* status, GfArbQueue* uq, GfUniQueue* wq,
                              double* t2sn, int cus, hipStream_t s)
{
    switch (c.ndim) {
    case 7: return launch_n<7>(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, uq, wq, t2sn, cus, s);
    case 12: return launch_n<12>(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, uq, wq, t2sn, cus, s);
    default: return launch_n<0>(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, uq, wq, t2sn, cus, s);
    }
}

// `uq` / `uq_cap` (items = walkers): the stream's arbitration queue, NULL / 0 when no status array is requested; `wq`: its queue of
// walkers for k_bsm_tier2 (`wq_cap` walkers) and `t2sn` their Hamiltonian terms ([wq_cap][18]), NULL = tiers inline.  With a status array
// an AoS batch is cut into pieces whose worst case (every walker undecided) fits the queues, each piece
// followed by the resolve kernel: evaluation and arbitration stay in stream order, nothing is read back.
hipError_t gf_launch_bsm(const GfCommon& c, const GfCommon* d_common, const GfBsm* d_bsm, int nbins, const double* ptab, const double* theta, int layout,
                         int64_t n, int with_llh, double* lnprob, double* fr, int32_t* status, GfArbQueue* uq, int64_t uq_cap, GfUniQueue* wq, int64_t wq_cap,
                         double* t2sn, unsigned int* seen, int cus, hipStream_t s)
{
    if (!status || !uq) return launch_eval(c, d_common, d_bsm, nbins, ptab, theta, layout, n, with_llh, lnprob, fr, status, nullptr, nullptr, nullptr, cus, s);
    int64_t piece = uq_cap;
    if (wq && wq_cap < piece) piece = wq_cap;                            // ... and the walker queue of the deferred tier 2
    // diagnostics only (needs GF_DIAGNOSTICS=1): pieces four times what the queues hold, to exercise the overflow report
    static const bool overcommit = gf_internal_env("GF_DIAG_UQ_OVERCOMMIT", 1) != nullptr;
    if (overcommit) piece *= 4;
    if (piece > 64) piece &= ~(int64_t)63;                               // whole tiles: every piece starts 16-B aligned like the batch
    if (piece < 1) piece = 1;
    if (layout != 0 && piece < n) return hipErrorInvalidValue;          // SoA columns cannot be cut: the caller sizes the queue for n
    for (int64_t w0 = 0; w0 < n; w0 += piece) {
        const int64_t m = n - w0 < piece ? n - w0 : piece;
        hipError_t e = launch_eval(c, d_common, d_bsm, nbins, ptab, layout == 0 ? theta + w0 * c.ndim : theta, layout, m, with_llh,
                                   lnprob ? lnprob + w0 : nullptr, fr ? fr + 3 * w0 : nullptr, status + w0, uq, wq, t2sn, cus, s);
        if (e != hipSuccess) return e;
        e = gf_launch_uni_resolve(d_common, d_bsm, layout == 0 ? theta + w0 * c.ndim : theta, layout, m, c.ndim,
                                  with_llh && lnprob ? lnprob + w0 : nullptr, status + w0, uq, wq, m, seen, cus, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
