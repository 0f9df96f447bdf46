// gf_unitarity.hip -- arbitration of the reference's unitarity assert (golemflavor/fr.py:461-499, raised from
// params_to_BSMu at fr.py:398-399) for the walkers whose verdict the evaluation kernels could not settle.
//
// The assert compares the rounding noise of an x87 (64-bit significand) evaluation of the closed-form eigenvectors
// with 1e-7.  The evaluation kernels (gf_bsm.hip) estimate that noise from an fp64 evaluation and queue the walkers that have
// energy bins whose estimate lies within ~2.7 decades of the threshold; here those bins are re-evaluated exactly as the
// reference does it -- same operations, same order, every result rounded to a 64-bit significand (gf_x87.hpp) -- and the
// verdict is the reference's: residual >= 1e-7 -> GF_ST_NON_UNITARY.
//
// Walker-centric since round 3.  Measured where a posterior crosses the failing region (tools/arb_probe.py,
// profiles/r03/arbitration_queue_census.txt: 12-column posterior, texture OEU, logLam over its whole range): 26 % of the
// walkers are queued with 8.5 undecided bins each on average; 61 % of them end up non-unitary, and THOSE bring 11 bins each --
// 80 % of all (walker, bin) pairs belong to walkers for which one failing bin settles everything.  So a lane takes one
// walker, builds the walker's part of the chain once (mixing matrices, the two Hamiltonian terms: ~40 % of a pair's cost in
// the pair-per-lane kernel of round 2), and evaluates its bins from the highest energy down (the residual grows with the
// energy: the likeliest to fail comes first), stopping at the first failure: ~0.6 M bin evaluations and 0.28 M walker
// set-ups per million walkers instead of 2.3 M of each.  Verdicts are unchanged: a walker fails iff one of its bins does, and
// a bin's residual does not depend on the order.
//
// Per-model constants (the NP mixing matrix of a fixed texture, the SM matrix when its angles are not sampled) were
// computed in long double by gf_model_create with the reference's own libm calls; per-walker matrices are built here
// with the emulated asin / acos / sin / cos.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "gf_consts.h"

extern "C" const char* gf_internal_env(const char* name, int affects_results);   // gf_capi.hip: getenv with a record
#include "gf_launch.h"
#include "gf_x87.hpp"

namespace {
using namespace gfx87;

constexpr int TEX_NONE = 4;
constexpr int ST_NON_UNITARY = 2;
constexpr int UNI_BLOCK = 128;
#ifndef GF_UNI_BLOCKS_PER_CU
#define GF_UNI_BLOCKS_PER_CU 8
#endif

__device__ inline double row_value(const double* __restrict__ theta, int layout, int64_t n, int ndim, int64_t i, int col)
{
    return layout == 0 ? theta[i * ndim + col] : theta[(int64_t)col * n + i];
}

__device__ inline void load_matrix(const double* hi, const double* lo, cx87 u[3][3])
{
    for (int k = 0; k < 9; ++k) {
        const x87 re = {hi[2 * k], lo[2 * k]}, im = {hi[2 * k + 1], lo[2 * k + 1]};
        u[k / 3][k % 3] = c_make(re, im);
    }
}

// The per-walker part of fr.py:380-399 in the reference's arithmetic: the two Hamiltonian terms before their energy factors,
// hsm = U diag(0, m21, m3x) U^+ (fr.py:383-386) and hnp = U~ diag(0, sc1, sc2) U~^+ (fr.py:380-393).
__device__ __attribute__((noinline)) void walker_terms(const GfCommon& c, const GfBsm& tb, const double* __restrict__ theta, int layout,
                                                        int64_t n, int64_t i, cx87 hsm[3][3], cx87 hnp[3][3])
{
    const int ndim = c.ndim;
    cx87 u[3][3];
    if (c.idx_sm[0] >= 0) {                                             // fr.py:425-431: all six from theta, or none
        double ang[4];
        for (int q = 0; q < 4; ++q) ang[q] = row_value(theta, layout, n, ndim, i, c.idx_sm[q]);
        angles_to_u(ang, u);
    } else {
        load_matrix(tb.smu_hi, tb.smu_lo, u);                           // fr.py:435 NUFIT_U (or the fixed angles)
    }
    const double m21 = c.idx_mass[0] >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_mass[0]) : c.mass_fixed[0];
    const double m3x = c.idx_mass[1] >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_mass[1]) : c.mass_fixed[1];
    sandwich(u, m21, m3x, hsm);                                         // fr.py:383-386 (before the 1/2E factor)
    if (tb.texture == TEX_NONE && c.idx_mm[0] >= 0) {                   // fr.py:378, 390
        double ang[4];
        for (int q = 0; q < 4; ++q) ang[q] = row_value(theta, layout, n, ndim, i, c.idx_mm[q]);
        angles_to_u(ang, u);
    } else {
        load_matrix(tb.npu_hi, tb.npu_lo, u);
    }
    const double ll = c.idx_scale >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_scale) : c.scale_fixed;
    const double sc2 = cr_pow10(ll);                                    // fr.py:380 np.power(10., sc2), fp64, correctly rounded
    const double sc1 = sc2 / 100.0;                                     // fr.py:381
    sandwich(u, sc1, sc2, hnp);                                         // fr.py:391-394 (before the E^(d-3) factor)
}

__device__ __attribute__((noinline)) double walker_bin_residual(const cx87 hsm[3][3], const cx87 hnp[3][3], double pre, double epow)
{
    return bin_residual(hsm, hnp, pre, epow);
}

// One lane = one walker at a time.  A lane fetches a walker from the queue (one atomic per wave and round, shared out by
// rank among the lanes that need one), builds its Hamiltonian terms, then takes its undecided bins from the highest energy
// down -- one bin per round of the wave -- until one fails (the walker is non-unitary: fr.py:398-399 raises on the first
// failing energy) or none is left; then it fetches the next walker.  Every wave leaves when the queue is exhausted and all its
// lanes have finished their walker: the exit condition is reached whatever the other waves do.
__global__ __launch_bounds__(UNI_BLOCK) void k_uni_resolve(const GfCommon* __restrict__ cp, const GfBsm* __restrict__ tbp,
                                                           const double* __restrict__ theta, int layout, int64_t n,
                                                           double* __restrict__ lnprob, int32_t* __restrict__ status,
                                                           GfArbQueue* __restrict__ uq, GfUniQueue* __restrict__ wq, unsigned int* __restrict__ seen)
{
    const unsigned int count = uq->count < uq->cap ? uq->count : uq->cap;
    const int lane = threadIdx.x & 63;
    cx87 hsm[3][3], hnp[3][3];
    unsigned long long mask = 0ull;
    int64_t wi = -1;
    bool exhausted = false;                                             // this lane found the queue empty
    for (;;) {
        const bool need = !exhausted && mask == 0ull;
        const unsigned long long nb = __ballot(need);
        if (nb != 0ull) {
            unsigned int base = 0;
            const int leader = __ffsll((long long)nb) - 1;
            if (lane == leader) base = atomicAdd(&uq->head, (unsigned int)__popcll(nb));
            base = (unsigned int)__shfl((int)base, leader);
            if (need) {
                const unsigned int idx = base + (unsigned int)__popcll(nb & ((1ull << lane) - 1ull));
                if (idx < count) {
                    const GfArbItem it = uq->items[idx];
                    wi = (int64_t)it.walker;
                    mask = wi < n ? it.mask : 0ull;
                    if (mask != 0ull) walker_terms(*cp, *tbp, theta, layout, n, wi, hsm, hnp);
                } else {
                    exhausted = true;
                }
            }
        }
        if (__ballot(!exhausted) == 0ull) break;                        // wave-uniform
        if (mask != 0ull) {
            const int k = 63 - __clzll((long long)mask);                // the highest undecided energy first: the likeliest to fail
            mask &= ~(1ull << k);
            const double r = walker_bin_residual(hsm, hnp, tbp->inv2e[k], tbp->epow[k]);
            if (!(r < 1e-7)) {                                          // fr.py:493-494 (NaN raises too)
                status[wi] = ST_NON_UNITARY;
                if (lnprob) lnprob[wi] = __longlong_as_double(0x7ff8000000000000LL);
                mask = 0ull;                                            // the reference has raised: the other bins never run
            }
        }
    }
    // the last block to finish re-arms the queue for the next launch on this stream
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&uq->done, 1u) == gridDim.x - 1) {
            // a producer found one of the queues full and dropped an item: the verdicts of this launch are incomplete
            const bool ov = uq->overflow != 0u || uq->count > uq->cap || (wq && (wq->overflow != 0u || wq->count > wq->cap));
            uq->count = 0;
            uq->done = 0;
            uq->head = 0;
            uq->overflow = 0;
            if (wq) { wq->count = 0; wq->overflow = 0; }                // k_bsm_tier2's walker queue: it ran before this kernel
            __threadfence();
            // what this launch found, in host memory: sizes the grid of the next one (gf_launch_uni_resolve); seen[2]: sticky
            // overflow report, read and cleared by the host (gf_capi.hip: check_queue_overflow)
            if (seen) {
                __atomic_store_n(seen, count, __ATOMIC_RELAXED);
                if (ov) __atomic_store_n(seen + 2, 1u, __ATOMIC_RELAXED);
                seen[3] += count;                                   // running totals (diagnostics: gf_internal_uni_stats):
                seen[4] += 1u;                                      // walkers arbitrated, launches
                __threadfence_system();
            }
        }
    }
}

}  // namespace

// The grid.  The chain keeps its 3x3 complex matrices in scratch (~2 KB per lane): a grid that fills the GPU asks the
// runtime for ~0.5 GB of scratch, more than a queue retains, so that EVERY launch would pay an allocation (~30 us
// measured on an empty queue, profiles/r02) -- as much as the evaluation of 130 000 walkers.  The queue is empty or
// nearly so wherever the posterior lives, so the grid follows what the previous launch on this model found (`seen`, a
// word of pinned host memory the kernel's last block writes; read here without synchronisation, stale is fine: any
// grid is correct, the lanes fetch from the queue until it is empty): a fraction of the GPU while the queue stays short, the whole GPU in the failing region.
hipError_t gf_launch_uni_resolve(const GfCommon* d_common, const GfBsm* d_bsm, const double* theta, int layout, int64_t n, int ndim,
                                 double* lnprob, int32_t* status, GfArbQueue* uq, GfUniQueue* wq, int64_t max_items, unsigned int* seen, int cus, hipStream_t s)
{
    (void)ndim;
    int64_t expect = max_items;
    if (seen && seen[1] == 0) {               // seen[1] (host only): the caller asked for full grids (gf_internal_full_arbitration_grids)
        const int64_t last = (int64_t)__atomic_load_n(seen, __ATOMIC_RELAXED);     // 0xffffffff: nothing seen yet
        expect = 2 * last < max_items ? 2 * last : max_items;
    }
    // lanes fetch their walkers dynamically, several each: half as many lanes as walkers expected, at most GF_UNI_BLOCKS_PER_CU
    // resident blocks per CU
    int64_t blocks = (expect / 2 + UNI_BLOCK - 1) / UNI_BLOCK;
    const int64_t cap = (int64_t)cus * GF_UNI_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
    // A floor under the grid: a queue that fills up unannounced -- the first batch of a scan that enters the failing region --
    // is worked off by whatever grid the hint gave.  Launch cost grows with the grid even where the queue keeps the scratch
    // (an empty queue: ~5 us up to 128 blocks, 8 at 256, 13 at 512, 35 at 2048), so small batches, where those microseconds
    // are the call, get 64 blocks, and batches whose evaluation takes longer than that anyway one block per CU.
    const int64_t floor_blocks = max_items >= 50000 ? cus : 64;
    if (blocks < floor_blocks) blocks = floor_blocks;
    static const int forced = [] { const char* e = gf_internal_env("GF_UNI_RESOLVE_BLOCKS", 0); return e ? std::atoi(e) : 0; }();   // diagnostics / A-B
    if (forced > 0) blocks = forced;
    hipLaunchKernelGGL(k_uni_resolve, dim3((unsigned)blocks), dim3(UNI_BLOCK), 0, s, d_common, d_bsm, theta, layout, n, lnprob, status, uq, wq, seen);
    return hipGetLastError();
}
