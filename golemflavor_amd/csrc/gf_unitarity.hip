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
#include "gf_unitarity_teams.hpp"

namespace {
using namespace gfx87;


// Three lanes = one walker at a time.  A group fetches a walker from the queue (one atomic per wave and round, shared out by
// rank among the groups that need one), builds its Hamiltonian terms, then takes its undecided bins from the highest energy
// down -- one bin per round of the wave -- until one fails (the walker is non-unitary: fr.py:398-399 raises on the first
// failing energy) or none is left; then it fetches the next walker.  Every wave leaves when the queue is exhausted and all its
// groups have finished their walker: the exit condition is reached whatever the other waves do.
template <class Team>
__global__ __launch_bounds__(UNI_BLOCK, GF_UNI_RESOLVE_WAVES) void k_uni_resolve(const GfCommon* __restrict__ cp, const GfBsm* __restrict__ tbp,
                                                           const double* __restrict__ theta, int layout, int64_t n,
                                                           double* __restrict__ lnprob, int32_t* __restrict__ status,
                                                           GfArbQueue* __restrict__ uq, GfUniQueue* __restrict__ wq, unsigned int* __restrict__ seen)
{
    __shared__ __attribute__((aligned(16))) double lds[(UNI_BLOCK / 64) * Team::PER_WAVE * Team::DOUBLES];
    const unsigned int count = uq->count < uq->cap ? uq->count : uq->cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / Team::LANES;
    const bool active = grp < Team::PER_WAVE;                           // lane 63 has no group
    Team tm;
    tm.init(lds + ((size_t)wave * Team::PER_WAVE + (active ? grp : 0)) * Team::DOUBLES, lane - grp * Team::LANES);
    const bool lead = tm.leader();
    const unsigned int fan = uni_fanout(count, gridDim.x * (UNI_BLOCK / 64) * Team::PER_WAVE);
    const unsigned long long vcount = (unsigned long long)count * fan;  // (walker, part) items; count <= 2^23, fan <= 20
    unsigned long long mask = 0ull;
    int64_t wi = -1;
    bool exhausted = !active;                                           // this group found the queue empty
    for (;;) {
        const bool need = !exhausted && mask == 0ull;
        const unsigned long long nb = __ballot(need && lead);           // one request per group
        if (nb != 0ull) {
            unsigned int base = 0;
            const int leader = __ffsll((long long)nb) - 1;
            if (lane == leader) base = atomicAdd(&uq->head, (unsigned int)__popcll(nb));
            base = (unsigned int)__shfl((int)base, leader);
            if (need) {
                const unsigned int idx = base + (unsigned int)__popcll(nb & ((1ull << (grp * Team::LANES)) - 1ull));
                if (idx < vcount) {
                    const GfArbItem it = uq->items[idx / fan];
                    wi = (int64_t)it.walker;
                    mask = wi < n ? uni_part_mask(it.mask, idx % fan, fan) : 0ull;
                    if (mask != 0ull) tm.terms(*cp, *tbp, theta, layout, n, wi, 0);
                } else {
                    exhausted = true;
                }
            }
        }
        if (__ballot(!exhausted) == 0ull) break;                        // wave-uniform
        if (fan > 1u && mask != 0ull && status[wi] == UT_NON_UNITARY) mask = 0ull;    // another part of this walker has failed already
        if (mask != 0ull) {
            const int k = 63 - __clzll((long long)mask);                // the highest undecided energy first: the likeliest to fail
            mask &= ~(1ull << k);
            const double res = tm.bin(tbp->inv2e[k], tbp->epow[k]);
            if (!(res < 1e-7)) {                                        // fr.py:493-494 (NaN raises too)
                if (lead) {
                    status[wi] = UT_NON_UNITARY;
                    if (lnprob) lnprob[wi] = __longlong_as_double(0x7ff8000000000000LL);
                }
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

// The device sampler's parked proposals (gf_sampler.hip, gf_launch.h GfSettleArgs): the same arbitration (on the nine-lane team:
// the half-step waits for this kernel), and at the end of a walker -- one bin failed, or all of them passed -- the group's leading
// lane completes that walker's half-step exactly
// as the half-step kernel does for the proposals it settles itself: a proposal the reference would have raised on is rejected
// and counted, any other goes through the accept test ln(z^(ndim-1) / u) > lnp(s) - lnp(q); the stored sample is written.
#ifdef GF_SETTLE_TIMING
// diagnostics build (tools/settle_timing.py): where a settle launch's time goes, on the 100 MHz wall clock all CUs share --
// [0] earliest kernel entry, [1] latest end of a walker's terms, [2] latest end of a bin, [3] latest completed walker, [4] latest block exit,
// [5] launches with work; all accumulated as (value - entry) sums over launches in [6..9]
__device__ unsigned long long g_settle_t[12];
__device__ unsigned long long g_settle_c[4];      // shader-clock counts: [0] sum over launches of the longest (terms), [1] of the longest (terms + bin)
__device__ unsigned long long g_settle_cmax[4];
#define GF_ST_MARK(i) do { if (lead) { atomicMax(&g_settle_t[i], (unsigned long long)wall_clock64()); if (i <= 2) atomicMax(&g_settle_cmax[i], (unsigned long long)(clock64() - c_entry)); } } while (0)
#else
#define GF_ST_MARK(i) do { } while (0)
#endif
template <class Team>
__global__ __launch_bounds__(UNI_BLOCK, GF_UNI_WAVES) void k_stretch_settle(const GfSettleArgs s)
{
    __shared__ __attribute__((aligned(16))) double lds[(UNI_BLOCK / 64) * Team::PER_WAVE * Team::DOUBLES];
    GfArbQueue* __restrict__ uq = s.pq;
    const unsigned int count = uq->count < uq->cap ? uq->count : uq->cap;
    // nothing parked -- the usual case wherever a posterior keeps away from the failing region: nothing to settle and nothing to
    // re-arm (count and head are zero already).  No block changes the queue before EVERY block has read the count: the re-arming
    // below is done by the last block to arrive, so all blocks take the same branch here.
    if (count == 0u && uq->overflow == 0u) return;
#ifdef GF_SETTLE_TIMING
    if (threadIdx.x == 0) atomicMin(&g_settle_t[0], (unsigned long long)wall_clock64());
    const long long c_entry = clock64();
#endif
    {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / Team::LANES;
    const bool active = grp < Team::PER_WAVE;
    Team tm;
    tm.init(lds + ((size_t)wave * Team::PER_WAVE + (active ? grp : 0)) * Team::DOUBLES, lane - grp * Team::LANES);
    const bool lead = tm.leader();
    const int ndim = s.ndim, nhalf = s.nwalkers / 2;
    const int64_t nprop = (int64_t)s.nchains * nhalf;
    const int64_t run_step = s.state->run_step_base + s.step_offset;
    const int thin = s.state->thin;
    const bool store_now = s.state->store != 0 && s.chain != nullptr && (run_step % thin) == 0;
    const int64_t store_index = s.state->store_base + (run_step + thin - 1) / thin;
    // A half-step waits for this kernel, and a handful of parked proposals is the usual case: what counts is the critical path
    // of ONE walker, so its bins are spread over as many groups as the grid has to spare (uni_fanout).  The last part of a
    // walker to finish completes its update.
    const unsigned int fan = uni_fanout(count, gridDim.x * (UNI_BLOCK / 64) * Team::PER_WAVE);
    const unsigned long long vcount = (unsigned long long)count * fan;
    unsigned long long mask = 0ull;
    int64_t t = -1;
    int chain = 0;
    unsigned int parts = 1;
    bool exhausted = !active, failed = false, mine = false;             // mine: this group holds a part whose end it must report
    unsigned int warm = 0u;
    // The parts are handed out by position, not by a counter: group G of the grid takes the slots G, G + (groups of the grid), ...
    // The queue is short (its slots rarely outnumber the groups) and a half-step waits for the LAST part, so there is no balance to
    // gain from fetching dynamically -- and the fetch was an atomic with a return on the path every part waits for (~2 us of 47).
    const unsigned long long total_groups = (unsigned long long)gridDim.x * (UNI_BLOCK / 64) * Team::PER_WAVE;
    unsigned long long next = ((unsigned long long)blockIdx.x * (UNI_BLOCK / 64) + (unsigned long long)wave) * Team::PER_WAVE + (unsigned long long)grp;
    for (;;) {
        const bool need = !exhausted && !mine;
        if (__ballot(need) != 0ull) {
            if (need) {
                const unsigned long long idx = next;
                next += total_groups;
                if (idx < vcount) {
                    const GfArbItem it = uq->items[idx / fan];
                    t = (int64_t)it.walker;
                    const unsigned int nbits = (unsigned int)__popcll(it.mask);
                    parts = nbits < fan ? nbits : fan;
                    const unsigned int part = idx % fan;
                    mask = (t < nprop && part < parts) ? uni_part_mask(it.mask, part, fan) : 0ull;
                    failed = false;
                    mine = mask != 0ull;
                    if (mine) {
                        chain = (int)(t / nhalf);
                        const GfCommon& c = s.commons[s.multi ? chain : 0];
                        const GfBsm& tb = *(s.multi ? s.tbs[chain] : s.tb);
                        // The chain's constants sit behind per-group pointers: every field the chain reads is a vector load, and
                        // between the fences of the row exchanges each one is waited for where it is used -- a dozen exposed
                        // latencies on the one path the whole half-step waits for.  Touch the lines now, back to back; the results
                        // are only looked at when the kernel ends.
                        {
                            const int kb = 63 - __clzll((long long)mask);
#define GF_TOUCH(p) warm ^= *reinterpret_cast<const volatile unsigned int*>(p)
                            GF_TOUCH(&c.idx_sm[0]); GF_TOUCH(&c.idx_mass[0]); GF_TOUCH(&c.mass_fixed[0]); GF_TOUCH(&c.idx_scale);
                            GF_TOUCH(&c.scale_fixed); GF_TOUCH(&c.idx_mm[0]); GF_TOUCH(&tb.texture); GF_TOUCH(&tb.npu_hi[0]);
                            GF_TOUCH(&tb.npu_lo[0]); GF_TOUCH(&tb.inv2e[kb]); GF_TOUCH(&tb.epow[kb]);
                            GF_TOUCH(s.pend_rows + (size_t)t * GF_PEND_STRIDE);
#undef GF_TOUCH
                        }
                        tm.terms(c, tb, s.pend_rows, 0, nprop, t, GF_PEND_STRIDE);
                        GF_ST_MARK(1);
                    }
                } else {
                    exhausted = true;
                }
            }
        }
        if (__ballot(!exhausted || mine) == 0ull) break;
        if (mine) {
            if (mask != 0ull && fan > 1u && __atomic_load_n(&s.ctl[2 * t + 1], __ATOMIC_RELAXED) != 0u) mask = 0ull;   // another part failed
            if (mask != 0ull) {
                const GfBsm* tbp = s.multi ? s.tbs[chain] : s.tb;
                const int k = 63 - __clzll((long long)mask);
                mask &= ~(1ull << k);
                const double res = tm.bin(tbp->inv2e[k], tbp->epow[k]);
                GF_ST_MARK(2);
                if (!(res < 1e-7)) { failed = true; mask = 0ull; }      // fr.py:493-494: the reference raises
            }
            if (mask == 0ull) {
                mine = false;
                if (lead) {
                    // report this part; the last part of the walker to report completes the walker's half-step
                    // (gf_sampler.hip stretch_body, after proposal_lnprob).  A walker in ONE part has nobody to meet: no counters.
                    bool last = parts == 1u, bad = failed;
                    if (!last) {
                        if (failed) atomicOr(&s.ctl[2 * t + 1], 1u);
                        __threadfence();
                        last = atomicAdd(&s.ctl[2 * t], 1u) == parts - 1u;
                        if (last) {
                            __threadfence();
                            bad = atomicOr(&s.ctl[2 * t + 1], 0u) != 0u;
                            s.ctl[2 * t] = 0u;                          // zero between uses
                            s.ctl[2 * t + 1] = 0u;
                        }
                    }
                    if (last) {
                        const double* row = s.pend_rows + (size_t)t * GF_PEND_STRIDE;
                        const int kk = (int)(t - (int64_t)chain * nhalf);
                        const int w = s.half * nhalf + kk;
                        const int64_t wi = (int64_t)chain * s.nwalkers + w;
                        const double lnq = row[GF_MAX_DIM], lhs = row[GF_MAX_DIM + 1];
                        const double lnk = s.lnp[wi];
                        bool accept = lhs > lnk - lnq;
                        if (bad) { accept = false; atomicAdd(s.flags, 1u); }
                        double* pw = s.pos + wi * ndim;
                        if (accept) {
                            for (int d = 0; d < ndim; ++d) pw[d] = row[d];
                            s.lnp[wi] = lnq;
                            s.naccept[wi] += 1u;
                        }
                        if (store_now) {
                            double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * s.nwalkers + w) * ndim;
                            for (int d = 0; d < ndim; ++d) dst[d] = accept ? row[d] : pw[d];
                            if (s.lnp_chain) s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * s.nwalkers + w] = accept ? lnq : lnk;
                        }
                        GF_ST_MARK(3);
                    }
                }
            }
        }
    }
    if (warm == 0x9e3779b9u && s.nchains < 0) s.flags[1] = warm;      // (never: keeps the early loads alive)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
#ifdef GF_SETTLE_TIMING
        atomicMax(&g_settle_t[4], (unsigned long long)wall_clock64());
#endif
        if (atomicAdd(&uq->done, 1u) == gridDim.x - 1) {                // the last block re-arms the queue for the next half-step
#ifdef GF_SETTLE_TIMING
            {   // fold this launch into the sums and re-arm the extrema
                const unsigned long long t0 = g_settle_t[0];
                for (int i = 1; i <= 4; ++i) { g_settle_t[5 + i] += g_settle_t[i] > t0 ? g_settle_t[i] - t0 : 0ull; g_settle_t[i] = 0ull; }
                g_settle_t[5] += 1ull; g_settle_t[10] += count;
                g_settle_c[0] += g_settle_cmax[1]; g_settle_c[1] += g_settle_cmax[2]; g_settle_cmax[1] = g_settle_cmax[2] = 0ull;
                g_settle_t[0] = ~0ull;
            }
#endif
            uq->count = 0;
            uq->done = 0;
            uq->head = 0;
            uq->overflow = 0;
            __threadfence();
        }
    }
}

// ---- test hooks (tests/test_gpu_unitarity_r3.py): the residual of explicit (walker, bin) pairs by the serial chain of
// gf_x87.hpp, one lane per pair -- the statement the host build checks against the CPU's x87 unit -- and by the three-lane
// distribution above.  They must agree bit for bit.
__global__ __launch_bounds__(64) void k_uni_debug_serial(const GfCommon* __restrict__ cp, const GfBsm* __restrict__ tbp,
                                                         const double* __restrict__ theta, int layout, int64_t n,
                                                         const int64_t* __restrict__ walkers, const int32_t* __restrict__ bins, int64_t npairs,
                                                         double* __restrict__ out)
{
    for (int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x; t < npairs; t += (int64_t)gridDim.x * 64) {
        cx87 hsm[3][3], hnp[3][3];
        walker_terms(*cp, *tbp, theta, layout, n, walkers[t], hsm, hnp);
        out[t] = walker_bin_residual(hsm, hnp, tbp->inv2e[bins[t]], tbp->epow[bins[t]]);
    }
}

template <class Team>
__global__ __launch_bounds__(UNI_BLOCK) void k_uni_debug_group(const GfCommon* __restrict__ cp, const GfBsm* __restrict__ tbp,
                                                               const double* __restrict__ theta, int layout, int64_t n,
                                                               const int64_t* __restrict__ walkers, const int32_t* __restrict__ bins, int64_t npairs,
                                                               double* __restrict__ out, int timing)
{
    __shared__ __attribute__((aligned(16))) double lds[(UNI_BLOCK / 64) * Team::PER_WAVE * Team::DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / Team::LANES;
    if (grp >= Team::PER_WAVE) return;
    Team tm;
    tm.init(lds + ((size_t)wave * Team::PER_WAVE + grp) * Team::DOUBLES, lane - grp * Team::LANES);
    const int64_t groups = (int64_t)gridDim.x * (UNI_BLOCK / 64) * Team::PER_WAVE;
    for (int64_t t = ((int64_t)blockIdx.x * (UNI_BLOCK / 64) + wave) * Team::PER_WAVE + grp; t < npairs; t += groups) {
        const long long c0 = clock64();
        const long long w0 = wall_clock64();
        tm.terms(*cp, *tbp, theta, layout, n, walkers[t], 0);
        const long long c1 = clock64();
        long long tick[5] = {0, 0, 0, 0, 0};
        const double res = tm.bin(tbp->inv2e[bins[t]], tbp->epow[bins[t]], timing >= 3 ? tick : nullptr);
        const long long c2 = clock64();
        // timing (tools/arb_latency_probe.py): shader-clock cycles of the walker's terms / of the bin / of the bin's sections
        // instead of the residual
        double o = res;
        if (timing == 1) o = (double)(c1 - c0);
        if (timing == 2) o = (double)(c2 - c1);
        if (timing == 3) o = (double)(tick[0] - c1);        // scaling + H
        if (timing == 4) o = (double)(tick[1] - tick[0]);   // tr, tr H^2, det, shared products
        if (timing == 5) o = (double)(tick[2] - tick[1]);   // b, Q, R, sqrt(Q^3), division, arccos
        if (timing == 6) o = (double)(tick[3] - tick[2]);   // sqrt Q, cos, eigenvalue
        if (timing == 7) o = (double)(tick[4] - tick[3]);   // eigenvector
        if (timing == 8) o = (double)(c2 - tick[4]);        // |X X^+|, sums
        if (timing == 9) o = (double)(c2 - c0) / (10.0 * (double)(wall_clock64() - w0));   // clock64 counts per ns of the 100 MHz wall clock
        if (timing == 10) o = 10.0 * (double)(wall_clock64() - w0);                         // terms + bin in ns of the wall clock
        if (tm.leader()) out[t] = o;
    }
}

}  // namespace

#ifdef GF_SETTLE_TIMING
extern "C" int gf_internal_settle_timing(unsigned long long* out, int reset)
{
    unsigned long long h[12];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_settle_t), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 12; ++i) out[i] = h[i];
    unsigned long long hc[4];
    if (hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_settle_c), sizeof(hc)) != hipSuccess) return 1;
    out[12] = hc[0]; out[13] = hc[1];
    if (reset) { hc[0] = hc[1] = hc[2] = hc[3] = 0ull; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_settle_c), hc, sizeof(hc)); }
    if (reset) { for (int i = 0; i < 12; ++i) h[i] = 0ull; h[0] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(g_settle_t), h, sizeof(h)) != hipSuccess) return 1; }
    return 0;
}
#endif

// the sampler's settle step; the count is on the device, the groups fetch dynamically: a modest fixed grid (it sits in a captured
// graph and runs after every half-step, almost always on an empty queue -- where its cost is the launch)
hipError_t gf_launch_stretch_settle(const GfSettleArgs& a, int cus, hipStream_t s)
{
    constexpr int64_t per_block = (UNI_BLOCK / 64) * Team9::PER_WAVE;
    const int64_t nprop = (int64_t)a.nchains * (a.nwalkers / 2);
    // enough groups to spread a short queue's walkers over (fan-out); on an empty queue every block returns after one load.  One
    // block per CU: measured on the C5 scan's sampling phase (256 chains x 512 walkers, a few hundred parked proposals per
    // half-step) 128 / 256 / 512 / 1024 / 2048 blocks give 0.105 / 0.120 / 0.116 / 0.130 / 0.141 s -- the step waits for ONE walker's
    // chain of dependent instructions (~70 us for the set-up and one bin, tools/arb_latency_probe.py), not for throughput
    int64_t blocks = (nprop * GF_UNI_MAX_FANOUT + per_block - 1) / per_block;
    if (blocks > cus) blocks = cus;
    if (blocks < 1) blocks = 1;
    static const int forced = [] { const char* e = gf_internal_env("GF_SETTLE_BLOCKS", 0); return e ? std::atoi(e) : 0; }();   // diagnostics / A-B
    if (forced > 0) blocks = forced;
    hipLaunchKernelGGL(k_stretch_settle<Team9>, dim3((unsigned)blocks), dim3(UNI_BLOCK), 0, s, a);
    return hipGetLastError();
}

// test hook: residuals of `npairs` explicit (walker, bin) pairs; which = 0 serial chain, 1 three-lane groups
hipError_t gf_launch_uni_debug(const GfCommon* d_common, const GfBsm* d_bsm, const double* theta, int layout, int64_t n, const int64_t* walkers,
                               const int32_t* bins, int64_t npairs, int which, double* out, hipStream_t s)
{
    if (which == 0) hipLaunchKernelGGL(k_uni_debug_serial, dim3(512), dim3(64), 0, s, d_common, d_bsm, theta, layout, n, walkers, bins, npairs, out);
    else if (which >= 100) hipLaunchKernelGGL(k_uni_debug_group<Team9>, dim3(512), dim3(UNI_BLOCK), 0, s, d_common, d_bsm, theta, layout, n, walkers, bins, npairs, out, which - 100);
    else hipLaunchKernelGGL(k_uni_debug_group<Team3>, dim3(512), dim3(UNI_BLOCK), 0, s, d_common, d_bsm, theta, layout, n, walkers, bins, npairs, out, which - 1);
    return hipGetLastError();
}

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
    // three lanes per walker, 21 walkers per wave at a time; the groups fetch their walkers dynamically, so any grid is correct:
    // one group per expected walker up to what is resident at once
    constexpr int64_t per_block = (UNI_BLOCK / 64) * (64 / 3);
    // (eight groups per expected walker: a short queue is spread over idle groups, uni_fanout)
    int64_t blocks = (expect * 8 + per_block - 1) / per_block;
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
    // A short queue is a latency: nine lanes per walker (one per matrix entry: 39 us for a walker's terms and one bin against 70 us on
    // three lanes); a long one is throughput: three lanes waste fewer on the scalar part of the chain.
    if (expect <= 2048 && forced == 0)
        hipLaunchKernelGGL(k_uni_resolve<Team9>, dim3((unsigned)blocks), dim3(UNI_BLOCK), 0, s, d_common, d_bsm, theta, layout, n, lnprob, status, uq, wq, seen);
    else
        hipLaunchKernelGGL(k_uni_resolve<Team3>, dim3((unsigned)blocks), dim3(UNI_BLOCK), 0, s, d_common, d_bsm, theta, layout, n, lnprob, status, uq, wq, seen);
    return hipGetLastError();
}
