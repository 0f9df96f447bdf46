// gf_sampler.hip -- device-resident affine-invariant ensemble sampler (the emcee step either side of
// the lnprob path).
//
// The reference drives the path through emcee's stretch move (un-vendored, requirements.txt:5; call
// sites golemflavor/mcmc.py:29-49).  Host-driven, every half-ensemble update costs a PCIe round trip and
// a launch for a few thousand evaluations.  Here the whole chain stays in HBM: one launch per
// half-ensemble update does proposal + lnprob + accept in place, for `nchains` independent ensembles
// stacked in one grid; the host only sees chains at the end.
//
// Stretch move (Goodman & Weare 2010, the published algorithm emcee runs by default), for walker k of
// the active half S with the complementary half C frozen during the launch:
//     z = ((a-1) u1 + 1)^2 / a,   j ~ U{0..|C|-1},   q = c_j - z (c_j - s_k),
//     accept iff (ndim-1) ln z + lnp(q) - lnp(s_k) > ln u3.
// Random numbers: one Philox4x32-10 block per (walker, half-step): counter = (global walker slot,
// 2*iteration + half), key = seed.  u1 has 53 bits, j and u3 32 bits each.  The accept test is evaluated
// as ln(z^(ndim-1) / u3) > lnp(s_k) - lnp(q): one logarithm.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <new>

#include "../../include/golemflavor_hip.h"
#include "gf_consts.h"

extern "C" const char* gf_internal_env(const char* name, int affects_results);   // gf_capi.hip: getenv with a record
#include "gf_device.hpp"
#include "gf_bsm_device.hpp"
#include "gf_launch.h"
#include "gf_unitarity_teams.hpp"      // Team9: the reference's unitarity chain on nine lanes (k_stretch_chain settles its own parked proposals)
#include "gf_devcache.h"                // large device allocations are cached, not handed back to the driver (hipMalloc / hipFree are macros from here on)

namespace {
using namespace gfdev;

__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                             uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(m1 >> 32), c1, k0, 0x96);     // xor of three, one instruction
        const uint32_t n1 = (uint32_t)m1;
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(m0 >> 32), c3, k1, 0x96);
        const uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Step counters live on the device so that a captured hipGraph of GRAPH_STEPS steps can be replayed with
// constant kernel arguments: each kernel node carries its own frozen `step_offset`; the base counters
// are advanced once per replay by k_tick (one extra 1-thread launch per GRAPH_STEPS steps).  (GfStepState: gf_launch.h)
typedef GfStepState StepState;

struct StretchArgs {
    StepState* state;
    double* pos;            // [nchains][nwalkers][ndim]
    double* lnp;            // [nchains][nwalkers]
    uint32_t* naccept;      // [nchains][nwalkers]
    uint32_t* flags;        // [0]: proposals the reference would have raised on (NON_UNITARY), whichever tier found out
    GfArbQueue* pq;         // BSM: proposals whose verdict the in-kernel tiers cannot settle are parked here ...
    double* pend_rows;      // ... with their row [GF_PEND_STRIDE] (theta | lnprob | ln(z^(ndim-1)/u)), for k_stretch_settle
    double* chain;          // [nchains][nstore_cap][nwalkers][ndim] or null
    double* lnp_chain;      // [nchains][nstore_cap][nwalkers] or null
    int64_t nstore_cap;
    uint64_t seed;
    int32_t nchains, nwalkers, half;
    int32_t step_offset;    // step index relative to the device-side base counters
    double a;
    // one model per chain (gf_sampler_create_multi): chain ch evaluates commons[ch] / tbs[ch] / ptabs[ch]
    const GfCommon* commons;
    const GfBsm* const* tbs;
    const double* const* ptabs;
    int32_t nbins_max;      // largest nbins over the chains' models (BSM); sizes the lane-group buffers
    int32_t lpw;            // lanes per walker for this run (host-side dispatch only)
    // random stream of chain ch: Philox counter word = stream_ids[ch] * (nwalkers / 2) + walker slot.  NULL: ch itself.
    // A scan hands in the GLOBAL grid index, so that a grid point's chain does not depend on which rank runs it or on
    // how many other points share its sampler.
    const uint64_t* stream_ids;
};

// start positions the reference would have died on (fr.py:493-498, raised while emcee evaluates p0): -inf and counted, so
// that an `-inf` run treats them as a host-driven one does (any finite proposal replaces them) and a `raise` run dies at once
__global__ void k_fix_start(const int32_t* __restrict__ status, int64_t n, double* __restrict__ lnp, uint32_t* __restrict__ flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (status[i] == ST_NON_UNITARY) { lnp[i] = -gf_inf(); atomicAdd(flags, 1u); }
}

__global__ void k_tick(StepState* st, int nsteps)
{
    st->iteration_base += (uint64_t)nsteps;
    st->run_step_base += nsteps;
}

// lnprob of the proposal held in LDS row `row`
// `pending` (out): the energy bins whose unitarity verdict the in-kernel tiers (gf_bsm_device.hpp) cannot settle, 0 = none.
// Such a proposal is not decided here: the half-step kernel parks it and k_stretch_settle (gf_unitarity.hip), next in stream
// order, takes the exact (emulated x87) verdict and completes the walker's update -- so that no sample enters the chain
// that the reference would have died on.
template <int NDIM, int MODE, int LPW>
__device__ __forceinline__ double proposal_lnprob(const GfCommon& c, const GfBsm* tb, const double* ctab,
                                                  const double* ttab, const double* row, int ndim, int& st, int sub,
                                                  double* fgrp, unsigned long long& pending)
{
    pending = 0ull;
    double val, fr[3];
    if (MODE == MODE_BSM_GAUSS) {
        double lp;
        const bool inbox = lnprior_tab<NDIM>(ctab, row, ndim, c.prior_const, lp);
        asm volatile("" : "+v"(lp));       // keeps the prior sum ahead of the bin loop (see k_bsm, gf_bsm.hip)
        val = -gf_inf();
        st = ST_OUT_OF_PRIOR;
        if (inbox) {
            UniAcc acc = {0.0, 0.0, 0ull, 2.0};
            flux_average<UNI_INLINE, LPW>(c, tb, ttab, row, fr, acc, sub, fgrp);
            st = (acc.clear_max < tb->uni_hi) ? ST_OK : ST_NON_UNITARY;       // tiers 1 and 2 (gf_bsm_device.hpp)
            pending = st == ST_OK ? uni_arbitration_mask(acc.amb, tb) : 0ull;
            val = lp + gauss_llh(c, fr);
            if (val != val && st == ST_OK) st = ST_NAN;
        }
    } else {
        eval_walker<NDIM, MODE, 0, false>(c, ctab, row, ndim, val, fr, st);
    }
    return val;
}

// One half-ensemble update for the walkers of one block.  `chain` / `k` = this thread's ensemble and its
// index in the active half (`valid` false for the padding threads of the last block of a chain).
// LPW > 1 (BSM posteriors on small ensembles): LPW adjacent lanes hold the same walker and split its energy
// bins (flux_average); everything else they compute redundantly and identically, lane `sub == 0` writes.
template <int NDIM, int MODE, int LPW>
__device__ __forceinline__ void stretch_body(const GfCommon& c, const GfBsm* __restrict__ tb, const double* __restrict__ ptab,
                                             const StretchArgs& s, const int chain, const int k, const bool valid, const int sub)
{
    extern __shared__ __attribute__((aligned(16))) double fdyn[];    // LPW > 1: per lane group [nbins_max][3] + [LPW]
    double* fgrp = LPW > 1 ? fdyn + (threadIdx.x / LPW) * GF_FGRP_DOUBLES(s.nbins_max, LPW) : nullptr;
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    __shared__ __attribute__((aligned(16))) double tiles[GF_WAVES_PER_BLOCK][GF_WAVE * ND];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4 + 20];
    double* ttab = ctab + GF_MAX_DIM * 4;
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    if (MODE == MODE_BSM_GAUSS && threadIdx.x >= 64 && threadIdx.x < 64 + 18) {
        const int k = threadIdx.x - 64, e = k >> 1;
        const int idx = e == 0 ? 0 : e == 1 ? 4 : e == 2 ? 8 : e <= 4 ? 1 : e <= 6 ? 2 : 5;
        const bool im = e == 4 || e == 6 || e == 8;
        const double* srcp = (k & 1) ? (im ? tb->t2_im : tb->t2_re) : (im ? tb->t1_im : tb->t1_re);
        ttab[k] = srcp[idx];
    }
    __syncthreads();

    const int ndim = NDIM ? NDIM : c.ndim;
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = threadIdx.x / GF_WAVE;
    double* row = tiles[wave] + lane * ndim;
    const uint64_t iteration = s.state->iteration_base + (uint64_t)s.step_offset;
    const int64_t run_step = s.state->run_step_base + s.step_offset;
    const int thin = s.state->thin;
    const bool store_now = s.state->store != 0 && s.chain != nullptr && (run_step % thin) == 0;
    const int64_t store_index = s.state->store_base + (run_step + thin - 1) / thin;   // stored steps before this one
    const int nhalf = s.nwalkers / 2;
    const uint64_t sid = s.stream_ids ? s.stream_ids[chain] : (uint64_t)chain;
    const uint64_t g = sid * (uint64_t)nhalf + (uint64_t)k;  // walker slot of the chain's random stream: the Philox counter
    if (valid) {
    const int w = s.half * nhalf + k;                        // this walker, in the active half
    const int cbase = (1 - s.half) * nhalf;                  // complementary half

    uint32_t r[4];
    philox_block((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(2 * iteration + s.half),
                 (uint32_t)((2 * iteration + s.half) >> 32), (uint32_t)s.seed, (uint32_t)(s.seed >> 32), r);
    const double u1 = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
    const int j = (int)(((uint64_t)r[2] * (uint64_t)nhalf) >> 32);
    const double u3 = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
    const double zr = fma(s.a - 1.0, u1, 1.0);
    const double z = zr * zr / s.a;

    const double* sk = s.pos + ((int64_t)chain * s.nwalkers + w) * ndim;
    const double* cj = s.pos + ((int64_t)chain * s.nwalkers + cbase + j) * ndim;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (!NDIM && d >= ndim) break;
        const double cv = cj[d];
        row[d] = fma(-z, cv - sk[d], cv);                    // q = c_j - z (c_j - s_k)
    }
    int st;
    unsigned long long pending;
    const double lnq = proposal_lnprob<NDIM, MODE, LPW>(c, tb, ctab, ttab, row, ndim, st, sub, fgrp, pending);
    const int64_t wi = (int64_t)chain * s.nwalkers + w;
    const double lnk = s.lnp[wi];
    // z^(ndim-1) / u3
    double zp = 1.0;
    for (int d = 1; d < ndim; ++d) zp *= z;
    const double lhs = log(zp / u3);
    if (MODE == MODE_BSM_GAUSS && pending != 0ull) {
        // undecided unitarity: park the proposal; k_stretch_settle completes this walker's half-step
        if (sub == 0) {
            const int64_t t = (int64_t)chain * nhalf + k;
            double* dst = s.pend_rows + (size_t)t * GF_PEND_STRIDE;
            for (int d = 0; d < ndim; ++d) dst[d] = row[d];
            dst[GF_MAX_DIM] = lnq;
            dst[GF_MAX_DIM + 1] = lhs;
            const unsigned int at = atomicAdd(&s.pq->count, 1u);
            if (at < s.pq->cap) {
                GfArbItem it;
                it.walker = (unsigned long long)t;
                it.mask = pending;
                s.pq->items[at] = it;
            } else {
                s.pq->overflow = 1u;                             // capacity = every proposal of a half-step: cannot happen
            }
        }
        return;
    }
    bool accept = lhs > lnk - lnq;                           // false for NaN and for lnq = -inf
    if (st == ST_NON_UNITARY) {                              // the reference raises inside ln_prob here
        accept = false;
        if (sub == 0) atomicAdd(s.flags, 1u);
    }
    if (LPW > 1 && sub != 0) return;                         // the group's results are identical: one writer
    if (accept) {
        double* dst = s.pos + wi * ndim;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (!NDIM && d >= ndim) break;
            dst[d] = row[d];
        }
        s.lnp[wi] = lnq;
        s.naccept[wi] += 1u;
    }
    if (store_now) {
        double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * s.nwalkers + w) * ndim;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (!NDIM && d >= ndim) break;
            dst[d] = accept ? row[d] : sk[d];
        }
        if (s.lnp_chain)
            s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * s.nwalkers + w] = accept ? lnq : lnk;
    }
    }   // valid
}

// every ensemble samples the same posterior: constants by value (scalar registers), walkers packed densely
#ifndef GF_STRETCH_WAVES
#define GF_STRETCH_WAVES 2      // blocks of 256 threads per CU the half-step kernels are compiled for (A/B: tools/build_variants.sh)
#endif
template <int NDIM, int MODE, int LPW>
__global__ __launch_bounds__(GF_BLOCK, GF_STRETCH_WAVES) void k_stretch(const GfCommon c, const GfBsm* __restrict__ tb,
                                                          const double* __restrict__ ptab, const StretchArgs s)
{
    const int nhalf = s.nwalkers / 2;
    const int64_t t = (int64_t)blockIdx.x * GF_BLOCK + threadIdx.x;
    const int64_t g = t / LPW;
    const bool valid = g < (int64_t)s.nchains * nhalf;
    const int chain = valid ? (int)(g / nhalf) : 0;
    const int k = valid ? (int)(g - (int64_t)chain * nhalf) : 0;
    stretch_body<NDIM, MODE, LPW>(c, tb, ptab, s, chain, k, valid, (int)(t % LPW));
}

// one posterior per ensemble (grid scans, SURVEY.md 8(e) "all chains of a GPU stacked into one launch"):
// blockIdx.y = chain, so that a block's constants are block-uniform and come from scalar loads
template <int NDIM, int MODE, int LPW>
__global__ __launch_bounds__(GF_BLOCK, GF_STRETCH_WAVES) void k_stretch_multi(const StretchArgs s)
{
    const int chain = blockIdx.y;
    const int t = blockIdx.x * GF_BLOCK + threadIdx.x;
    const int k = t / LPW;
    stretch_body<NDIM, MODE, LPW>(s.commons[chain], s.tbs[chain], s.ptabs[chain], s, chain, k, k < s.nwalkers / 2, t % LPW);
}

// Lanes per walker for a BSM half-step of `walkers` proposals, from {1, 2, 4, 16}.  Splitting a walker's bins over L lanes shortens
// its critical path from nbins to nbins / L bin evaluations but repeats the per-walker prologue on every lane and multiplies the
// waves: small ensembles (a latency: one walker's chain on a GPU that is mostly idle) want the widest split, large ones the
// narrowest that still keeps more than one wave on a SIMD.  A cost model instead of a table (round 4; tools/c5_lpw_ab.py): a lane
// runs P + ceil(nbins / L) B instructions (P ~ 2000: prologue + the tier-2 terms of the chains that need them, B ~ 400 per bin),
// a SIMD with w resident waves gives each an issue slot every max(7, 4 w) cycles, and waves beyond what is resident (three per
// SIMD by registers; two at L = 2, whose 128 group buffers per block take more LDS) queue for another round.  C5's half-step of
// 65 536 proposals: L = 2 (one round of two waves per SIMD) -- measured 79 us per half-step against 84 at L = 4, 94 at L = 1,
// 122 at L = 16; ensembles of a few thousand proposals keep 16.
inline int lanes_per_walker(int mode, int64_t walkers, int nbins_max, int cus)
{
    if (mode != MODE_BSM_GAUSS || nbins_max < 2) return 1;
    const char* force = gf_internal_env("GF_SAMPLER_LPW", 0);                 // diagnostics / A-B, read per run
    if (force) { const int f = std::atoi(force); if (f == 1 || f == 2 || f == 4 || f == 16) return f; }
    const int64_t simds = (int64_t)(cus > 0 ? cus : 256) * 4;
    int best = 1;
    double best_cost = 0.0;
    for (int lpw : {1, 2, 4, 16}) {
        const size_t lds = (size_t)(GF_BLOCK / lpw) * GF_FGRP_DOUBLES(nbins_max, lpw) * sizeof(double);
        if (lpw > 1 && lds > 48 * 1024) continue;                            // group buffers no longer fit beside the tiles
        const int64_t waves = (walkers * lpw + GF_WAVE - 1) / GF_WAVE;
        const int64_t wmax = lpw == 2 ? 2 : 3;
        const int64_t per_simd = (waves + simds - 1) / simds;
        const int64_t w = per_simd < wmax ? per_simd : wmax;
        const int64_t rounds = (waves + simds * wmax - 1) / (simds * wmax);
        const double interval = 4.0 * (double)w > 7.0 ? 4.0 * (double)w : 7.0;
        const double cost = (double)rounds * (2000.0 + 400.0 * (double)((nbins_max + lpw - 1) / lpw)) * interval;
        if (lpw == 1 || cost < best_cost) { best = lpw; best_cost = cost; }
    }
    return best;
}

template <int NDIM, int MODE, int LPW>
hipError_t launch_stretch_nml(const GfCommon& c, const GfBsm* tb, const double* ptab, const StretchArgs& a, hipStream_t st)
{
    const size_t lds = LPW > 1 ? (size_t)(GF_BLOCK / LPW) * GF_FGRP_DOUBLES(a.nbins_max, LPW) * sizeof(double) : 0;
    if (a.commons) {
        const dim3 grid((unsigned)(((int64_t)(a.nwalkers / 2) * LPW + GF_BLOCK - 1) / GF_BLOCK), a.nchains);
        hipLaunchKernelGGL((k_stretch_multi<NDIM, MODE, LPW>), grid, dim3(GF_BLOCK), lds, st, a);
    } else {
        const int64_t total = (int64_t)a.nchains * (a.nwalkers / 2) * LPW;
        hipLaunchKernelGGL((k_stretch<NDIM, MODE, LPW>), dim3((unsigned)((total + GF_BLOCK - 1) / GF_BLOCK)), dim3(GF_BLOCK), lds, st,
                           c, tb, ptab, a);
    }
    return hipGetLastError();
}

// ---- one workgroup per ensemble, a whole run in ONE launch --------------------------------------------
// An ensemble of up to a few thousand walkers is a few waves of work per half-step: launched as a grid, every
// half-step is a ~3 us launch that the GPU spends mostly idle (the emcee regime: 100-walker chains, C1).  Here
// block b owns ensemble b for the whole run: walkers, their lnprob and acceptance counters live in LDS, the two
// half-steps of a step are separated by a workgroup barrier instead of a kernel boundary, and the only HBM
// traffic is the stored chain.  Random stream, proposal, evaluation and accept rule are those of k_stretch, so
// the chain is bitwise the same (tests compare the two).  PRIOR_ONLY and SM_GAUSS posteriors.
struct PersistArgs {
    const GfCommon* commons;        // [nmodels]
    const double* const* ptabs;     // [nmodels]
    int32_t nmodels;                // 1: every chain samples commons[0]; else one per chain
    int32_t nwalkers;
    double* pos;                    // [nchains][nwalkers][ndim]
    double* lnp;                    // [nchains][nwalkers]
    uint32_t* naccept;              // [nchains][nwalkers]
    double* chain;                  // [nchains][nstore_cap][nwalkers][ndim] or null
    double* lnp_chain;
    int64_t nstore_cap, store_base;
    uint64_t seed, iteration_base;
    int64_t nsteps;
    int32_t thin, store;
    double a;
    const uint64_t* stream_ids;     // as StretchArgs::stream_ids
    int32_t workers;                // threads that move walkers; the block's other threads (if any) draw the random numbers one half-step ahead
};

// MAXT: the largest workgroup the instance is compiled for.  1024 threads would cap the kernel at 128 VGPRs, which the
// likelihood instances overflow (32-256 B of scratch per lane inside the step loop, round 2); workgroups are at most 512 threads
// (two waves per SIMD, up to 256 VGPRs; the instances use 57-195) and a larger half-ensemble takes two or more passes of the loop
// over k.  The reference's 100-walker chain runs on ONE wave.
template <int NDIM, int MODE, int MAXT>
__global__ __launch_bounds__(MAXT) void k_stretch_persist(const PersistArgs s)
{
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int chain = blockIdx.x;
    const GfCommon& c = s.commons[s.nmodels > 1 ? chain : 0];
    const double* __restrict__ ptab = s.ptabs[s.nmodels > 1 ? chain : 0];
    const int ndim = NDIM ? NDIM : c.ndim;
    const int nw = s.nwalkers, nhalf = nw / 2, nt = blockDim.x;
    // Workers and producers.  A half-step's random quantities -- the stretch factor z, the partner j and the left side of the
    // accept test, ln(z^(ndim-1) / u) -- depend on nothing but the counters, a third of a half-step's instructions (Philox4x32-10,
    // the conversions, a logarithm) on the one wave the reference's 100-walker chain runs on.  With `workers` < blockDim the
    // block's other threads compute them ONE HALF-STEP AHEAD into a double-buffered LDS table while the workers move the
    // walkers; the barrier that separates the half-steps anyway hands the table over.  Same expressions, same values: the chain
    // is bitwise the one-wave chain (tests).
    const int ntw = s.workers > 0 && s.workers < nt ? s.workers : nt;        // worker threads
    const int ntp = nt - ntw;                                                // producer threads (0: the workers draw for themselves)
    const bool producer = (int)threadIdx.x >= ntw;
    // LDS: ctab[64] | pos[nw][ndim] | lnp[nw] | rows[ntw][ndim] | naccept[nw] (u32) | [producers: z[2][nhalf] | lhs[2][nhalf] | j[2][nhalf]]
    double* ctab = dyn;
    double* pos = ctab + GF_MAX_DIM * 4;
    double* lnp = pos + (size_t)nw * ndim;
    double* rows = lnp + nw;
    uint32_t* nacc = reinterpret_cast<uint32_t*>(rows + (size_t)ntw * ndim);
    double* dz = reinterpret_cast<double*>(nacc + ((nw + 1) & ~1));
    double* dlhs = dz + 2 * (size_t)nhalf;
    int* dj = reinterpret_cast<int*>(dlhs + 2 * (size_t)nhalf);
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    const int64_t cw = (int64_t)chain * nw;
    for (int i = threadIdx.x; i < nw * ndim; i += nt) pos[i] = s.pos[cw * ndim + i];
    for (int i = threadIdx.x; i < nw; i += nt) { lnp[i] = s.lnp[cw + i]; nacc[i] = s.naccept[cw + i]; }
    double* row = rows + (size_t)(producer ? 0 : threadIdx.x) * ndim;
    const uint32_t k0 = (uint32_t)s.seed, k1 = (uint32_t)(s.seed >> 32);
    const uint64_t gbase = (s.stream_ids ? s.stream_ids[chain] : (uint64_t)chain) * (uint64_t)nhalf;
    // the draws of walker slot k at half-step counter ctr
    auto draw = [&](int k, uint64_t ctr, double& z, int& j, double& lhs) {
        const uint64_t g = gbase + (uint64_t)k;
        uint32_t r[4];
        philox_block((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), k0, k1, r);
        const double u1 = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
        j = (int)(((uint64_t)r[2] * (uint64_t)nhalf) >> 32);
        const double u3 = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
        const double zr = fma(s.a - 1.0, u1, 1.0);
        z = zr * zr / s.a;
        double zp = 1.0;
        for (int d = 1; d < ndim; ++d) zp *= z;
        lhs = log(zp / u3);
    };
    const uint64_t ctr_first = 2 * s.iteration_base, ctr_end = ctr_first + 2 * (uint64_t)s.nsteps;
    if (producer && s.nsteps > 0)
        for (int k = (int)threadIdx.x - ntw; k < nhalf; k += ntp) {
            double z, lhs; int j;
            draw(k, ctr_first, z, j, lhs);
            dz[k] = z; dlhs[k] = lhs; dj[k] = j;                     // buffer 0 = (ctr_first & 1) * nhalf: see below
        }
    __syncthreads();

    for (int64_t step = 0; step < s.nsteps; ++step) {
        const uint64_t iteration = s.iteration_base + (uint64_t)step;
        const bool store_now = s.store != 0 && s.chain != nullptr && (step % s.thin) == 0;
        const int64_t store_index = s.store_base + (step + s.thin - 1) / s.thin;
        for (int half = 0; half < 2; ++half) {
            const int cbase = (1 - half) * nhalf;
            const uint64_t ctr = 2 * iteration + half;
            const size_t buf = (size_t)(half) * nhalf;                // ctr_first is even: the buffer of counter ctr is its parity
            if (producer) {
                if (ctr + 1 < ctr_end)
                    for (int k = (int)threadIdx.x - ntw; k < nhalf; k += ntp) {
                        double z, lhs; int j;
                        draw(k, ctr + 1, z, j, lhs);
                        const size_t nb = (size_t)(1 - half) * nhalf;
                        dz[nb + k] = z; dlhs[nb + k] = lhs; dj[nb + k] = j;
                    }
            } else
            for (int k = threadIdx.x; k < nhalf; k += ntw) {
                const int w = half * nhalf + k;
                double z, lhs; int j;
                if (ntp > 0) { z = dz[buf + k]; lhs = dlhs[buf + k]; j = dj[buf + k]; }
                else draw(k, ctr, z, j, lhs);
                double* sk = pos + (size_t)w * ndim;
                const double* cj = pos + (size_t)(cbase + j) * ndim;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (!NDIM && d >= ndim) break;
                    const double cv = cj[d];
                    row[d] = fma(-z, cv - sk[d], cv);
                }
                int st;
                unsigned long long pending;
                const double lnq = proposal_lnprob<NDIM, MODE, 1>(c, nullptr, ctab, nullptr, row, ndim, st, 0, nullptr, pending);
                const double lnk = lnp[w];
                const bool accept = lhs > lnk - lnq;
                if (accept) {
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        if (!NDIM && d >= ndim) break;
                        sk[d] = row[d];
                    }
                    lnp[w] = lnq;
                    nacc[w] += 1u;
                }
                if (store_now) {
                    double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * nw + w) * ndim;
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        if (!NDIM && d >= ndim) break;
                        dst[d] = sk[d];                           // the walker's position after this half-step
                    }
                    if (s.lnp_chain) s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * nw + w] = lnp[w];
                }
            }
            __syncthreads();                                      // the other half moves next: it reads these walkers (and the next draws are in place)
        }
    }
    for (int i = threadIdx.x; i < nw * ndim; i += nt) s.pos[cw * ndim + i] = pos[i];
    for (int i = threadIdx.x; i < nw; i += nt) { s.lnp[cw + i] = lnp[i]; s.naccept[cw + i] = nacc[i]; }
}

// threads and dynamic LDS of the persistent kernel; lds == 0: the ensemble does not fit one workgroup
inline void persist_geometry(int nwalkers, int ndim, int* threads, size_t* lds, int* workers = nullptr, int nchains = 1, int cus = 256)
{
    const int nhalf = nwalkers / 2;
    int nt = ((nhalf + GF_WAVE - 1) / GF_WAVE) * GF_WAVE;
    if (nt > 512) nt = 512;       // a workgroup of 1024 threads is capped at 128 VGPRs, which the likelihood instances overflow (32-256 B of
                                  // scratch per lane inside the step loop); the half-ensemble loop strides by the block size anyway
    // as many producer threads again while the block stays within 512 threads (k_stretch_persist) -- where a chain's LATENCY is what
    // counts: up to two ensembles per CU one 100-walker chain steps in 3.6 us instead of 4.1; with the GPU full of ensembles the
    // producers only take issue slots and LDS from the workers (4096 chains: 2.8e10 evals/s against 3.9e10; tools/bench_c1_chains.py)
    static const bool no_producers = gf_internal_env("GF_SAMPLER_NO_PRODUCERS", 0) != nullptr;       // diagnostics / A-B
    const size_t base = sizeof(double) * ((size_t)GF_MAX_DIM * 4 + (size_t)nwalkers * ndim + nwalkers + (size_t)nt * ndim) +
                        sizeof(uint32_t) * (size_t)((nwalkers + 1) & ~1);
    const size_t draws = (sizeof(double) * 4 + sizeof(int) * 2) * (size_t)nhalf;
    const bool producers = nt <= 256 && !no_producers && base + draws <= 64 * 1024 && nchains <= 2 * cus;
    const size_t bytes = base + (producers ? draws : 0);
    if (workers) *workers = nt;
    *threads = producers ? 2 * nt : nt;
    *lds = (bytes <= 64 * 1024 && nhalf <= 4 * 1024) ? bytes : 0;
}

template <int NDIM>
hipError_t launch_persist_n(int mode, int nchains, int threads, size_t lds, const PersistArgs& a, hipStream_t st)
{
    if (mode == MODE_PRIOR_ONLY) {
        if (threads <= 256) hipLaunchKernelGGL((k_stretch_persist<NDIM, MODE_PRIOR_ONLY, 256>), dim3(nchains), dim3(threads), lds, st, a);
        else hipLaunchKernelGGL((k_stretch_persist<NDIM, MODE_PRIOR_ONLY, 512>), dim3(nchains), dim3(threads), lds, st, a);
    } else {
        if (threads <= 256) hipLaunchKernelGGL((k_stretch_persist<NDIM, MODE_SM_GAUSS, 256>), dim3(nchains), dim3(threads), lds, st, a);
        else hipLaunchKernelGGL((k_stretch_persist<NDIM, MODE_SM_GAUSS, 512>), dim3(nchains), dim3(threads), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_persist(int mode, int ndim, int nchains, int threads, size_t lds, const PersistArgs& a, hipStream_t st)
{
    switch (ndim) {
    case 4: return launch_persist_n<4>(mode, nchains, threads, lds, a, st);
    case 6: return launch_persist_n<6>(mode, nchains, threads, lds, a, st);
    case 7: return launch_persist_n<7>(mode, nchains, threads, lds, a, st);
    case 12: return launch_persist_n<12>(mode, nchains, threads, lds, a, st);
    default: return launch_persist_n<0>(mode, nchains, threads, lds, a, st);
    }
}

// ---- one workgroup per CHAIN, BSM posteriors ------------------------------------------------------------------------------------
// The grid kernels above advance ALL chains of a sampler half-step by half-step, and since round 3 a half-step is followed by
// k_stretch_settle, which takes the exact (emulated x87) unitarity verdict of the proposals the in-kernel tiers could not settle:
// ~66 us of ONE walker's dependent arithmetic.  Measured on the C5 scan (256 chains x 512 walkers, profiles/r03/
// arbitration_latency.txt): ~120 of 65 536 proposals park per half-step, in 30 of the 256 chains -- and all 256 chains, which share
// nothing, waited 66 us for them at the kernel boundary: 37 us of proposals + 67 us of settling per half-step.
//
// Chains are independent (emcee runs them as separate jobs: submitter/sens_dag.py:75-95), so here a workgroup OWNS a chain for a
// whole block of steps: its half-steps follow each other behind a workgroup barrier, the walkers stay where the grid kernels
// keep them (HBM; a chain's 49 KB live in this CU's caches), and when a half-step parks proposals the SAME workgroup settles
// them on the spot -- its four waves become 56 nine-lane teams (gf_unitarity_teams.hpp), a walker's undecided bins fan out over
// them, the last part to finish completes the walker's accept step -- and goes on.  Nobody else waits.  Same Philox counters,
// same proposal_lnprob instance, same chain arithmetic, same accept rule as k_stretch + k_stretch_settle: the chain is bitwise
// the grid sampler's (tests/test_gpu_sampler.py).
struct ChainArgs {
    const GfCommon* commons;        // [nmodels]
    const GfBsm* const* tbs;        // [nmodels] (nmodels > 1), else null
    const GfBsm* tb;                // nmodels == 1
    const double* const* ptabs;     // [nmodels]
    int32_t nmodels;                // 1: every chain samples commons[0]
    int32_t nwalkers;
    double* pos;                    // [nchains][nwalkers][ndim]
    double* lnp;                    // [nchains][nwalkers]
    uint32_t* naccept;              // [nchains][nwalkers]
    uint32_t* flags;                // [0]: proposals the reference would have raised on
    double* pend_rows;              // [nchains * nwalkers / 2][GF_PEND_STRIDE]: a parked proposal's row
    double* chain;                  // [nchains][nstore_cap][nwalkers][ndim] or null
    double* lnp_chain;
    int64_t nstore_cap, store_base; // chain slot of the RUN's first stored step
    uint64_t seed, iteration_base;  // Philox counter word of this launch's first step
    int64_t run_step_base;          // steps of the run done before this launch (thinning counts from the run's start)
    int32_t nsteps, thin, store;
    double a;
    const uint64_t* stream_ids;     // as StretchArgs::stream_ids
    double* lazy_rows;              // [nchains][lazy_cap][GF_PEND_STRIDE]: undecided proposals that are rejected either way ...
    unsigned long long* lazy_mask;  // [nchains][lazy_cap]: ... with their undecided bins (k_stretch_chain settles them in bulk)
    int32_t lazy_cap;
    // k_stretch_flow: versioned walker states and the in-flight proposals' rows / Hamiltonian terms
    double* pv;                     // [nchains][FLOW_VERS][nwalkers][ndim]: a walker's position after its v-th update, in slot v mod FLOW_VERS
    double* lv;                     // [nchains][FLOW_VERS][nwalkers]
    double* frows;                  // [nchains][nwalkers][GF_PEND_STRIDE]: the parked proposal of a walker (it has at most one in flight)
    double* fterms;                 // [nchains][nwalkers][9][8]: its two Hamiltonian terms once a team has built them
    unsigned long long* stats;      // [nchains][8] (may be null): per chain, ns on the 100 MHz wall clock spent in [0] proposals, [1] settling
                                    // parked proposals, [2] bulk settlement; [3] proposals waited for, [4] passes that waited, [5] settled in bulk,
                                    // [6] passes, [7] bulk settlements
};

// workgroup of the per-chain sampler: eight waves -- two per SIMD, 256 VGPRs each -- = 56 nine-lane teams when proposals are settled
constexpr int CH_BLOCK = 512;
constexpr int CH_WAVES = CH_BLOCK / GF_WAVE;

// LDS pointers that stay LDS pointers across a call (ds_ instructions, not flat_)
typedef __attribute__((address_space(3))) double* LdsD;
typedef __attribute__((address_space(3))) unsigned int* LdsU;
typedef __attribute__((address_space(3))) int* LdsI;
typedef __attribute__((address_space(3))) unsigned long long* LdsL;

// One walker's proposal of k_stretch_chain -- exactly as stretch_body makes it -- as a function of its own: the kernel around it holds
// the nine-lane teams' code (~210 VGPRs of its own), and compiled as one body for eight waves per CU (256 VGPRs) the allocator spilled
// inside the evaluation's bin loop (36 us per pass instead of 24).  Out of line, each side has the whole budget.
template <int NDIM>
__device__ __attribute__((noinline)) void chain_propose(const ChainArgs& s, const GfCommon& c, const GfBsm* __restrict__ tb, LdsD ctab_l, LdsD ttab_l,
                                                        LdsD row_l, const int chain, const int k, const int half, const uint64_t iteration,
                                                        const bool store_now, const int64_t store_index, const uint64_t sid, double* my_pend,
                                                        double* lz_rows, unsigned long long* lz_mask, LdsU pk_n, LdsI pk_k, LdsL pk_mask, LdsU lz_n)
{
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    const double* ctab = (const double*)ctab_l;
    const double* ttab = (const double*)ttab_l;
    double* row = (double*)row_l;
    const int ndim = NDIM ? NDIM : c.ndim;
    const int nw = s.nwalkers, nhalf = nw / 2;
    const int cbase = (1 - half) * nhalf;
    const uint32_t k0s = (uint32_t)s.seed, k1s = (uint32_t)(s.seed >> 32);
    {
        // ---- the proposal, exactly as stretch_body makes it
        const uint64_t g = sid * (uint64_t)nhalf + (uint64_t)k;
        const int w = half * nhalf + k;
        uint32_t r[4];
        philox_block((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(2 * iteration + half),
                     (uint32_t)((2 * iteration + half) >> 32), k0s, k1s, r);
        const double u1 = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
        const int j = (int)(((uint64_t)r[2] * (uint64_t)nhalf) >> 32);
        const double u3 = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
        const double zr = fma(s.a - 1.0, u1, 1.0);
        const double z = zr * zr / s.a;
        const int64_t wi = (int64_t)chain * nw + w;
        const double* sk = s.pos + wi * ndim;
        const double* cj = s.pos + ((int64_t)chain * nw + cbase + j) * ndim;
    #pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (!NDIM && d >= ndim) break;
            const double cv = cj[d];
            row[d] = fma(-z, cv - sk[d], cv);
        }
        int st;
        unsigned long long pending;
        const double lnq = proposal_lnprob<NDIM, MODE_BSM_GAUSS, 1>(c, tb, ctab, ttab, row, ndim, st, 0, nullptr, pending);
        const double lnk = s.lnp[wi];
        double zp = 1.0;
        for (int d = 1; d < ndim; ++d) zp *= z;
        const double lhs = log(zp / u3);
        bool accept = lhs > lnk - lnq;                    // false for NaN and for lnq = -inf
        if (pending != 0ull && accept) {
            // undecided unitarity and the verdict DECIDES: park it; this workgroup settles it below, before anybody
            // reads this walker again
            double* dst = my_pend + (size_t)k * GF_PEND_STRIDE;
            for (int d = 0; d < ndim; ++d) dst[d] = row[d];
            dst[GF_MAX_DIM] = lnq;
            dst[GF_MAX_DIM + 1] = lhs;
            const unsigned int at = atomicAdd((unsigned int*)pk_n, 1u);
            pk_k[at] = k;
            pk_mask[at] = pending;
        } else {
            if (pending != 0ull) {
                // undecided, but rejected either way: only the count wants the verdict -- later
                const unsigned int at = atomicAdd((unsigned int*)lz_n, 1u);
                double* dst = lz_rows + (size_t)at * GF_PEND_STRIDE;
                for (int d = 0; d < ndim; ++d) dst[d] = row[d];
                lz_mask[at] = pending;
            }
            if (st == ST_NON_UNITARY) { accept = false; atomicAdd(s.flags, 1u); }      // the reference raises inside ln_prob here
            if (accept) {
                double* dst = s.pos + wi * ndim;
    #pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (!NDIM && d >= ndim) break;
                    dst[d] = row[d];
                }
                s.lnp[wi] = lnq;
                s.naccept[wi] += 1u;
            }
            if (store_now) {
                double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * nw + w) * ndim;
    #pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (!NDIM && d >= ndim) break;
                    dst[d] = accept ? row[d] : sk[d];
                }
                if (s.lnp_chain) s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * nw + w] = accept ? lnq : lnk;
            }
        }
    }
}

// The workgroup's eight waves as 56 nine-lane teams on a list of `count` proposals whose unitarity is undecided.  Two stages per chunk
// of up to WT_CAP walkers, so that nothing is computed twice and no team idles behind a walker with many bins:
//   A  a team per WALKER builds the walker's two Hamiltonian terms (fr.py:380-394 in emulated x87: ~16 us) and leaves them in LDS;
//   B  a team per (walker, bin) PAIR takes the terms from LDS and evaluates that one energy bin (~23 us).  Pairs are handed out
//      rank-major -- every walker's highest undecided energy first (the likeliest to fail), then the second-highest ... -- and a pair
//      whose walker has failed meanwhile (fr.py:493-494: the reference raises at the first) is dropped unevaluated.  The last pair of
//      a walker to report calls done(item, failed).
// (k_stretch_settle, gf_unitarity.hip, gives every part of a walker its own copy of the terms: right for a kernel the whole GPU runs,
// 3 584 teams for a few hundred pairs; here 56 teams face up to ~100 pairs per half-step of a chain in the failing region, and the
// terms were 40 % of the work.)
//   row_of(item) -> the proposal's row (theta at [0, ndim))       mask_of(item) -> its undecided bins (non-zero)
//   head, aux -> LDS words;   ctl -> LDS [2 * CH_BLOCK], zero on entry and on exit;   wt -> LDS [WT_CAP][9][8] doubles
// A workgroup-uniform call: it contains workgroup barriers.
constexpr int WT_CAP = 64;                             // walkers whose terms are held at once (64 x 576 B = 36 KB)
constexpr int WT_DOUBLES = WT_CAP * Team9::LANES * 8;
__device__ __forceinline__ int nth_bit_from_top(unsigned long long m, unsigned int n)
{
    for (unsigned int i = 0; i < n; ++i) m &= ~(1ull << (63 - __clzll((long long)m)));
    return 63 - __clzll((long long)m);
}
template <class RowOf, class MaskOf, class Done>
__device__ __forceinline__ void chain_settle(double* uni, double* wt, unsigned int count, unsigned int* head, unsigned int* aux, unsigned int* ctl,
                                             const GfCommon& c, const GfBsm* __restrict__ tb, RowOf row_of, MaskOf mask_of, Done done)
{
    const int lane = threadIdx.x & (GF_WAVE - 1), wave = threadIdx.x / GF_WAVE;
    const int grp = lane / Team9::LANES;
    const bool team_active = grp < Team9::PER_WAVE;                                    // lane 63 has no team
    const int tl = lane - grp * Team9::LANES;                                          // lane within the team
    Team9 tm;
    tm.init(uni + ((size_t)wave * Team9::PER_WAVE + (team_active ? grp : 0)) * Team9::DOUBLES, tl);
    const bool lead = tm.leader();
    for (unsigned int first = 0; first < count; first += (unsigned int)WT_CAP) {
        const unsigned int nw_chunk = count - first < (unsigned int)WT_CAP ? count - first : (unsigned int)WT_CAP;
        if (threadIdx.x == 0) { *head = 0u; *aux = 0u; }
        __syncthreads();
        // ---- A: the walkers' terms
        for (;;) {
            const unsigned long long nb = __ballot(team_active && lead);
            unsigned int base = 0;
            const int leader = __ffsll((long long)nb) - 1;
            if (lane == leader) base = atomicAdd(head, (unsigned int)__popcll(nb));
            base = (unsigned int)__shfl((int)base, leader);
            const unsigned int idx = base + (unsigned int)grp;                       // teams 0..6 of the wave take consecutive walkers
            const bool have = team_active && idx < nw_chunk;
            if (__ballot(have) == 0ull) break;                                        // wave-uniform: the chunk's walkers are handed out
            if (have) {
                const unsigned int item = first + idx;
                tm.terms(c, *tb, row_of(item), 0, 1, 0, GF_PEND_STRIDE);
                double* w = wt + ((size_t)idx * Team9::LANES + tl) * 8;
                w[0] = tm.hs.re.hi; w[1] = tm.hs.re.lo; w[2] = tm.hs.im.hi; w[3] = tm.hs.im.lo;
                w[4] = tm.hn.re.hi; w[5] = tm.hn.re.lo; w[6] = tm.hn.im.hi; w[7] = tm.hn.im.lo;
                if (lead) atomicMax(aux, (unsigned int)__popcll(mask_of(item)));      // the most bins any walker of the chunk brings
            }
        }
        __syncthreads();
        const unsigned int maxb = *aux;
        const unsigned int npairs = nw_chunk * maxb;                                  // rank-major grid of pairs; holes where a walker has fewer bins
        __syncthreads();
        if (threadIdx.x == 0) *head = 0u;
        __syncthreads();
        // ---- B: one energy bin per team and round
        for (;;) {
            const unsigned long long nb = __ballot(team_active && lead);
            unsigned int base = 0;
            const int leader = __ffsll((long long)nb) - 1;
            if (lane == leader) base = atomicAdd(head, (unsigned int)__popcll(nb));
            base = (unsigned int)__shfl((int)base, leader);
            const unsigned int v = base + (unsigned int)grp;
            const bool inrange = team_active && v < npairs;
            if (__ballot(inrange) == 0ull) break;
            if (inrange) {
                const unsigned int idx = v % nw_chunk, rank = v / nw_chunk;
                const unsigned int item = first + idx;
                const unsigned long long im = mask_of(item);
                const unsigned int nbits = (unsigned int)__popcll(im);
                if (rank < nbits) {
                    bool failed = false;
                    if (__atomic_load_n(&ctl[2 * idx + 1], __ATOMIC_RELAXED) == 0u) {   // else: another bin of this walker has failed already
                        const double* w = wt + ((size_t)idx * Team9::LANES + tl) * 8;
                        tm.hs.re.hi = w[0]; tm.hs.re.lo = w[1]; tm.hs.im.hi = w[2]; tm.hs.im.lo = w[3];
                        tm.hn.re.hi = w[4]; tm.hn.re.lo = w[5]; tm.hn.im.hi = w[6]; tm.hn.im.lo = w[7];
                        const int kk = nth_bit_from_top(im, rank);
                        const double res = tm.bin(tb->inv2e[kk], tb->epow[kk]);
                        failed = !(res < 1e-7);                                       // fr.py:493-494 (NaN raises too)
                    }
                    if (lead) {
                        if (failed) atomicOr(&ctl[2 * idx + 1], 1u);
                        __threadfence_block();
                        const unsigned int before = atomicAdd(&ctl[2 * idx], 1u);
                        if (before == nbits - 1u) {                                   // the last pair of the walker to report
                            __threadfence_block();
                            const bool bad = atomicOr(&ctl[2 * idx + 1], 0u) != 0u;
                            ctl[2 * idx] = 0u;
                            ctl[2 * idx + 1] = 0u;
                            done(item, bad);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int NDIM>
__global__ __launch_bounds__(CH_BLOCK) void k_stretch_chain(const ChainArgs s)
{
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    constexpr int TEAMS = CH_WAVES * Team9::PER_WAVE;                       // 56 nine-lane teams
    constexpr int TEAM_DOUBLES = TEAMS * Team9::DOUBLES;
    constexpr int TILE_DOUBLES = CH_WAVES * GF_WAVE * ND;
    // one buffer, two lives: the proposals' rows while a pass of the half-step is evaluated, the teams' slots while its parked
    // proposals are settled (their rows are in `pend_rows` by then)
    __shared__ __attribute__((aligned(16))) double uni[TEAM_DOUBLES > TILE_DOUBLES ? TEAM_DOUBLES : TILE_DOUBLES];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4 + 20];
    __shared__ unsigned long long pk_mask[CH_BLOCK];      // the pass's parked proposals: undecided bins ...
    __shared__ int pk_k[CH_BLOCK];                        // ... walker slot in the active half
    __shared__ unsigned int pk_ctl[2 * CH_BLOCK];         // per listed proposal: parts finished, one of them failed (zero between uses)
    __shared__ unsigned int pk_n, pk_head, pk_aux, lz_n;
    __shared__ __attribute__((aligned(16))) double wt[WT_DOUBLES];    // chain_settle: the Hamiltonian terms of the walkers being settled

    const int chain = blockIdx.x;
    const GfCommon& c = s.commons[s.nmodels > 1 ? chain : 0];
    const GfBsm* __restrict__ tb = s.nmodels > 1 ? s.tbs[chain] : s.tb;
    const double* __restrict__ ptab = s.ptabs[s.nmodels > 1 ? chain : 0];
    double* ttab = ctab + GF_MAX_DIM * 4;
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 18) {                                // as stretch_body: the texture's two projectors
        const int k = threadIdx.x - 64, e = k >> 1;
        const int idx = e == 0 ? 0 : e == 1 ? 4 : e == 2 ? 8 : e <= 4 ? 1 : e <= 6 ? 2 : 5;
        const bool im = e == 4 || e == 6 || e == 8;
        const double* srcp = (k & 1) ? (im ? tb->t2_im : tb->t2_re) : (im ? tb->t1_im : tb->t1_re);
        ttab[k] = srcp[idx];
    }
    for (int i = threadIdx.x; i < 2 * CH_BLOCK; i += CH_BLOCK) pk_ctl[i] = 0u;
    if (threadIdx.x == 0) lz_n = 0u;
    __syncthreads();

    const int ndim = NDIM ? NDIM : c.ndim;
    const int lane = threadIdx.x & (GF_WAVE - 1), wave = threadIdx.x / GF_WAVE;
    const int nw = s.nwalkers, nhalf = nw / 2;
    const uint64_t sid = s.stream_ids ? s.stream_ids[chain] : (uint64_t)chain;
    const uint32_t k0s = (uint32_t)s.seed, k1s = (uint32_t)(s.seed >> 32);
    double* row = uni + (size_t)wave * GF_WAVE * ND + (size_t)lane * ndim;          // (stretch_body: tiles[wave] + lane * ndim)
    double* const my_pend = s.pend_rows + (size_t)chain * nhalf * GF_PEND_STRIDE;    // this chain's parked rows, by walker slot
    // proposals that are REJECTED WHATEVER THEIR VERDICT (the accept test fails even if they are unitary): the chain does not wait
    // for them -- the verdict only feeds the count of proposals the reference would have raised on.  Their rows queue up here
    // and are settled in bulk when the list fills and at the end of the launch.
    double* const lz_rows = s.lazy_rows + (size_t)chain * s.lazy_cap * GF_PEND_STRIDE;
    unsigned long long* const lz_mask = s.lazy_mask + (size_t)chain * s.lazy_cap;
    unsigned long long* const stat = s.stats ? s.stats + (size_t)chain * 8 : nullptr;
    auto flush_lazy = [&]() {                              // workgroup-uniform call; barriers inside
        const unsigned int n = lz_n;
        if (n != 0u) {
            const unsigned long long t0 = wall_clock64();
            if (threadIdx.x == 0) pk_head = 0u;
            __syncthreads();
            chain_settle(uni, wt, n, &pk_head, &pk_aux, pk_ctl, c, tb,
                         [&](unsigned int i) { return lz_rows + (size_t)i * GF_PEND_STRIDE; },
                         [&](unsigned int i) { return lz_mask[i]; },
                         [&](unsigned int, bool bad) { if (bad) atomicAdd(s.flags, 1u); });
            __syncthreads();
            if (threadIdx.x == 0) {
                lz_n = 0u;
                if (stat) { stat[2] += 10ull * (wall_clock64() - t0); stat[5] += n; stat[7] += 1ull; }
            }
            __syncthreads();
        }
    };

    for (int step = 0; step < s.nsteps; ++step) {
        const uint64_t iteration = s.iteration_base + (uint64_t)step;
        const int64_t run_step = s.run_step_base + step;
        const bool store_now = s.store != 0 && s.chain != nullptr && (run_step % s.thin) == 0;
        const int64_t store_index = s.store_base + (run_step + s.thin - 1) / s.thin;
        for (int half = 0; half < 2; ++half) {
            const int cbase = (1 - half) * nhalf;
            for (int kb = 0; kb < nhalf; kb += CH_BLOCK) {
                if (lz_n + (unsigned int)(nhalf < CH_BLOCK ? nhalf : CH_BLOCK) > (unsigned int)s.lazy_cap) flush_lazy();   // (uniform: lz_n is stable between barriers)
                if (threadIdx.x == 0) { pk_n = 0u; pk_head = 0u; }
                __syncthreads();
                const unsigned long long t_pass = wall_clock64();
                const int k = kb + (int)threadIdx.x;
                if (k < nhalf)
                    chain_propose<NDIM>(s, c, tb, (LdsD)ctab, (LdsD)ttab, (LdsD)row, chain, k, half, iteration, store_now, store_index, sid,
                                        my_pend, lz_rows, lz_mask, (LdsU)&pk_n, (LdsI)pk_k, (LdsL)pk_mask, (LdsU)&lz_n);
                __syncthreads();                                      // rows are done with; pk_n is final; this pass's updates are visible
                const unsigned int count = pk_n;
                const unsigned long long t_settle = wall_clock64();
                if (threadIdx.x == 0 && stat) { stat[0] += 10ull * (t_settle - t_pass); stat[6] += 1ull; }
                if (count != 0u) {
                    chain_settle(uni, wt, count, &pk_head, &pk_aux, pk_ctl, c, tb,
                                 [&](unsigned int i) { return my_pend + (size_t)pk_k[i] * GF_PEND_STRIDE; },
                                 [&](unsigned int i) { return pk_mask[i]; },
                                 [&](unsigned int i, bool bad) {
                                     // the walker's accept step, as k_stretch_settle ends it: the accept test itself has passed
                                     const double* prow = my_pend + (size_t)pk_k[i] * GF_PEND_STRIDE;
                                     const int w = half * nhalf + pk_k[i];
                                     const int64_t wi = (int64_t)chain * nw + w;
                                     const double lnq = prow[GF_MAX_DIM];
                                     const double lnk = s.lnp[wi];
                                     const bool accept = !bad;
                                     if (bad) atomicAdd(s.flags, 1u);
                                     double* pw = s.pos + wi * ndim;
                                     if (store_now) {
                                         double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * nw + w) * ndim;
                                         for (int d = 0; d < ndim; ++d) dst[d] = accept ? prow[d] : pw[d];
                                         if (s.lnp_chain) s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * nw + w] = accept ? lnq : lnk;
                                     }
                                     if (accept) {
                                         for (int d = 0; d < ndim; ++d) pw[d] = prow[d];
                                         s.lnp[wi] = lnq;
                                         s.naccept[wi] += 1u;
                                     }
                                 });
                    __syncthreads();                                  // the teams' slots become rows again; the settled walkers are in place
                    if (threadIdx.x == 0 && stat) { stat[1] += 10ull * (wall_clock64() - t_settle); stat[3] += count; stat[4] += 1ull; }
                }
            }
        }
    }
    flush_lazy();
}

#ifdef GF_EXPERIMENTAL_FLOW
// (measured and not kept, round 4 -- compiled only with -DGF_EXPERIMENTAL_FLOW, selected with GF_SAMPLER_CHAIN=3: bitwise the grid
// sampler's chain on the full C5 scan, 480 rounds for 400 half-steps on the heaviest chain -- the dataflow does what it should -- but
// a round costs a full proposal phase AND a full settle slice one after the other (55 + 65 us measured, thread 0 building the unit
// list serially among it), 190 us per half-step against 152 for k_stretch_chain and 117 for the grid kernels; a work-conserving
// version (any wave proposing or settling as needed) is bounded at ~64 us per half-step for such a chain by the emulated-x87 work
// itself: profiles/r04/chain_dataflow_not_kept.txt)
// ---- the same chain as a DATAFLOW inside the workgroup ---------------------------------------------------------------------------
// k_stretch_chain still moves a chain half-step by half-step: when a half-step parks proposals, ALL its walkers wait until the parked
// ones are settled -- and the census says that the chains which bound a scan park ~10 proposals in every half-step and spend half
// their time in that wait (profiles/r04/chain_census.txt).  But a stretch move needs only TWO inputs: the walker's own position and
// its partner's.  So here every walker carries the count of updates it has completed; a walker's next update runs as soon as its own
// previous update and its partner's required one are FINAL, whatever the rest of the ensemble is doing; a parked proposal is settled
// in slices (one team-unit of ~16-23 us per round: the walker's terms, then its undecided bins, highest energy first, stopping at the
// first failure) while everybody who does not depend on it moves on, and the walkers that fell behind catch up at one update per
// round.  A walker's last FLOW_VERS positions are kept (slot = update count mod FLOW_VERS) and nobody runs more than FLOW_AHEAD
// updates ahead of the slowest, so a partner's required version is always still there.  Every update is the computation
// k_stretch / k_stretch_chain make -- same Philox counter (walker slot, half-step), same inputs, same accept rule, same stored
// sample -- only the ORDER in which independent updates are executed differs: the chain is the same bit for bit.
//
// One round:  P  thread w (= walker w) proposes if its inputs are final: decided -> final at once; undecided and rejected either way ->
//                final, verdict counted later (bulk); undecided and the verdict decides -> parked (row kept, walker busy)
//             S  one slice of settling: up to 56 units -- TERMS of a newly parked proposal, or one (walker, bin) PAIR of one whose
//                terms are ready, handed out rank-major -- one per nine-lane team; proposals whose bins are all done (or one failed)
//                are completed: accept step, new version, walker free again
// Barriers only, no wave ever waits for another outside them.
constexpr int FLOW_VERS = 4;
constexpr int FLOW_AHEAD = 2;
constexpr unsigned int FLOW_TERMS = 255u;                 // unit kind: build the Hamiltonian terms (else: the rank of the bin to evaluate)
// e_info: bits 0-6 undecided bins (<= 64), 7-13 ranks handed out, 14-20 pairs done, 21 one failed, 22 terms being built, 23 terms ready, 24 dead
#define FLOW_NBITS(i) ((i) & 127u)
#define FLOW_STARTED(i) (((i) >> 7) & 127u)
#define FLOW_DONE(i) (((i) >> 14) & 127u)
#define FLOW_ONE_STARTED (1u << 7)
#define FLOW_ONE_DONE (1u << 14)
#define FLOW_FAIL (1u << 21)
#define FLOW_TBUSY (1u << 22)
#define FLOW_TREADY (1u << 23)
#define FLOW_DEAD (1u << 24)

// one walker's update u of k_stretch_flow: returns true when the walker is final again (decided here), false when its proposal was parked
template <int NDIM>
__device__ __attribute__((noinline)) bool flow_propose(const ChainArgs& s, const GfCommon& c, const GfBsm* __restrict__ tb, LdsD ctab_l, LdsD ttab_l,
                                                       LdsD row_l, const int chain, const int w, const unsigned int u, const int wj,
                                                       const unsigned int need, const double z, const double u3, double* lz_rows,
                                                       unsigned long long* lz_mask, LdsU new_n, LdsI new_w, LdsL new_mask, LdsU lz_n)
{
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    const double* ctab = (const double*)ctab_l;
    const double* ttab = (const double*)ttab_l;
    double* row = (double*)row_l;
    const int ndim = NDIM ? NDIM : c.ndim;
    const int nw = s.nwalkers;
    const int64_t run_step = s.run_step_base + (int64_t)u;
    const bool store_now = s.store != 0 && s.chain != nullptr && (run_step % s.thin) == 0;
    const int64_t store_index = s.store_base + (run_step + s.thin - 1) / s.thin;
    const size_t vbase = (size_t)chain * FLOW_VERS;
    const double* sk = s.pv + ((vbase + (u & (FLOW_VERS - 1))) * nw + w) * ndim;
    const double* cj = s.pv + ((vbase + (need & (FLOW_VERS - 1))) * nw + wj) * ndim;
    double* nx = s.pv + ((vbase + ((u + 1u) & (FLOW_VERS - 1))) * nw + w) * ndim;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (!NDIM && d >= ndim) break;
        const double cv = cj[d];
        row[d] = fma(-z, cv - sk[d], cv);
    }
    int st;
    unsigned long long pending;
    const double lnq = proposal_lnprob<NDIM, MODE_BSM_GAUSS, 1>(c, tb, ctab, ttab, row, ndim, st, 0, nullptr, pending);
    const double lnk = s.lv[(vbase + (u & (FLOW_VERS - 1))) * nw + w];
    double zp = 1.0;
    for (int d = 1; d < ndim; ++d) zp *= z;
    const double lhs = log(zp / u3);
    bool accept = lhs > lnk - lnq;                        // false for NaN and for lnq = -inf
    if (pending != 0ull && accept) {
        double* dst = s.frows + ((size_t)chain * nw + w) * GF_PEND_STRIDE;
        for (int d = 0; d < ndim; ++d) dst[d] = row[d];
        dst[GF_MAX_DIM] = lnq;
        const unsigned int at = atomicAdd((unsigned int*)new_n, 1u);
        new_w[at] = w;
        new_mask[at] = pending;
        return false;
    }
    if (pending != 0ull) {                                // undecided, rejected either way: only the count wants the verdict
        const unsigned int at = atomicAdd((unsigned int*)lz_n, 1u);
        double* dst = lz_rows + (size_t)at * GF_PEND_STRIDE;
        for (int d = 0; d < ndim; ++d) dst[d] = row[d];
        lz_mask[at] = pending;
    }
    if (st == ST_NON_UNITARY) { accept = false; atomicAdd(s.flags, 1u); }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (!NDIM && d >= ndim) break;
        nx[d] = accept ? row[d] : sk[d];
    }
    s.lv[(vbase + ((u + 1u) & (FLOW_VERS - 1))) * nw + w] = accept ? lnq : lnk;
    if (accept) s.naccept[(int64_t)chain * nw + w] += 1u;
    if (store_now) {
        double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * nw + w) * ndim;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (!NDIM && d >= ndim) break;
            dst[d] = accept ? row[d] : sk[d];
        }
        if (s.lnp_chain) s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * nw + w] = accept ? lnq : lnk;
    }
    return true;
}

template <int NDIM>
__global__ __launch_bounds__(CH_BLOCK) void k_stretch_flow(const ChainArgs s)
{
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    constexpr int TEAMS = CH_WAVES * Team9::PER_WAVE;
    constexpr int TEAM_DOUBLES = TEAMS * Team9::DOUBLES;
    constexpr int TILE_DOUBLES = CH_WAVES * GF_WAVE * ND;
    __shared__ __attribute__((aligned(16))) double uni[TEAM_DOUBLES > TILE_DOUBLES ? TEAM_DOUBLES : TILE_DOUBLES];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4 + 20];
    __shared__ __attribute__((aligned(16))) double wt[WT_DOUBLES];      // chain_settle (the bulk settlement of the count-only proposals)
    __shared__ unsigned int du[CH_BLOCK];                 // updates completed, per walker
    __shared__ unsigned int busy[CH_BLOCK];               // the walker has a parked proposal in flight
    __shared__ int e_w[CH_BLOCK];                         // parked proposals in flight: walker ...
    __shared__ unsigned long long e_mask[CH_BLOCK];       // ... undecided bins ...
    __shared__ unsigned int e_info[CH_BLOCK];             // ... progress (FLOW_* above)
    __shared__ int new_w[CH_BLOCK];
    __shared__ unsigned long long new_mask[CH_BLOCK];
    __shared__ unsigned int unit_e[TEAMS], unit_r[TEAMS];
    __shared__ unsigned int pk_ctl[2 * CH_BLOCK];         // chain_settle's counters (zero between uses)
    __shared__ unsigned int e_n, new_n, unit_n, lz_n, min_u, pk_head, pk_aux;

    const int chain = blockIdx.x;
    const GfCommon& c = s.commons[s.nmodels > 1 ? chain : 0];
    const GfBsm* __restrict__ tb = s.nmodels > 1 ? s.tbs[chain] : s.tb;
    const double* __restrict__ ptab = s.ptabs[s.nmodels > 1 ? chain : 0];
    double* ttab = ctab + GF_MAX_DIM * 4;
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 18) {
        const int k = threadIdx.x - 64, e = k >> 1;
        const int idx = e == 0 ? 0 : e == 1 ? 4 : e == 2 ? 8 : e <= 4 ? 1 : e <= 6 ? 2 : 5;
        const bool im = e == 4 || e == 6 || e == 8;
        const double* srcp = (k & 1) ? (im ? tb->t2_im : tb->t2_re) : (im ? tb->t1_im : tb->t1_re);
        ttab[k] = srcp[idx];
    }
    const int ndim = NDIM ? NDIM : c.ndim;
    const int nw = s.nwalkers, nhalf = nw / 2;
    const unsigned int U_END = (unsigned int)s.nsteps;
    const int tid = threadIdx.x, lane = tid & (GF_WAVE - 1), wave = tid / GF_WAVE;
    const size_t vbase = (size_t)chain * FLOW_VERS;
    // version 0 = the state the launch starts from
    for (int i = tid; i < nw * ndim; i += CH_BLOCK) s.pv[vbase * nw * ndim + i] = s.pos[(size_t)chain * nw * ndim + i];
    for (int i = tid; i < nw; i += CH_BLOCK) s.lv[vbase * nw + i] = s.lnp[(size_t)chain * nw + i];
    du[tid] = tid < nw ? 0u : U_END;
    busy[tid] = 0u;
    pk_ctl[tid] = 0u; pk_ctl[CH_BLOCK + tid] = 0u;
    if (tid == 0) { e_n = 0u; lz_n = 0u; }
    __syncthreads();

    const uint64_t sid = s.stream_ids ? s.stream_ids[chain] : (uint64_t)chain;
    const uint32_t k0s = (uint32_t)s.seed, k1s = (uint32_t)(s.seed >> 32);
    double* row = uni + (size_t)wave * GF_WAVE * ND + (size_t)lane * ndim;
    double* const lz_rows = s.lazy_rows + (size_t)chain * s.lazy_cap * GF_PEND_STRIDE;
    unsigned long long* const lz_mask = s.lazy_mask + (size_t)chain * s.lazy_cap;
    unsigned long long* const stat = s.stats ? s.stats + (size_t)chain * 8 : nullptr;
    const int grp = lane / Team9::LANES;
    const bool team_active = grp < Team9::PER_WAVE;
    const int tl = lane - grp * Team9::LANES;
    const int team = wave * Team9::PER_WAVE + grp;
    auto flush_lazy = [&]() {                              // workgroup-uniform call; barriers inside
        const unsigned int n = lz_n;
        if (n != 0u) {
            const unsigned long long t0 = wall_clock64();
            chain_settle(uni, wt, n, &pk_head, &pk_aux, pk_ctl, c, tb,
                         [&](unsigned int i) { return lz_rows + (size_t)i * GF_PEND_STRIDE; },
                         [&](unsigned int i) { return lz_mask[i]; },
                         [&](unsigned int, bool bad) { if (bad) atomicAdd(s.flags, 1u); });
            __syncthreads();
            if (tid == 0) {
                lz_n = 0u;
                if (stat) { stat[2] += 10ull * (wall_clock64() - t0); stat[5] += n; stat[7] += 1ull; }
            }
            __syncthreads();
        }
    };

    const unsigned int round_limit = 64u * U_END + 4096u;  // a chain needs 2 U_END rounds when nothing is parked: far below this
    for (unsigned int round = 0;; ++round) {
        if (tid == 0) { min_u = U_END; new_n = 0u; }
        __syncthreads();
        if (tid < nw) atomicMin(&min_u, du[tid]);
        __syncthreads();
        const unsigned int mu = min_u;
        if (mu >= U_END && e_n == 0u) break;               // every walker has made its updates and nothing is in flight
        if (round > round_limit) { if (tid == 0) atomicAdd(s.flags + 1, 1u); break; }    // (never: reported by the host)
        if (lz_n + (unsigned int)nw > (unsigned int)s.lazy_cap) flush_lazy();
        const unsigned long long t_p = wall_clock64();
        // ---- P: every walker whose inputs are final makes its next update
        bool fin = false;
        unsigned int u = 0u;
        if (tid < nw && busy[tid] == 0u) {
            u = du[tid];
            if (u < U_END && u <= mu + (unsigned int)FLOW_AHEAD) {
                const int half = tid >= nhalf ? 1 : 0;
                const int k = tid - half * nhalf;
                const uint64_t g = sid * (uint64_t)nhalf + (uint64_t)k;
                const uint64_t ctr = 2 * (s.iteration_base + (uint64_t)u) + (uint64_t)half;
                uint32_t r[4];
                philox_block((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), k0s, k1s, r);
                const int j = (int)(((uint64_t)r[2] * (uint64_t)nhalf) >> 32);
                const int wj = (1 - half) * nhalf + j;
                const unsigned int need = u + (unsigned int)half;     // updates the partner must have completed: its position BEFORE this half-step
                if (du[wj] >= need) {
                    const double u1 = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) * (1.0 / 9007199254740992.0);
                    const double u3 = ((double)r[3] + 0.5) * (1.0 / 4294967296.0);
                    const double zr = fma(s.a - 1.0, u1, 1.0);
                    const double z = zr * zr / s.a;
                    fin = flow_propose<NDIM>(s, c, tb, (LdsD)ctab, (LdsD)ttab, (LdsD)row, chain, tid, u, wj, need, z, u3, lz_rows, lz_mask,
                                             (LdsU)&new_n, (LdsI)new_w, (LdsL)new_mask, (LdsU)&lz_n);
                    if (!fin) busy[tid] = 1u;
                }
            }
        }
        __syncthreads();                                    // every read of du[] of this round is done
        if (fin) du[tid] = u + 1u;
        // the newly parked proposals join the list
        const unsigned int nn = new_n, base = e_n;
        if ((unsigned int)tid < nn) {
            e_w[base + tid] = new_w[tid];
            e_mask[base + tid] = new_mask[tid];
            e_info[base + tid] = (unsigned int)__popcll(new_mask[tid]);
        }
        __syncthreads();
        if (tid == 0) {
            e_n = base + nn;
            if (stat) { stat[0] += 10ull * (wall_clock64() - t_p); stat[6] += 1ull; stat[3] += nn; }
        }
        __syncthreads();
        const unsigned int ne = e_n;
        if (ne == 0u) continue;
        // ---- S: one slice.  Thread 0 hands out the units: the terms of every proposal that has none yet, then pairs rank-major
        const unsigned long long t_s = wall_clock64();
        if (tid == 0) {
            unsigned int n = 0;
            for (unsigned int e = 0; e < ne && n < (unsigned int)TEAMS; ++e)
                if ((e_info[e] & (FLOW_TBUSY | FLOW_TREADY)) == 0u) { unit_e[n] = e; unit_r[n] = FLOW_TERMS; ++n; e_info[e] |= FLOW_TBUSY; }
            bool more = true;
            while (more && n < (unsigned int)TEAMS) {
                more = false;
                for (unsigned int e = 0; e < ne && n < (unsigned int)TEAMS; ++e) {
                    const unsigned int inf = e_info[e];
                    if ((inf & FLOW_TREADY) && !(inf & FLOW_FAIL) && FLOW_STARTED(inf) < FLOW_NBITS(inf)) {
                        unit_e[n] = e; unit_r[n] = FLOW_STARTED(inf); ++n;
                        e_info[e] = inf + FLOW_ONE_STARTED;
                        more = true;
                    }
                }
            }
            unit_n = n;
        }
        __syncthreads();
        if (team_active && (unsigned int)team < unit_n) {
            Team9 tm;
            tm.init(uni + (size_t)team * Team9::DOUBLES, tl);
            const unsigned int e = unit_e[team], r = unit_r[team];
            const int w = e_w[e];
            double* ft = s.fterms + (((size_t)chain * nw + w) * Team9::LANES + tl) * 8;
            if (r == FLOW_TERMS) {
                tm.terms(c, *tb, s.frows + ((size_t)chain * nw + w) * GF_PEND_STRIDE, 0, 1, 0, GF_PEND_STRIDE);
                ft[0] = tm.hs.re.hi; ft[1] = tm.hs.re.lo; ft[2] = tm.hs.im.hi; ft[3] = tm.hs.im.lo;
                ft[4] = tm.hn.re.hi; ft[5] = tm.hn.re.lo; ft[6] = tm.hn.im.hi; ft[7] = tm.hn.im.lo;
                if (tm.leader()) atomicOr(&e_info[e], FLOW_TREADY);
            } else {
                tm.hs.re.hi = ft[0]; tm.hs.re.lo = ft[1]; tm.hs.im.hi = ft[2]; tm.hs.im.lo = ft[3];
                tm.hn.re.hi = ft[4]; tm.hn.re.lo = ft[5]; tm.hn.im.hi = ft[6]; tm.hn.im.lo = ft[7];
                const int kk = nth_bit_from_top(e_mask[e], r);
                const double res = tm.bin(tb->inv2e[kk], tb->epow[kk]);
                if (tm.leader()) {
                    if (!(res < 1e-7)) atomicOr(&e_info[e], FLOW_FAIL);          // fr.py:493-494 (NaN raises too)
                    atomicAdd(&e_info[e], FLOW_ONE_DONE);
                }
            }
        }
        __syncthreads();
        // complete what is settled: every bin done, or one failed (its started pairs have all ended with this slice)
        if ((unsigned int)tid < ne) {
            const unsigned int inf = e_info[tid];
            const bool failed = (inf & FLOW_FAIL) != 0u;
            if ((inf & FLOW_TREADY) && (failed || FLOW_DONE(inf) == FLOW_NBITS(inf)) && FLOW_DONE(inf) == FLOW_STARTED(inf)) {
                const int w = e_w[tid];
                const unsigned int uw = du[w];
                const double* prow = s.frows + ((size_t)chain * nw + w) * GF_PEND_STRIDE;
                const double* old = s.pv + ((vbase + (uw & (FLOW_VERS - 1))) * nw + w) * ndim;
                double* nx = s.pv + ((vbase + ((uw + 1u) & (FLOW_VERS - 1))) * nw + w) * ndim;
                const double lnq = prow[GF_MAX_DIM];
                const double lnk = s.lv[(vbase + (uw & (FLOW_VERS - 1))) * nw + w];
                const bool accept = !failed;                 // the accept test itself passed when the proposal was parked
                if (failed) atomicAdd(s.flags, 1u);
                for (int d = 0; d < ndim; ++d) nx[d] = accept ? prow[d] : old[d];
                s.lv[(vbase + ((uw + 1u) & (FLOW_VERS - 1))) * nw + w] = accept ? lnq : lnk;
                if (accept) s.naccept[(int64_t)chain * nw + w] += 1u;
                const int64_t run_step = s.run_step_base + (int64_t)uw;
                if (s.store != 0 && s.chain != nullptr && (run_step % s.thin) == 0) {
                    const int64_t store_index = s.store_base + (run_step + s.thin - 1) / s.thin;
                    double* dst = s.chain + (((int64_t)chain * s.nstore_cap + store_index) * nw + w) * ndim;
                    for (int d = 0; d < ndim; ++d) dst[d] = accept ? prow[d] : old[d];
                    if (s.lnp_chain) s.lnp_chain[((int64_t)chain * s.nstore_cap + store_index) * nw + w] = accept ? lnq : lnk;
                }
                du[w] = uw + 1u;
                busy[w] = 0u;
                e_info[tid] = inf | FLOW_DEAD;
            }
        }
        __syncthreads();
        if (tid == 0) {                                     // close the gaps
            unsigned int m = 0;
            for (unsigned int e = 0; e < ne; ++e)
                if (!(e_info[e] & FLOW_DEAD)) { if (m != e) { e_w[m] = e_w[e]; e_mask[m] = e_mask[e]; e_info[m] = e_info[e]; } ++m; }
            e_n = m;
            if (stat) { stat[1] += 10ull * (wall_clock64() - t_s); stat[4] += 1ull; }
        }
        __syncthreads();
    }
    __syncthreads();
    flush_lazy();
    // the state the launch ends with
    for (int i = tid; i < nw * ndim; i += CH_BLOCK) s.pos[(size_t)chain * nw * ndim + i] = s.pv[(vbase + (U_END & (FLOW_VERS - 1))) * nw * ndim + i];
    for (int i = tid; i < nw; i += CH_BLOCK) s.lnp[(size_t)chain * nw + i] = s.lv[(vbase + (U_END & (FLOW_VERS - 1))) * nw + i];
}

#endif  // GF_EXPERIMENTAL_FLOW

hipError_t launch_chain(int ndim, int nchains, const ChainArgs& a, hipStream_t st, bool flow)
{
#ifdef GF_EXPERIMENTAL_FLOW
    if (flow) {
        switch (ndim) {
        case 7: hipLaunchKernelGGL(k_stretch_flow<7>, dim3(nchains), dim3(CH_BLOCK), 0, st, a); break;
        case 12: hipLaunchKernelGGL(k_stretch_flow<12>, dim3(nchains), dim3(CH_BLOCK), 0, st, a); break;
        default: hipLaunchKernelGGL(k_stretch_flow<0>, dim3(nchains), dim3(CH_BLOCK), 0, st, a); break;
        }
        return hipGetLastError();
    }
#else
    (void)flow;
#endif
    switch (ndim) {
    case 7: hipLaunchKernelGGL(k_stretch_chain<7>, dim3(nchains), dim3(CH_BLOCK), 0, st, a); break;
    case 12: hipLaunchKernelGGL(k_stretch_chain<12>, dim3(nchains), dim3(CH_BLOCK), 0, st, a); break;
    default: hipLaunchKernelGGL(k_stretch_chain<0>, dim3(nchains), dim3(CH_BLOCK), 0, st, a); break;
    }
    return hipGetLastError();
}

// Ensemble mean of every stored step: chain [nchains][cap][nwalkers][ndim] -> mean [nchains][nstored][ndim]
// (the series emcee's acor works on, golemflavor/mcmc.py:45-51).  One block per (step, chain); thread t sums
// the elements t, t + 256, ... of the step's contiguous nwalkers x ndim block whose column is (t mod ndim)
// -- 256 mod ndim must be 0 for that, otherwise threads stride by the largest multiple of ndim <= 256 -- then
// a fixed LDS tree: deterministic, coalesced, one pass over the chain at HBM rate.
__global__ __launch_bounds__(GF_BLOCK) void k_walker_mean(const double* __restrict__ chain, int64_t cap, int64_t nstored,
                                                          int nwalkers, int ndim, double* __restrict__ mean)
{
    __shared__ double part[GF_BLOCK];
    const int64_t step = blockIdx.x;
    const int ch = blockIdx.y;
    const double* src = chain + ((int64_t)ch * cap + step) * nwalkers * ndim;
    const int stride = (GF_BLOCK / ndim) * ndim;            // threads in use; a multiple of ndim
    const int n = nwalkers * ndim;
    double acc = 0.0;
    if ((int)threadIdx.x < stride)
        for (int i = threadIdx.x; i < n; i += stride) acc += src[i];
    part[threadIdx.x] = (int)threadIdx.x < stride ? acc : 0.0;
    __syncthreads();
    // thread d < ndim folds the partial sums of its column in a fixed order
    if ((int)threadIdx.x < ndim) {
        double sum = 0.0;
        for (int t = threadIdx.x; t < stride; t += ndim) sum += part[t];
        mean[((int64_t)ch * nstored + step) * ndim + threadIdx.x] = sum / (double)nwalkers;
    }
}

template <int NDIM>
hipError_t launch_stretch_n(const GfCommon& c, const GfBsm* tb, const double* ptab, const StretchArgs& a, hipStream_t st)
{
    switch (c.mode) {
    case MODE_PRIOR_ONLY: return launch_stretch_nml<NDIM, MODE_PRIOR_ONLY, 1>(c, tb, ptab, a, st);
    case MODE_SM_GAUSS: return launch_stretch_nml<NDIM, MODE_SM_GAUSS, 1>(c, tb, ptab, a, st);
    default:
        switch (a.lpw) {
        case 2: return launch_stretch_nml<NDIM, MODE_BSM_GAUSS, 2>(c, tb, ptab, a, st);
        case 4: return launch_stretch_nml<NDIM, MODE_BSM_GAUSS, 4>(c, tb, ptab, a, st);
        case 16: return launch_stretch_nml<NDIM, MODE_BSM_GAUSS, 16>(c, tb, ptab, a, st);
        default: return launch_stretch_nml<NDIM, MODE_BSM_GAUSS, 1>(c, tb, ptab, a, st);
        }
    }
}

hipError_t launch_stretch(const GfCommon& c, const GfBsm* tb, const double* ptab, const StretchArgs& a, hipStream_t st)
{
    switch (c.ndim) {
    case 4: return launch_stretch_n<4>(c, tb, ptab, a, st);
    case 6: return launch_stretch_n<6>(c, tb, ptab, a, st);
    case 7: return launch_stretch_n<7>(c, tb, ptab, a, st);
    case 12: return launch_stretch_n<12>(c, tb, ptab, a, st);
    default: return launch_stretch_n<0>(c, tb, ptab, a, st);
    }
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------
struct gf_sampler {
    gf_model* model = nullptr;          // chain 0's model: its stream carries the sampler's launches
    gf_model** models = nullptr;        // [nchains] when every chain has its own posterior, else null
    GfCommon* d_commons = nullptr;      // device copies for k_stretch_multi
    const GfBsm** d_tbs = nullptr;
    const double** d_ptabs = nullptr;
    int nchains = 0, nwalkers = 0, ndim = 0;
    int cus = 256, nbins_max = 0;
    uint64_t seed = 0, iteration = 0;
    double a = 2.0;
    double* d_pos = nullptr;
    double* d_lnp = nullptr;
    uint32_t* d_naccept = nullptr;
    uint32_t* d_flags = nullptr;
    GfArbQueue* d_pq = nullptr;         // BSM posteriors: proposals parked for k_stretch_settle (capacity: one half-step's proposals)
    double* d_pend_rows = nullptr;      // [nchains * nwalkers / 2][GF_PEND_STRIDE]
    unsigned int* d_pend_ctl = nullptr; // [nchains * nwalkers / 2][2]: k_stretch_settle's per-walker counters, zero between uses
    double* d_lazy_rows = nullptr;      // k_stretch_chain: [nchains][lazy_cap][GF_PEND_STRIDE], undecided proposals that are rejected either way
    unsigned long long* d_lazy_mask = nullptr;   // [nchains][lazy_cap]
    int lazy_cap = 0;
    unsigned long long* d_chain_stats = nullptr;   // [nchains][8]: k_stretch_chain's per-chain census (ChainArgs::stats), zeroed by gf_sampler_reset
    double* d_pv = nullptr;                        // k_stretch_flow: [nchains][FLOW_VERS][nwalkers][ndim] versioned positions ...
    double* d_lv = nullptr;                        // ... [nchains][FLOW_VERS][nwalkers] lnprob
    double* d_frows = nullptr;                     // ... [nchains][nwalkers][GF_PEND_STRIDE] parked proposals
    double* d_fterms = nullptr;                    // ... [nchains][nwalkers][72] their Hamiltonian terms
    // launch shape of a BSM sampler on small ensembles: 0 = not decided yet, 1 = one workgroup per chain (k_stretch_chain), 2 = the
    // per-half-step grid kernels + k_stretch_settle.  Decided at the start of every run of 128 steps or more, by timing a block of 16
    // steps of each on the sampler's own chains (gf_sampler_run); GF_SAMPLER_CHAIN=0 / 1 forces one
    int shape = 0;
    double probe_us[2] = {0.0, 0.0};               // what the decision was taken on: us per block of 16 steps, [0] per chain, [1] grid
    double* d_chain = nullptr;
    double* d_lnp_chain = nullptr;
    int64_t nstore_cap = 0, nstored = 0;
    int64_t steps_since_reset = 0;
    uint64_t* d_stream_ids = nullptr;   // per-chain random stream ids (gf_sampler_set_stream_ids), else null
    StepState* d_state = nullptr;
    StepState h_state = {};
    // captured graph of GRAPH_STEPS steps (2 nodes per step + one tick), valid for the chain pointers it was built with
    hipGraphExec_t graph = nullptr;
    double* graph_chain = nullptr;
    double* graph_lnp_chain = nullptr;
    int64_t graph_cap = -1;
    int graph_has_chain = -1;
    // Blocks of steps in flight.  An event is recorded behind every block a run enqueues (a graph replay of 16 steps, or up to 64
    // eager steps), in a ring of FLIGHT slots: block k is not enqueued before block k - FLIGHT has completed, so the host never runs
    // more than FLIGHT blocks (~2 000 AQL packets) ahead of the GPU however long the run is.  gf_sampler_run_to_host hangs its
    // read-back on the same events (`sink`): a completed block's stored steps are copied to the host before its slot is reused.
    static constexpr int FLIGHT = 8;
    hipEvent_t flight_ev[FLIGHT] = {};
    int64_t flight_nstored[FLIGHT] = {};     // stored steps of every chain that are final once the slot's event has passed
    int64_t flight_enq = 0, flight_done = 0; // blocks of the current run: enqueued / consumed (waited for and, with a sink, copied)
    struct HostSink* sink = nullptr;
};
// gf_sampler_run_to_host's destination
struct HostSink {
    double* chain; double* lnp; int64_t total;     // host arrays of `total` stored steps per chain
    void* copy_stream; int device;
    struct gf_d2h_pipe* pipe = nullptr;            // ONE read-back pipeline for the whole run (gf_capi.hip)
    int64_t copied = 0;                            // stored steps whose copy has been issued
    int rc = GF_OK;
    std::chrono::steady_clock::time_point last_event;   // when the newest consumed block was seen complete
    // where the host thread's time went (diagnostics: gf_internal_run_to_host_times): [0] total / [1] longest call issuing a block's
    // copy, [2] total / [3] longest wait for a block to complete, [4] blocks, [5] total / [6] longest enqueue of a block of steps,
    // [7] from the call's entry to the first block's mark (chain buffer, graph capture, the launch-shape probe)
    double t[8] = {};
    std::chrono::steady_clock::time_point t_entry;
};
double g_last_run_to_host_times[8] = {};
double g_last_run_prologue[4] = {};             // the last gf_sampler_run: [0] growing the chain buffers, [1] capturing + instantiating the graph (seconds)

// accessors implemented in gf_capi.hip (gf_model is private to it)
extern "C" {
int gf_model_internal(gf_model* m, const GfCommon** c, const GfBsm** d_bsm, const double** d_ptab, void** stream,
                      int* device);
int gf_model_constants(gf_model* m, const GfCommon** c, const GfBsm** d_bsm, const double** d_ptab, int* device, int* cus,
                       int* nbins);
void gf_internal_set_error(const char* msg);
int gf_model_lnprob_on(gf_model* m, void* stream, const double* d_theta, int layout, int64_t n, double* d_lnprob,
                       double* d_fr, int32_t* d_status);
void gf_internal_full_arbitration_grids(int device, void* stream, int on);
int gf_internal_borrow_stream(int device, void** stream);
int gf_internal_borrow_copy_stream(int device, void** stream);
void gf_internal_return_copy_stream(int device, void* stream);
void gf_internal_return_stream(int device, void* stream);
int gf_model_propagate_on(gf_model* m, void* stream, const double* d_theta, int layout, int64_t n, double* d_fr,
                          int32_t* d_status);
int gf_internal_check_overflow(int device, void* stream);
int gf_internal_d2h_gated(int device, void* stream, void* dst_host, const void* src_dev, size_t bytes,
                          int (*gate)(void* ctx, size_t upto), void* gate_ctx);
int gf_internal_d2h_2d(int device, void* stream, void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width,
                       size_t height);
struct gf_d2h_pipe;
int gf_internal_d2h_pipe_open(int device, void* stream, gf_d2h_pipe** out);
int gf_internal_d2h_pipe_rows(gf_d2h_pipe* p, void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height);
int gf_internal_d2h_pipe_close(gf_d2h_pipe* p);
}

namespace {
thread_local char g_serr[256] = "";     // composed here, published through gf_last_hip_error()
int sfail(hipError_t e, const char* what)
{
    std::snprintf(g_serr, sizeof(g_serr), "%s: %s", what, hipGetErrorString(e));
    gf_internal_set_error(g_serr);
    return GF_ERR_HIP;
}
#define GFS_HIP(call)                              \
    do {                                           \
        hipError_t e_ = (call);                    \
        if (e_ != hipSuccess) return sfail(e_, #call); \
    } while (0)
}  // namespace

extern "C" {


int gf_sampler_create(gf_model* m, int nchains, int nwalkers, uint64_t seed, double a, gf_sampler** out)
{
    if (!m || !out || nchains < 1 || nwalkers < 2 || (nwalkers & 1) || !(a > 1.0)) return GF_ERR_INVALID_ARG;
    *out = nullptr;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(m, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    if (nwalkers < 2 * c->ndim) return GF_ERR_INVALID_ARG;        // emcee's own requirement
    gf_sampler* s = new (std::nothrow) gf_sampler();
    if (!s) return GF_ERR_ALLOC;
    s->model = m; s->nchains = nchains; s->nwalkers = nwalkers; s->ndim = c->ndim; s->seed = seed; s->a = a;
    {
        const GfCommon* cc; const GfBsm* tbb; const double* pt; int dev;
        gf_model_constants(m, &cc, &tbb, &pt, &dev, &s->cus, &s->nbins_max);
    }
    const size_t nw = (size_t)nchains * nwalkers;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_pos, sizeof(double) * nw * s->ndim);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_lnp, sizeof(double) * nw);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_naccept, sizeof(uint32_t) * nw);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_flags, sizeof(uint32_t) * 4);
    if (e == hipSuccess && c->mode == MODE_BSM_GAUSS) {
        const size_t nprop = (size_t)nchains * (nwalkers / 2);
        e = hipMalloc((void**)&s->d_pq, sizeof(GfArbQueue) + sizeof(GfArbItem) * nprop);
        if (e == hipSuccess) e = hipMalloc((void**)&s->d_pend_rows, sizeof(double) * nprop * GF_PEND_STRIDE);
        if (e == hipSuccess) e = hipMalloc((void**)&s->d_pend_ctl, sizeof(unsigned int) * 2 * nprop);
        if (e == hipSuccess) e = hipMemsetAsync(s->d_pend_ctl, 0, sizeof(unsigned int) * 2 * nprop, (hipStream_t)stream);
        if (e == hipSuccess) {
            GfArbQueue ah;
            std::memset(&ah, 0, sizeof(ah));
            ah.cap = (unsigned int)nprop;
            e = hipMemcpyAsync(s->d_pq, &ah, offsetof(GfArbQueue, items), hipMemcpyHostToDevice, (hipStream_t)stream);
            if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);      // `ah` is a local
        }
    }
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_state, sizeof(StepState));
    // every transfer of this file goes through the sampler's stream: a synchronous (null-stream) hipMemcpy / hipMemset
    // issued while ANOTHER host thread is capturing its sampler's graph fails and poisons that capture on this runtime
    hipStream_t st0 = (hipStream_t)stream;
    if (e == hipSuccess) e = hipMemsetAsync(s->d_state, 0, sizeof(StepState), st0);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_naccept, 0, sizeof(uint32_t) * nw, st0);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_flags, 0, sizeof(uint32_t) * 4, st0);
    // device copies of the constants for the kernels that take them by pointer (k_stretch_persist)
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_commons, sizeof(GfCommon));
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_ptabs, sizeof(void*));
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_commons, c, sizeof(GfCommon), hipMemcpyHostToDevice, st0);
    if (e == hipSuccess) e = hipMemcpyAsync((void*)s->d_ptabs, &ptab, sizeof(void*), hipMemcpyHostToDevice, st0);
    if (e == hipSuccess) e = hipStreamSynchronize(st0);
    if (e != hipSuccess) { int rc = sfail(e, "gf_sampler_create"); gf_sampler_destroy(s); return rc; }
    *out = s;
    return GF_OK;
}

void gf_sampler_destroy(gf_sampler* s)
{
    if (!s) return;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) == GF_OK) {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize((hipStream_t)stream);
    }
    if (s->d_pos) (void)hipFree(s->d_pos);
    if (s->d_lnp) (void)hipFree(s->d_lnp);
    if (s->d_naccept) (void)hipFree(s->d_naccept);
    if (s->d_flags) (void)hipFree(s->d_flags);
    if (s->d_pq) (void)hipFree(s->d_pq);
    if (s->d_pend_rows) (void)hipFree(s->d_pend_rows);
    if (s->d_pend_ctl) (void)hipFree(s->d_pend_ctl);
    if (s->d_lazy_rows) (void)hipFree(s->d_lazy_rows);
    if (s->d_lazy_mask) (void)hipFree(s->d_lazy_mask);
    if (s->d_chain_stats) (void)hipFree(s->d_chain_stats);
    if (s->d_pv) (void)hipFree(s->d_pv);
    if (s->d_lv) (void)hipFree(s->d_lv);
    if (s->d_frows) (void)hipFree(s->d_frows);
    if (s->d_fterms) (void)hipFree(s->d_fterms);
    if (s->d_state) (void)hipFree(s->d_state);
    if (s->graph) (void)hipGraphExecDestroy(s->graph);
    for (int i = 0; i < gf_sampler::FLIGHT; ++i) if (s->flight_ev[i]) (void)hipEventDestroy(s->flight_ev[i]);
    if (s->d_chain) (void)hipFree(s->d_chain);
    if (s->d_lnp_chain) (void)hipFree(s->d_lnp_chain);
    if (s->d_commons) (void)hipFree(s->d_commons);
    if (s->d_stream_ids) (void)hipFree(s->d_stream_ids);
    if (s->d_tbs) (void)hipFree((void*)s->d_tbs);
    if (s->d_ptabs) (void)hipFree((void*)s->d_ptabs);
    delete[] s->models;
    delete s;
}

// One ensemble per model: chain ch samples the posterior of models[ch].  All models must live on the same
// device and share ndim and mode (one kernel instance); everything else -- priors, fixed values, best fit,
// smearing, texture, dimension, binning -- may differ.  The models must outlive the sampler.
int gf_sampler_create_multi(gf_model* const* models, int nchains, int nwalkers, uint64_t seed, double a, gf_sampler** out)
{
    if (!models || !out || nchains < 1 || nchains > 65535) return GF_ERR_INVALID_ARG;   // blockIdx.y = chain
    *out = nullptr;
    for (int ch = 0; ch < nchains; ++ch)
        if (!models[ch]) return GF_ERR_INVALID_ARG;
    const GfCommon* c0; const GfBsm* tb0; const double* ptab0; void* stream0; int device0;
    if (gf_model_internal(models[0], &c0, &tb0, &ptab0, &stream0, &device0) != GF_OK) return GF_ERR_INVALID_ARG;
    int nbins_max = 0;
    GfCommon* hc = new (std::nothrow) GfCommon[nchains];
    const GfBsm** htb = new (std::nothrow) const GfBsm*[nchains];
    const double** hpt = new (std::nothrow) const double*[nchains];
    gf_model** keep = new (std::nothrow) gf_model*[nchains];
    auto cleanup = [&]() { delete[] hc; delete[] htb; delete[] hpt; };
    if (!hc || !htb || !hpt || !keep) { cleanup(); delete[] keep; return GF_ERR_ALLOC; }
    for (int ch = 0; ch < nchains; ++ch) {
        const GfCommon* c; int device, cus, nbins;
        if (gf_model_constants(models[ch], &c, &htb[ch], &hpt[ch], &device, &cus, &nbins) != GF_OK || device != device0 ||
            c->ndim != c0->ndim || c->mode != c0->mode) {
            std::snprintf(g_serr, sizeof(g_serr), "gf_sampler_create_multi: model %d differs from model 0 in device, ndim or mode", ch);
            gf_internal_set_error(g_serr);
            cleanup(); delete[] keep;
            return GF_ERR_INVALID_ARG;
        }
        hc[ch] = *c;
        keep[ch] = models[ch];
        if (nbins > nbins_max) nbins_max = nbins;
    }
    gf_sampler* s = nullptr;
    int rc = gf_sampler_create(models[0], nchains, nwalkers, seed, a, &s);
    if (rc != GF_OK) { cleanup(); delete[] keep; return rc; }
    s->models = keep;
    s->nbins_max = nbins_max;
    (void)hipFree(s->d_commons); s->d_commons = nullptr;            // the one-entry tables of gf_sampler_create
    (void)hipFree((void*)s->d_ptabs); s->d_ptabs = nullptr;
    hipError_t e = hipMalloc((void**)&s->d_commons, sizeof(GfCommon) * nchains);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_tbs, sizeof(void*) * nchains);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_ptabs, sizeof(void*) * nchains);
    hipStream_t st0 = (hipStream_t)stream0;
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_commons, hc, sizeof(GfCommon) * nchains, hipMemcpyHostToDevice, st0);
    if (e == hipSuccess) e = hipMemcpyAsync((void*)s->d_tbs, htb, sizeof(void*) * nchains, hipMemcpyHostToDevice, st0);
    if (e == hipSuccess) e = hipMemcpyAsync((void*)s->d_ptabs, hpt, sizeof(void*) * nchains, hipMemcpyHostToDevice, st0);
    if (e == hipSuccess) e = hipStreamSynchronize(st0);               // the host arrays are freed next
    cleanup();
    if (e != hipSuccess) { rc = sfail(e, "gf_sampler_create_multi"); gf_sampler_destroy(s); return rc; }
    *out = s;
    return GF_OK;
}

// Random stream of every chain (default: the chain's index in this sampler).  ids [nchains]; call before the first run.
int gf_sampler_set_stream_ids(gf_sampler* s, const uint64_t* ids)
{
    if (!s || !ids) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    GFS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    GFS_HIP(hipStreamSynchronize(st));
    if (!s->d_stream_ids) GFS_HIP(hipMalloc((void**)&s->d_stream_ids, sizeof(uint64_t) * (size_t)s->nchains));
    GFS_HIP(hipMemcpyAsync(s->d_stream_ids, ids, sizeof(uint64_t) * (size_t)s->nchains, hipMemcpyHostToDevice, st));
    GFS_HIP(hipStreamSynchronize(st));
    if (s->graph) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; }     // its kernel arguments froze the old pointer
    return GF_OK;
}

// p0: [nchains][nwalkers][ndim] host; evaluates lnprob of the start positions on the device.
int gf_sampler_set_state(gf_sampler* s, const double* pos)
{
    if (!s || !pos) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device);
    GFS_HIP(hipSetDevice(device));
    const size_t nw = (size_t)s->nchains * s->nwalkers;
    GFS_HIP(hipMemcpyAsync(s->d_pos, pos, sizeof(double) * nw * s->ndim, hipMemcpyHostToDevice, (hipStream_t)stream));
    // BSM posteriors: with the unitarity status, so that a start position the reference would have raised on is treated as such
    int32_t* d_st = nullptr;
    if (c->mode == MODE_BSM_GAUSS) GFS_HIP(hipMalloc((void**)&d_st, sizeof(int32_t) * nw));
    int rc = GF_OK;
    if (!s->models) {
        rc = gf_model_lnprob_on(s->model, stream, s->d_pos, GF_LAYOUT_AOS, (int64_t)nw, s->d_lnp, nullptr, d_st);
    } else {
        for (int ch = 0; ch < s->nchains && rc == GF_OK; ++ch)         // every chain's own posterior, on the sampler's stream
            rc = gf_model_lnprob_on(s->models[ch], stream, s->d_pos + (size_t)ch * s->nwalkers * s->ndim, GF_LAYOUT_AOS,
                                    s->nwalkers, s->d_lnp + (size_t)ch * s->nwalkers, nullptr, d_st ? d_st + (size_t)ch * s->nwalkers : nullptr);
    }
    hipError_t e = hipSuccess;
    if (rc == GF_OK && d_st) {
        hipLaunchKernelGGL(k_fix_start, dim3((unsigned)((nw + 255) / 256 < 1024 ? (nw + 255) / 256 : 1024)), dim3(256), 0, (hipStream_t)stream,
                           d_st, (int64_t)nw, s->d_lnp, s->d_flags);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
    if (d_st) (void)hipFree(d_st);
    if (rc != GF_OK) return rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return sfail(e, "gf_sampler_set_state");
    return gf_internal_check_overflow(device, stream);
}

int gf_sampler_reset(gf_sampler* s)
{
    if (!s) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device);
    GFS_HIP(hipSetDevice(device));
    GFS_HIP(hipStreamSynchronize((hipStream_t)stream));
    GFS_HIP(hipMemsetAsync(s->d_naccept, 0, sizeof(uint32_t) * (size_t)s->nchains * s->nwalkers, (hipStream_t)stream));
    GFS_HIP(hipMemsetAsync(s->d_flags, 0, sizeof(uint32_t) * 4, (hipStream_t)stream));
    if (s->d_chain_stats) GFS_HIP(hipMemsetAsync(s->d_chain_stats, 0, sizeof(unsigned long long) * 8 * (size_t)s->nchains, (hipStream_t)stream));
    GFS_HIP(hipStreamSynchronize((hipStream_t)stream));
    s->nstored = 0;
    s->steps_since_reset = 0;
    return GF_OK;
}

namespace {
// the stored steps [sink->copied, upto) of every chain -> host: their chunks are ISSUED into the sink's pipe (pinned ring, copy stream); the
// host threads empty them into the destination while the caller goes on
int sink_copy(gf_sampler* s, int64_t upto)
{
    HostSink* k = s->sink;
    if (!k || k->rc != GF_OK || upto <= k->copied) return k ? k->rc : GF_OK;
    if (upto > k->total) upto = k->total;
    const size_t row = sizeof(double) * (size_t)s->nwalkers * s->ndim, lrow = sizeof(double) * (size_t)s->nwalkers;
    const size_t prev = (size_t)k->copied, now = (size_t)(upto - k->copied);
    int rc;
    const auto t_in = std::chrono::steady_clock::now();
    struct Clock { HostSink* k; std::chrono::steady_clock::time_point t0;
                   ~Clock() { const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); k->t[0] += d; if (d > k->t[1]) k->t[1] = d; k->t[4] += 1.0; } } clock_{k, t_in};
    if (k->pipe) {
        rc = gf_internal_d2h_pipe_rows(k->pipe, reinterpret_cast<char*>(k->chain) + row * prev, row * (size_t)k->total,
                                       reinterpret_cast<const char*>(s->d_chain) + row * prev, row * (size_t)s->nstore_cap, row * now,
                                       (size_t)s->nchains);
        if (rc == GF_OK && k->lnp)
            rc = gf_internal_d2h_pipe_rows(k->pipe, reinterpret_cast<char*>(k->lnp) + lrow * prev, lrow * (size_t)k->total,
                                           reinterpret_cast<const char*>(s->d_lnp_chain) + lrow * prev, lrow * (size_t)s->nstore_cap, lrow * now,
                                           (size_t)s->nchains);
    } else {                                               // GF_RUN_TO_HOST_NO_PIPE (A/B): a pipeline of its own per block, as before the pipe
        rc = gf_internal_d2h_2d(k->device, k->copy_stream, reinterpret_cast<char*>(k->chain) + row * prev, row * (size_t)k->total,
                                reinterpret_cast<const char*>(s->d_chain) + row * prev, row * (size_t)s->nstore_cap, row * now,
                                (size_t)s->nchains);
        if (rc == GF_OK && k->lnp)
            rc = gf_internal_d2h_2d(k->device, k->copy_stream, reinterpret_cast<char*>(k->lnp) + lrow * prev, lrow * (size_t)k->total,
                                    reinterpret_cast<const char*>(s->d_lnp_chain) + lrow * prev, lrow * (size_t)s->nstore_cap, lrow * now,
                                    (size_t)s->nchains);
    }
    k->rc = rc;
    k->copied = upto;
    return rc;
}

// Block j of the current run: has it completed?  wait = true blocks until it has.  A completed block is consumed -- its stored
// steps go to the sink, if there is one -- and its slot is free again.  Blocks are consumed in order.
hipError_t flight_consume(gf_sampler* s, bool wait)
{
    const int slot = (int)(s->flight_done % gf_sampler::FLIGHT);
    const auto t_w = std::chrono::steady_clock::now();
    hipError_t e = wait ? hipEventSynchronize(s->flight_ev[slot]) : hipEventQuery(s->flight_ev[slot]);
    if (wait && s->sink) {
        const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w).count();
        s->sink->t[2] += d; if (d > s->sink->t[3]) s->sink->t[3] = d;
    }
    if (e != hipSuccess) return e;                                  // hipErrorNotReady: still running
    if (s->sink) { s->sink->last_event = std::chrono::steady_clock::now(); (void)sink_copy(s, s->flight_nstored[slot]); }
    ++s->flight_done;
    return hipSuccess;
}

// before block `flight_enq` is enqueued: at most FLIGHT - 1 earlier blocks may still be running; and whatever has completed
// meanwhile is consumed on the way (with a sink: copied while the GPU works on the blocks behind it)
hipError_t flight_admit(gf_sampler* s)
{
    while (s->flight_enq - s->flight_done >= gf_sampler::FLIGHT) {
        const hipError_t e = flight_consume(s, true);
        if (e != hipSuccess) return e;
    }
    while (s->sink && s->flight_done < s->flight_enq) {
        const hipError_t e = flight_consume(s, false);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// an event behind the block just enqueued; `nstored` stored steps of every chain are final once it has passed
hipError_t flight_mark(gf_sampler* s, hipStream_t st, int64_t nstored)
{
    const int slot = (int)(s->flight_enq % gf_sampler::FLIGHT);
    if (s->sink && s->sink->t[7] == 0.0) s->sink->t[7] = std::chrono::duration<double>(std::chrono::steady_clock::now() - s->sink->t_entry).count();
    hipEvent_t& e = s->flight_ev[slot];
    if (!e) { const hipError_t ec = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (ec != hipSuccess) { e = nullptr; return ec; } }
    const hipError_t er = hipEventRecord(e, st);
    if (er != hipSuccess) return er;
    s->flight_nstored[slot] = nstored;
    ++s->flight_enq;
    return hipSuccess;
}
}  // namespace

// Advance every ensemble by nsteps stretch-move steps (2 launches each), asynchronously on the model's
// stream.  store != 0 appends every `thin`-th step to the device chain (capacity grows as needed).
int gf_sampler_run(gf_sampler* s, int64_t nsteps, int thin, int store)
{
    if (!s || nsteps < 0 || thin < 1) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device);
    GFS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nw = (size_t)s->nchains * s->nwalkers;
    if (store) {
        const int64_t need = s->nstored + (nsteps + thin - 1) / thin;
        if (need > s->nstore_cap) {
            // grow: chains are [chain][slot][walker][dim]; a new capacity changes the chain stride, so repack
            int64_t cap = s->nstore_cap ? s->nstore_cap : 64;
            while (cap < need) cap *= 2;
            double *nc = nullptr, *nl = nullptr;
            GFS_HIP(hipStreamSynchronize(st));
            const auto t_grow = std::chrono::steady_clock::now();
            struct Grow { std::chrono::steady_clock::time_point t0; ~Grow() { g_last_run_prologue[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } grow_{t_grow};
            GFS_HIP(hipMalloc((void**)&nc, sizeof(double) * nw * s->ndim * cap));
            {
                hipError_t e_ = hipMalloc((void**)&nl, sizeof(double) * nw * cap);
                if (e_ != hipSuccess) { (void)hipFree(nc); return sfail(e_, "hipMalloc(lnprob chain)"); }
            }
            if (s->nstored > 0) {
                // every chain's stored prefix in one strided copy: row = chain, pitch = old / new chain stride
                const size_t row = sizeof(double) * s->nwalkers * s->ndim, lrow = sizeof(double) * s->nwalkers;
                hipError_t e_ = hipMemcpy2DAsync(nc, row * cap, s->d_chain, row * s->nstore_cap, row * s->nstored, s->nchains,
                                                 hipMemcpyDeviceToDevice, st);
                if (e_ == hipSuccess)
                    e_ = hipMemcpy2DAsync(nl, lrow * cap, s->d_lnp_chain, lrow * s->nstore_cap, lrow * s->nstored, s->nchains,
                                          hipMemcpyDeviceToDevice, st);
                if (e_ == hipSuccess) e_ = hipStreamSynchronize(st);                  // the old buffers are freed next
                if (e_ != hipSuccess) { (void)hipFree(nc); (void)hipFree(nl); return sfail(e_, "chain repack"); }
            }
            if (s->d_chain) (void)hipFree(s->d_chain);
            if (s->d_lnp_chain) (void)hipFree(s->d_lnp_chain);
            s->d_chain = nc; s->d_lnp_chain = nl; s->nstore_cap = cap;
            // the captured graph froze the old buffers and their capacity stride in its kernel arguments
            if (s->graph) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; }
        }
    }
    // device-side step counters for this run
    StepState& hs = s->h_state;                   // member: outlives the asynchronous upload
    hs.iteration_base = s->iteration; hs.run_step_base = 0; hs.store_base = s->nstored;
    hs.store = store ? 1 : 0; hs.thin = thin;
    GFS_HIP(hipStreamSynchronize(st));          // earlier runs must be done with the counters
    s->flight_enq = s->flight_done = 0;         // (so nothing of an earlier run is in flight either)
    GFS_HIP(hipMemcpyAsync(s->d_state, &hs, sizeof(hs), hipMemcpyHostToDevice, st));
    StretchArgs a;
    a.state = s->d_state;
    a.pos = s->d_pos; a.lnp = s->d_lnp; a.naccept = s->d_naccept; a.flags = s->d_flags;
    a.pq = s->d_pq; a.pend_rows = s->d_pend_rows;
    a.chain = store ? s->d_chain : nullptr;
    a.lnp_chain = store ? s->d_lnp_chain : nullptr;
    a.nstore_cap = s->nstore_cap; a.seed = s->seed; a.nchains = s->nchains; a.nwalkers = s->nwalkers; a.a = s->a;
    a.commons = s->models ? s->d_commons : nullptr; a.tbs = s->d_tbs; a.ptabs = s->d_ptabs;
    a.nbins_max = s->nbins_max;
    a.stream_ids = s->d_stream_ids;
    a.lpw = lanes_per_walker(c->mode, (int64_t)s->nchains * (s->nwalkers / 2), s->nbins_max, s->cus);
    GfSettleArgs sa;
    sa.state = s->d_state; sa.pq = s->d_pq; sa.pend_rows = s->d_pend_rows; sa.ctl = s->d_pend_ctl; sa.pos = s->d_pos; sa.lnp = s->d_lnp; sa.naccept = s->d_naccept;
    sa.flags = s->d_flags; sa.chain = a.chain; sa.lnp_chain = a.lnp_chain; sa.nstore_cap = s->nstore_cap; sa.nchains = s->nchains;
    sa.nwalkers = s->nwalkers; sa.half = 0; sa.step_offset = 0; sa.ndim = s->ndim; sa.commons = s->d_commons; sa.tbs = s->d_tbs; sa.tb = tb;
    sa.multi = s->models ? 1 : 0;
    auto steps = [&](int count) -> hipError_t {       // `count` steps relative to the current base, then tick
        for (int i = 0; i < count; ++i) {
            a.step_offset = i;
            for (int half = 0; half < 2; ++half) {
                a.half = half;
                hipError_t e = launch_stretch(*c, tb, ptab, a, st);
                if (e != hipSuccess) return e;
                if (c->mode == MODE_BSM_GAUSS) {
                    // the proposals whose unitarity the half-step could not settle: exact verdict, then their accept step
                    sa.half = half; sa.step_offset = i;
                    e = gf_launch_stretch_settle(sa, s->cus, st);
                    if (e != hipSuccess) return e;
                }
            }
        }
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, st, s->d_state, count);
        return hipGetLastError();
    };
    // small ensembles of a PRIOR_ONLY / SM_GAUSS posterior: one workgroup per ensemble, the whole run in one launch
    {
        int threads = 0;
        size_t lds = 0;
        int workers = 0;
        persist_geometry(s->nwalkers, s->ndim, &threads, &lds, &workers, s->nchains, s->cus);
        const char* env = gf_internal_env("GF_SAMPLER_PERSIST", 0);          // "0": always the per-half-step grid kernels
        if (c->mode != MODE_BSM_GAUSS && lds > 0 && !(env && env[0] == '0')) {
            PersistArgs pa;
            pa.commons = s->d_commons; pa.ptabs = s->d_ptabs; pa.nmodels = s->models ? s->nchains : 1;
            pa.nwalkers = s->nwalkers; pa.pos = s->d_pos; pa.lnp = s->d_lnp; pa.naccept = s->d_naccept;
            pa.chain = store ? s->d_chain : nullptr; pa.lnp_chain = store ? s->d_lnp_chain : nullptr;
            pa.nstore_cap = s->nstore_cap; pa.seed = s->seed; pa.thin = thin; pa.store = store ? 1 : 0; pa.a = s->a;
            pa.stream_ids = s->d_stream_ids;
            pa.workers = workers;
            constexpr int64_t CHUNK = 1 << 16;                          // steps per launch: bounds a kernel's run time
            int64_t done_p = 0;
            while (done_p < nsteps) {
                // every launch but the last covers a whole multiple of `thin` steps, so that the kernel's
                // (step % thin) == 0 test, which counts from the launch's first step, stays aligned with the run
                int64_t count = nsteps - done_p < CHUNK ? nsteps - done_p : CHUNK;
                if (count < nsteps - done_p) {
                    const int64_t whole = (count / thin) * thin;
                    count = whole > 0 ? whole : thin;
                    if (count > nsteps - done_p) count = nsteps - done_p;      // thin longer than what is left
                }
                pa.iteration_base = s->iteration + (uint64_t)done_p;
                pa.store_base = s->nstored + (done_p + thin - 1) / thin;
                pa.nsteps = count;
                hipError_t e = launch_persist(c->mode, s->ndim, s->nchains, threads, lds, pa, st);
                if (e != hipSuccess) return sfail(e, "persistent stretch launch");
                done_p += count;
            }
            s->iteration += (uint64_t)nsteps;
            s->steps_since_reset += nsteps;
            if (store) s->nstored += (nsteps + thin - 1) / thin;
            return GF_OK;
        }
    }
    // BSM posteriors on ensembles of up to 1024 walkers have two launch shapes with the same chain, bit for bit:
    //   per chain   one workgroup owns a chain for a block of 16 steps and settles its own parked proposals (k_stretch_chain): a chain that
    //               parks nothing never waits for one that does -- 21 us per half-step where nothing is parked (C5: 256 x 512 walkers);
    //   grid        one launch per half-step for all chains + k_stretch_settle on the whole GPU (round 3): every chain waits for the
    //               slowest parked proposal, but a chain that parks TEN proposals per half-step (the top-scale grid points of C5) gets
    //               3 584 teams for them instead of its workgroup's 56 -- 115 us per half-step against 150.
    // Which is faster depends on where the chains live NOW (a burn-in that starts below the failing region and drifts into it parks
    // nothing at first), so every run of 128 steps or more times one block of each shape on its own chains at its start (the second
    // block of two, the first warms the kernel up; the blocks are the run's own steps, nothing is computed twice) and takes the faster
    // for the rest of the run; shorter runs take the last decision (none yet: per chain).  GF_SAMPLER_CHAIN=1 / 0 forces a shape.
    int64_t done = 0;
    const bool small_bsm = c->mode == MODE_BSM_GAUSS && s->nwalkers / 2 <= 512;
    constexpr int64_t CHAIN_STEPS = 16;                                  // steps per launch: the granule of the overlapped read-back
    ChainArgs ca = {};
    bool flow = false;                                                   // per chain as a dataflow (k_stretch_flow) or half-step by half-step (k_stretch_chain)
    auto chain_block = [&](int64_t count) -> int {                       // `count` steps from `done` on, one launch
        hipError_t e = flight_admit(s);
        if (e != hipSuccess) return sfail(e, "block in flight");
        ca.iteration_base = s->iteration + (uint64_t)done;
        ca.run_step_base = done;
        ca.nsteps = (int32_t)count;
        e = launch_chain(s->ndim, s->nchains, ca, st, flow);
        if (e != hipSuccess) return sfail(e, "chain launch");
        done += count;
        e = flight_mark(s, st, store ? s->nstored + (done + thin - 1) / thin : s->nstored);
        if (e != hipSuccess) return sfail(e, "hipEventRecord");
        return GF_OK;
    };
    auto grid_block = [&](int count) -> int {                            // the same with the grid kernels, launched one by one
        hipError_t e = flight_admit(s);
        if (e != hipSuccess) return sfail(e, "block in flight");
        e = steps(count);
        if (e != hipSuccess) return sfail(e, "stretch launch");
        done += count;
        e = flight_mark(s, st, store ? hs.store_base + (done + thin - 1) / thin : hs.store_base);
        if (e != hipSuccess) return sfail(e, "hipEventRecord");
        return GF_OK;
    };
    if (small_bsm) {
        const char* env = gf_internal_env("GF_SAMPLER_CHAIN", 0);       // "0": grid kernels; "1": per chain (k_stretch_chain)
        const bool forced = env && (env[0] == '0' || env[0] == '1' || env[0] == '3');
        int shape = forced ? (env[0] == '0' ? 2 : 1) : (nsteps >= 8 * CHAIN_STEPS ? 0 : s->shape);
#ifdef GF_EXPERIMENTAL_FLOW
        flow = s->nwalkers <= CH_BLOCK && env && env[0] == '3';               // the dataflow kernel has one thread per walker
#endif
        if (shape != 2) {
            if (!s->d_lazy_rows) {
                // room for the proposals whose verdict only the count waits for: at least two passes' worth per chain, ~64 MB in all
                const int pass = s->nwalkers < CH_BLOCK ? s->nwalkers : CH_BLOCK;          // (k_stretch_flow: up to nwalkers per round)
                int64_t cap = ((int64_t)64 << 20) / ((int64_t)s->nchains * (int64_t)(sizeof(double) * GF_PEND_STRIDE + 8));
                if (cap > 1024) cap = 1024;
                if (cap < 2 * pass) cap = 2 * pass;
                GFS_HIP(hipMalloc((void**)&s->d_lazy_rows, sizeof(double) * GF_PEND_STRIDE * (size_t)cap * s->nchains));
                {
                    hipError_t e_ = hipMalloc((void**)&s->d_lazy_mask, sizeof(unsigned long long) * (size_t)cap * s->nchains);
                    if (e_ != hipSuccess) { (void)hipFree(s->d_lazy_rows); s->d_lazy_rows = nullptr; return sfail(e_, "hipMalloc(lazy list)"); }
                }
                s->lazy_cap = (int)cap;
                if (hipMalloc((void**)&s->d_chain_stats, sizeof(unsigned long long) * 8 * (size_t)s->nchains) == hipSuccess)
                    (void)hipMemsetAsync(s->d_chain_stats, 0, sizeof(unsigned long long) * 8 * (size_t)s->nchains, st);
                else { s->d_chain_stats = nullptr; (void)hipGetLastError(); }
            }
            ca.lazy_rows = s->d_lazy_rows; ca.lazy_mask = s->d_lazy_mask; ca.lazy_cap = s->lazy_cap;
            ca.stats = s->d_chain_stats;
            ca.commons = s->d_commons; ca.tbs = s->models ? s->d_tbs : nullptr; ca.tb = tb; ca.ptabs = s->d_ptabs;
            ca.nmodels = s->models ? s->nchains : 1; ca.nwalkers = s->nwalkers;
            ca.pos = s->d_pos; ca.lnp = s->d_lnp; ca.naccept = s->d_naccept; ca.flags = s->d_flags; ca.pend_rows = s->d_pend_rows;
            ca.chain = store ? s->d_chain : nullptr; ca.lnp_chain = store ? s->d_lnp_chain : nullptr;
            ca.nstore_cap = s->nstore_cap; ca.store_base = s->nstored; ca.seed = s->seed; ca.thin = thin; ca.store = store ? 1 : 0;
            ca.a = s->a; ca.stream_ids = s->d_stream_ids;
#ifdef GF_EXPERIMENTAL_FLOW
            if (flow && !s->d_pv) {
                const size_t nwk = (size_t)s->nchains * s->nwalkers;
                hipError_t e_ = hipMalloc((void**)&s->d_pv, sizeof(double) * FLOW_VERS * nwk * s->ndim);
                if (e_ == hipSuccess) e_ = hipMalloc((void**)&s->d_lv, sizeof(double) * FLOW_VERS * nwk);
                if (e_ == hipSuccess) e_ = hipMalloc((void**)&s->d_frows, sizeof(double) * GF_PEND_STRIDE * nwk);
                if (e_ == hipSuccess) e_ = hipMalloc((void**)&s->d_fterms, sizeof(double) * Team9::LANES * 8 * nwk);
                if (e_ != hipSuccess) return sfail(e_, "hipMalloc(dataflow sampler)");
            }
            ca.pv = s->d_pv; ca.lv = s->d_lv; ca.frows = s->d_frows; ca.fterms = s->d_fterms;
#endif
        }
        if (shape == 0 && !forced && nsteps >= 8 * CHAIN_STEPS) {
            // the probe: two blocks per chain, two blocks on the grid, the second of each timed
            double us[2] = {0.0, 0.0};
            for (int which = 0; which < 2; ++which)
                for (int rep = 0; rep < 2; ++rep) {
                    GFS_HIP(hipStreamSynchronize(st));
                    const auto t0 = std::chrono::steady_clock::now();
                    int rc = GF_OK;
                    if (which == 0) {
                        rc = chain_block(CHAIN_STEPS);
                        if (rc == GF_OK) { hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, st, s->d_state, (int)CHAIN_STEPS); if (hipGetLastError() != hipSuccess) rc = GF_ERR_HIP; }
                    } else {
                        rc = grid_block((int)CHAIN_STEPS);
                    }
                    if (rc != GF_OK) return rc;
                    GFS_HIP(hipStreamSynchronize(st));
                    if (rep == 1) us[which] = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                }
            s->probe_us[0] = us[0]; s->probe_us[1] = us[1];
            shape = s->shape = us[0] <= us[1] ? 1 : 2;
        }
        if (shape != 2) {                                                  // per chain (also: not decided, and the run too short to probe)
            while (done < nsteps) {
                const int rc = chain_block(nsteps - done < CHAIN_STEPS ? nsteps - done : CHAIN_STEPS);
                if (rc != GF_OK) return rc;
            }
            s->iteration += (uint64_t)nsteps;
            s->steps_since_reset += nsteps;
            if (store) s->nstored += (nsteps + thin - 1) / thin;
            return GF_OK;
        }
    }
    constexpr int GRAPH_STEPS = 16;
    const bool no_graph = gf_internal_env("GF_SAMPLER_NO_GRAPH", 0) != nullptr;          // diagnostics, read per run
    if (!no_graph && nsteps >= 2 * GRAPH_STEPS) {
        // launch-bound inner loop -> hipGraph: capture GRAPH_STEPS steps once, replay
        if (!s->graph || s->graph_chain != a.chain || s->graph_lnp_chain != a.lnp_chain || s->graph_cap != a.nstore_cap ||
            s->graph_has_chain != (store ? 1 : 0)) {
            if (s->graph) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; }
            hipGraph_t g = nullptr;
            const auto t_cap = std::chrono::steady_clock::now();
            struct Cap { std::chrono::steady_clock::time_point t0; ~Cap() { g_last_run_prologue[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } cap_{t_cap};
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                e = steps(GRAPH_STEPS);
                hipError_t e2 = hipStreamEndCapture(st, &g);
                if (e == hipSuccess) e = e2;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess) {                      // graphs unavailable: fall through to eager launches
                (void)hipGetLastError();
                s->graph = nullptr;
            } else {
                s->graph_chain = a.chain;
                s->graph_lnp_chain = a.lnp_chain;
                s->graph_cap = a.nstore_cap;
                s->graph_has_chain = store ? 1 : 0;
            }
        }
        while (s->graph && nsteps - done >= GRAPH_STEPS) {
            hipError_t e = flight_admit(s);
            if (e != hipSuccess) return sfail(e, "block in flight");
            const auto t_g = std::chrono::steady_clock::now();
            e = hipGraphLaunch(s->graph, st);
            if (s->sink) {
                const double d = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_g).count();
                s->sink->t[5] += d; if (d > s->sink->t[6]) s->sink->t[6] = d;
            }
            if (e != hipSuccess) return sfail(e, "hipGraphLaunch");
            done += GRAPH_STEPS;
            e = flight_mark(s, st, store ? hs.store_base + (done + thin - 1) / thin : hs.store_base);
            if (e != hipSuccess) return sfail(e, "hipEventRecord");
        }
    }
    while (done < nsteps) {
        const int count = (int)((nsteps - done < 64) ? (nsteps - done) : 64);
        hipError_t e = flight_admit(s);
        if (e != hipSuccess) return sfail(e, "block in flight");
        e = steps(count);
        if (e != hipSuccess) return sfail(e, "stretch launch");
        done += count;
        e = flight_mark(s, st, store ? hs.store_base + (done + thin - 1) / thin : hs.store_base);
        if (e != hipSuccess) return sfail(e, "hipEventRecord");
    }
    s->iteration += (uint64_t)nsteps;
    s->steps_since_reset += nsteps;
    if (store) s->nstored += (nsteps + thin - 1) / thin;
    return GF_OK;
}

int gf_sampler_sync(gf_sampler* s)
{
    if (!s) return GF_ERR_INVALID_ARG;
    return gf_model_sync(s->model);
}

// gf_sampler_run(..., store = 1) with the read-back of the chain overlapped: the run is enqueued as usual (asynchronously, an
// event behind every block of 16 steps), and while the GPU works through it this thread copies every finished block of steps
// of all chains to the host on a second stream, through the library's pinned ring.  On return the run is complete and
// chain [nchains][nstored][nwalkers][ndim] / lnprob_chain [nchains][nstored][nwalkers] (may be NULL) hold the WHOLE stored chain
// (steps stored by earlier runs included).  What sampler.chain needs after run_mcmc (golemflavor/mcmc.py:41-43), without the
// wait for the read-back at the end.  *readback_tail_s (may be NULL): seconds between the end of the run on the GPU and the end of
// the last copy -- what of the read-back was NOT hidden behind the run.
int gf_sampler_run_to_host(gf_sampler* s, int64_t nsteps, int thin, double* chain, double* lnprob_chain, double* readback_tail_s)
{
    if (!s || !chain || nsteps < 0 || thin < 1) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    GFS_HIP(hipSetDevice(device));
    void* copy_stream = nullptr;
    int rc = gf_internal_borrow_copy_stream(device, &copy_stream);
    if (rc != GF_OK) return rc;
    // One thread, one pipeline: gf_sampler_run enqueues block after block and, between two blocks, hands every block the GPU has
    // finished meanwhile to the sink (flight_admit) -- at most FLIGHT blocks are ever enqueued ahead of the GPU, and the copies run
    // while the GPU works through them.  (Round 3 enqueued the whole run first and only then started to copy.)
    HostSink sink;
    sink.chain = chain; sink.lnp = lnprob_chain; sink.total = s->nstored + (nsteps + thin - 1) / thin;
    sink.copy_stream = copy_stream; sink.device = device;
    sink.last_event = sink.t_entry = std::chrono::steady_clock::now();
    if (gf_internal_env("GF_RUN_TO_HOST_NO_PIPE", 0) == nullptr) {       // diagnostics / A-B: else every block is a read-back of its own
        rc = gf_internal_d2h_pipe_open(device, copy_stream, &sink.pipe);
        if (rc != GF_OK) { gf_internal_return_copy_stream(device, copy_stream); return rc; }
    }
    const bool marks = gf_internal_env("GF_RUN_TO_HOST_NO_MARKS", 0) == nullptr;      // diagnostics: no sink during the run = one copy after it
    s->sink = marks ? &sink : nullptr;
    rc = gf_sampler_run(s, nsteps, thin, 1);
    hipError_t e = hipSuccess;
    while (rc == GF_OK && e == hipSuccess && s->flight_done < s->flight_enq) e = flight_consume(s, true);   // the blocks still in flight
    s->sink = nullptr;
    const hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
    auto t_done = marks && s->flight_enq > 0 ? sink.last_event : std::chrono::steady_clock::now();        // when the run itself was complete on the GPU
    if (rc == GF_OK && e == hipSuccess && e2 == hipSuccess && sink.rc == GF_OK) {
        s->sink = &sink;                                            // whatever no block's event covered (one-launch runs; no marks)
        (void)sink_copy(s, s->nstored);
        s->sink = nullptr;
    }
    const int rc_pipe = gf_internal_d2h_pipe_close(sink.pipe);          // drains: every chunk is in the destination
    (void)hipStreamSynchronize((hipStream_t)copy_stream);
    gf_internal_return_copy_stream(device, copy_stream);
    if (rc == GF_OK && sink.rc == GF_OK) sink.rc = rc_pipe;
    if (readback_tail_s) *readback_tail_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_done).count();
    for (int i = 0; i < 8; ++i) g_last_run_to_host_times[i] = sink.t[i];
    if (rc != GF_OK) return rc;
    if (sink.rc != GF_OK) return sink.rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return sfail(e, "gf_sampler_run_to_host");
    return GF_OK;
}

// diagnostics (tools/, not part of the ABI): k_stretch_chain's per-chain census since the last reset, out [nchains][8] (ChainArgs::stats);
// GF_ERR_UNSUPPORTED where the sampler has not run that kernel
int gf_internal_sampler_chain_stats(gf_sampler* s, unsigned long long* out)
{
    if (!s || !out) return GF_ERR_INVALID_ARG;
    if (!s->d_chain_stats) return GF_ERR_UNSUPPORTED;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    GFS_HIP(hipMemcpyAsync(out, s->d_chain_stats, sizeof(unsigned long long) * 8 * (size_t)s->nchains, hipMemcpyDeviceToHost, (hipStream_t)stream));
    GFS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GF_OK;
}

// diagnostics (not part of the ABI): the host thread's times of the process's LAST gf_sampler_run_to_host (HostSink::t), seconds / counts
int gf_internal_run_to_host_times(double out[8])
{
    if (!out) return GF_ERR_INVALID_ARG;
    for (int i = 0; i < 8; ++i) out[i] = g_last_run_to_host_times[i];
    return GF_OK;
}
int gf_internal_run_prologue_times(double out[4])
{
    if (!out) return GF_ERR_INVALID_ARG;
    for (int i = 0; i < 4; ++i) out[i] = g_last_run_prologue[i];
    return GF_OK;
}

// diagnostics (not part of the ABI): out[0] = the launch shape of a small-ensemble BSM sampler (0 undecided, 1 per chain, 2 grid),
// out[1], out[2] = the probe's us per block of 16 steps, per chain / grid (0 when it has not run)
int gf_internal_sampler_shape(const gf_sampler* s, double out[3])
{
    if (!s || !out) return GF_ERR_INVALID_ARG;
    out[0] = (double)s->shape; out[1] = s->probe_us[0]; out[2] = s->probe_us[1];
    return GF_OK;
}

int64_t gf_sampler_nstored(const gf_sampler* s) { return s ? s->nstored : -1; }
int64_t gf_sampler_iterations(const gf_sampler* s) { return s ? s->steps_since_reset : -1; }

// pos [nchains][nwalkers][ndim], lnprob [nchains][nwalkers] (either may be NULL)
int gf_sampler_get_state(gf_sampler* s, double* pos, double* lnprob)
{
    if (!s) return GF_ERR_INVALID_ARG;
    int rc = gf_model_sync(s->model);
    if (rc != GF_OK) return rc;
    const size_t nw = (size_t)s->nchains * s->nwalkers;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (pos) GFS_HIP(hipMemcpyAsync(pos, s->d_pos, sizeof(double) * nw * s->ndim, hipMemcpyDeviceToHost, st));
    if (lnprob) GFS_HIP(hipMemcpyAsync(lnprob, s->d_lnp, sizeof(double) * nw, hipMemcpyDeviceToHost, st));
    GFS_HIP(hipStreamSynchronize(st));
    return GF_OK;
}

// chain [nchains][nstored][nwalkers][ndim], lnprob_chain [nchains][nstored][nwalkers],
// naccepted [nchains][nwalkers], nonunitary[1]; any may be NULL.
int gf_sampler_get_chain(gf_sampler* s, double* chain, double* lnprob_chain, uint32_t* naccepted, uint32_t* nonunitary)
{
    if (!s) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;                 // in order behind the sampler's launches
    const size_t per = (size_t)s->nwalkers;
    if (s->nstored > 0) {
        // one strided copy per array: row = chain (stored prefix), device pitch = capacity stride
        const size_t row = sizeof(double) * per * s->ndim, lrow = sizeof(double) * per;
        // (gf_internal_d2h_2d: a plain strided copy below 16 MB, the pinned ring + host copy pool from there on)
        if (chain) {
            const int rc = gf_internal_d2h_2d(device, stream, chain, row * s->nstored, s->d_chain, row * s->nstore_cap, row * s->nstored,
                                              (size_t)s->nchains);
            if (rc != GF_OK) return rc;
        }
        if (lnprob_chain) {
            const int rc = gf_internal_d2h_2d(device, stream, lnprob_chain, lrow * s->nstored, s->d_lnp_chain, lrow * s->nstore_cap,
                                              lrow * s->nstored, (size_t)s->nchains);
            if (rc != GF_OK) return rc;
        }
    }
    if (naccepted)
        GFS_HIP(hipMemcpyAsync(naccepted, s->d_naccept, sizeof(uint32_t) * (size_t)s->nchains * per, hipMemcpyDeviceToHost, st));
    if (nonunitary) GFS_HIP(hipMemcpyAsync(nonunitary, s->d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GFS_HIP(hipStreamSynchronize(st));
    return GF_OK;
}

// The stored chain packed into caller-owned DEVICE buffers (the capacity padding of the sampler's own buffer removed):
// d_chain [nchains][nstored][nwalkers][ndim], d_lnprob_chain [nchains][nstored][nwalkers]; either may be NULL.
// Synchronous on return.  What a multi-GPU gather sends (gf_comm_allgather takes device pointers).
int gf_sampler_get_chain_device(gf_sampler* s, double* d_chain, double* d_lnprob_chain)
{
    if (!s) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    GFS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (s->nstored > 0) {
        const size_t row = sizeof(double) * (size_t)s->nwalkers * s->ndim, lrow = sizeof(double) * (size_t)s->nwalkers;
        if (d_chain)
            GFS_HIP(hipMemcpy2DAsync(d_chain, row * s->nstored, s->d_chain, row * s->nstore_cap, row * s->nstored, s->nchains,
                                     hipMemcpyDeviceToDevice, st));
        if (d_lnprob_chain)
            GFS_HIP(hipMemcpy2DAsync(d_lnprob_chain, lrow * s->nstored, s->d_lnp_chain, lrow * s->nstore_cap, lrow * s->nstored,
                                     s->nchains, hipMemcpyDeviceToDevice, st));
    }
    GFS_HIP(hipStreamSynchronize(st));
    return GF_OK;
}

// mean [nchains][nstored][ndim]: the ensemble-averaged series whose integrated autocorrelation time the
// reference prints (golemflavor/mcmc.py:45-51 sampler.acor); reduced on the device, only the means cross PCIe.
int gf_sampler_walker_mean(gf_sampler* s, double* mean)
{
    if (!s || !mean) return GF_ERR_INVALID_ARG;
    const GfCommon* c; const GfBsm* tb; const double* ptab; void* stream; int device;
    if (gf_model_internal(s->model, &c, &tb, &ptab, &stream, &device) != GF_OK) return GF_ERR_INVALID_ARG;
    GFS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (s->nstored == 0) { GFS_HIP(hipStreamSynchronize(st)); return GF_OK; }
    const size_t bytes = sizeof(double) * (size_t)s->nchains * s->nstored * s->ndim;
    double* d_mean = nullptr;
    GFS_HIP(hipMalloc((void**)&d_mean, bytes));
    hipError_t e = hipSuccess;
    for (int ch0 = 0; ch0 < s->nchains && e == hipSuccess; ch0 += 65535) {          // gridDim.y <= 65535
        const int nch = s->nchains - ch0 < 65535 ? s->nchains - ch0 : 65535;
        hipLaunchKernelGGL(k_walker_mean, dim3((unsigned)s->nstored, (unsigned)nch), dim3(GF_BLOCK), 0, st,
                           s->d_chain + (size_t)ch0 * s->nstore_cap * s->nwalkers * s->ndim, s->nstore_cap, s->nstored, s->nwalkers,
                           s->ndim, d_mean + (size_t)ch0 * s->nstored * s->ndim);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mean, d_mean, bytes, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_mean);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return sfail(e, "gf_sampler_walker_mean");
    return GF_OK;
}

// Chain post-processing on the device (scripts/mc_unitary.py:189-193, mc_texture.py:216-221, and the
// histogram of golemflavor/plot.py:365-370): measured composition of every stored sample, optionally
// reduced to the [nbins]^3 flavor histogram so that only the counts cross PCIe.
//   fr      [nchains][nstored][nwalkers][3]  or NULL
//   status  [nchains][nstored][nwalkers]     or NULL
//   counts  [nchains][nbins][nbins][nbins]   or NULL (nbins ignored then)
int gf_sampler_postprocess(gf_sampler* s, double* fr, int32_t* status, int nbins, uint64_t* counts)
{
    return gf_sampler_postprocess_with(s, nullptr, fr, status, nbins, counts);
}

// Same, but chain ch is propagated with models[ch] instead of the posterior it was sampled from
// (scripts/mc_texture.py: the chain samples the priors, mc_texture.py:148-170, and every sample is then pushed
// through flux_averaged_BSMu at the grid point's scale and source, mc_texture.py:216-221).  models == NULL:
// the sampling models.  Each model must have the sampler's ndim and device.
// Same with DEVICE destinations: d_fr [nchains][nstored][nwalkers][3], d_status [nchains][nstored][nwalkers] (NULL = skip);
// synchronous on return.
int gf_sampler_postprocess_device(gf_sampler* s, gf_model* const* models, double* d_fr, int32_t* d_status)
{
    if (!s || !d_fr) return GF_ERR_INVALID_ARG;
    const GfCommon* c0; const GfBsm* tb; const double* ptab; void* stream; int device0;
    if (gf_model_internal(s->model, &c0, &tb, &ptab, &stream, &device0) != GF_OK) return GF_ERR_INVALID_ARG;
    for (int ch = 0; ch < s->nchains; ++ch) {
        gf_model* mc = models ? models[ch] : s->models ? s->models[ch] : s->model;
        const GfCommon* c; int device, cus, nbins;
        if (gf_model_constants(mc, &c, &tb, &ptab, &device, &cus, &nbins) != GF_OK || c->ndim != s->ndim || device != device0)
            return GF_ERR_INVALID_ARG;
    }
    GFS_HIP(hipSetDevice(device0));
    hipStream_t st = (hipStream_t)stream;
    const int64_t per_chain = s->nstored * s->nwalkers;
    int rc = GF_OK;
    // the chains are enqueued faster than they run, so the arbitration grid of each would follow what some EARLIER chain found,
    // and the chains of a scan differ (its high-scale grid points sit in the failing region, the others have empty queues):
    // full grids throughout, ~30 us per chain (measured: the hint left 57 of 64 chains of the C4 scan on a sixth of the
    // GPU, 114 ms of arbitration instead of ~20)
    gf_internal_full_arbitration_grids(device0, stream, 1);
    for (int ch = 0; ch < s->nchains && rc == GF_OK && per_chain > 0; ++ch) {
        gf_model* mc = models ? models[ch] : s->models ? s->models[ch] : s->model;
        const double* d_theta = s->d_chain + (size_t)ch * s->nstore_cap * s->nwalkers * s->ndim;
        rc = gf_model_propagate_on(mc, stream, d_theta, GF_LAYOUT_AOS, per_chain, d_fr + (size_t)ch * per_chain * 3,
                                   d_status ? d_status + (size_t)ch * per_chain : nullptr);
    }
    gf_internal_full_arbitration_grids(device0, stream, 0);
    GFS_HIP(hipStreamSynchronize(st));
    if (rc == GF_OK && d_status) rc = gf_internal_check_overflow(device0, stream);
    return rc;
}

// The scan's output rows on the device: d_rows [nchains][nstored][nwalkers][3 + ndim] = composition (NaN where the
// reference would have raised) then the sample, each chain propagated with models[ch] (NULL: the sampling models).
// Synchronous on return.  scripts/mc_texture.py:216-223.
int gf_sampler_postprocess_rows_device(gf_sampler* s, gf_model* const* models, double* d_rows)
{
    if (!s || !d_rows) return GF_ERR_INVALID_ARG;
    const GfCommon* c0; const GfBsm* tb; const double* ptab; void* stream; int device0;
    if (gf_model_internal(s->model, &c0, &tb, &ptab, &stream, &device0) != GF_OK) return GF_ERR_INVALID_ARG;
    GFS_HIP(hipSetDevice(device0));
    hipStream_t st = (hipStream_t)stream;
    const int64_t per_chain = s->nstored * s->nwalkers;
    if (per_chain == 0) { GFS_HIP(hipStreamSynchronize(st)); return GF_OK; }
    double* d_fr = nullptr;
    int32_t* d_st = nullptr;
    GFS_HIP(hipMalloc((void**)&d_fr, sizeof(double) * 3 * per_chain * s->nchains));
    {
        hipError_t e_ = hipMalloc((void**)&d_st, sizeof(int32_t) * per_chain * s->nchains);
        if (e_ != hipSuccess) { (void)hipFree(d_fr); return sfail(e_, "hipMalloc(status)"); }
    }
    int rc = gf_sampler_postprocess_device(s, models, d_fr, d_st);
    hipError_t e = hipSuccess;
    for (int ch = 0; ch < s->nchains && rc == GF_OK && e == hipSuccess; ++ch) {
        const double* d_theta = s->d_chain + (size_t)ch * s->nstore_cap * s->nwalkers * s->ndim;
        e = gf_launch_join_rows(d_fr + (size_t)ch * per_chain * 3, d_st + (size_t)ch * per_chain, d_theta, s->ndim, per_chain,
                                d_rows + (size_t)ch * per_chain * (3 + s->ndim), s->cus, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_fr);
    (void)hipFree(d_st);
    if (rc != GF_OK) return rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return sfail(e, "gf_sampler_postprocess_rows_device");
    return GF_OK;
}

// The scan's rows straight to the host.  The chains are post-processed in turn on the sampler's stream (everything is enqueued
// at once); an event marks the end of every group of chains, and one pinned-ring copy on a SECOND stream follows the events
// chunk by chunk -- so the read-back of finished groups overlaps the evaluation (and the x87 arbitration, which dominates a
// texture scan's post-processing) of the later ones.
int gf_sampler_postprocess_rows(gf_sampler* s, gf_model* const* models, double* rows)
{
    if (!s || !rows) return GF_ERR_INVALID_ARG;
    const GfCommon* c0; const GfBsm* tb; const double* ptab; void* stream; int device0;
    if (gf_model_internal(s->model, &c0, &tb, &ptab, &stream, &device0) != GF_OK) return GF_ERR_INVALID_ARG;
    for (int ch = 0; ch < s->nchains; ++ch) {
        gf_model* mc = models ? models[ch] : s->models ? s->models[ch] : s->model;
        const GfCommon* c; int device, cus, nbins;
        if (gf_model_constants(mc, &c, &tb, &ptab, &device, &cus, &nbins) != GF_OK || c->ndim != s->ndim || device != device0)
            return GF_ERR_INVALID_ARG;
    }
    GFS_HIP(hipSetDevice(device0));
    hipStream_t st = (hipStream_t)stream;
    const int64_t per_chain = s->nstored * s->nwalkers;
    if (per_chain == 0) { GFS_HIP(hipStreamSynchronize(st)); return GF_OK; }
    const size_t width = 3 + (size_t)s->ndim, chain_bytes = sizeof(double) * width * (size_t)per_chain;
    constexpr int MAX_GROUPS = 16;
    const int per_group = (s->nchains + MAX_GROUPS - 1) / MAX_GROUPS;
    const int ngroups = (s->nchains + per_group - 1) / per_group;
    double *d_fr = nullptr, *d_rows = nullptr;
    int32_t* d_st = nullptr;
    void* copy_stream = nullptr;
    hipEvent_t ev[MAX_GROUPS] = {};
    int rc = GF_OK;
    hipError_t e = hipMalloc((void**)&d_fr, sizeof(double) * 3 * per_chain * s->nchains);
    if (e == hipSuccess) e = hipMalloc((void**)&d_st, sizeof(int32_t) * per_chain * s->nchains);
    if (e == hipSuccess) e = hipMalloc((void**)&d_rows, chain_bytes * s->nchains);
    for (int g = 0; g < ngroups && e == hipSuccess; ++g) e = hipEventCreateWithFlags(&ev[g], hipEventDisableTiming);
    if (e == hipSuccess) rc = gf_internal_borrow_copy_stream(device0, &copy_stream);
    if (e == hipSuccess && rc == GF_OK) {
        gf_internal_full_arbitration_grids(device0, stream, 1);      // see gf_sampler_postprocess_device
        for (int ch = 0; ch < s->nchains && rc == GF_OK && e == hipSuccess; ++ch) {
            gf_model* mc = models ? models[ch] : s->models ? s->models[ch] : s->model;
            const double* d_theta = s->d_chain + (size_t)ch * s->nstore_cap * s->nwalkers * s->ndim;
            rc = gf_model_propagate_on(mc, stream, d_theta, GF_LAYOUT_AOS, per_chain, d_fr + (size_t)ch * per_chain * 3,
                                       d_st + (size_t)ch * per_chain);
            if (rc == GF_OK)
                e = gf_launch_join_rows(d_fr + (size_t)ch * per_chain * 3, d_st + (size_t)ch * per_chain, d_theta, s->ndim, per_chain,
                                        d_rows + (size_t)ch * per_chain * width, s->cus, st);
            if (rc == GF_OK && e == hipSuccess && ((ch + 1) % per_group == 0 || ch + 1 == s->nchains))
                e = hipEventRecord(ev[ch / per_group], st);
        }
        gf_internal_full_arbitration_grids(device0, stream, 0);
        // the rows cross PCIe on the copy stream through the library's pinned ring (gf_internal_d2h_gated: the DMA fills a slot
        // while host threads empty the previous ones into `rows`, mapping its pages as they go) -- ONE pipeline over the whole
        // block, each 16 MB chunk issued as soon as the group of chains it ends in has been post-processed on the sampler's stream
        if (rc == GF_OK && e == hipSuccess) {
            struct Gate { hipEvent_t* ev; size_t group_bytes; int ngroups, passed; } gt = {ev, chain_bytes * (size_t)per_group, ngroups, 0};
            auto gate = [](void* ctx, size_t upto) -> int {
                Gate* g = static_cast<Gate*>(ctx);
                int need = (int)((upto + g->group_bytes - 1) / g->group_bytes);
                if (need > g->ngroups) need = g->ngroups;
                for (; g->passed < need; ++g->passed)
                    if (hipEventSynchronize(g->ev[g->passed]) != hipSuccess) return 1;
                return 0;
            };
            rc = gf_internal_d2h_gated(device0, copy_stream, rows, d_rows, chain_bytes * (size_t)s->nchains, gate, &gt);
        }
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    if (copy_stream) { (void)hipStreamSynchronize((hipStream_t)copy_stream); gf_internal_return_copy_stream(device0, copy_stream); }
    for (int g = 0; g < ngroups; ++g) if (ev[g]) (void)hipEventDestroy(ev[g]);
    if (d_fr) (void)hipFree(d_fr);
    if (d_st) (void)hipFree(d_st);
    if (d_rows) (void)hipFree(d_rows);
    if (rc != GF_OK) return rc;
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return sfail(e, "gf_sampler_postprocess_rows");
    return gf_internal_check_overflow(device0, stream);
}

int gf_sampler_postprocess_with(gf_sampler* s, gf_model* const* models, double* fr, int32_t* status, int nbins,
                                uint64_t* counts)
{
    if (!s || (counts && (nbins < 1 || nbins > 1024))) return GF_ERR_INVALID_ARG;
    const GfCommon* c0; const GfBsm* tb; const double* ptab; void* stream; int device0;
    if (gf_model_internal(s->model, &c0, &tb, &ptab, &stream, &device0) != GF_OK) return GF_ERR_INVALID_ARG;
    int cus = 256;
    for (int ch = 0; ch < s->nchains; ++ch) {
        gf_model* mc = models ? models[ch] : s->models ? s->models[ch] : s->model;
        const GfCommon* c; int device, nbins;
        if (gf_model_constants(mc, &c, &tb, &ptab, &device, &cus, &nbins) != GF_OK || c->ndim != s->ndim || device != device0)
            return GF_ERR_INVALID_ARG;
    }
    GFS_HIP(hipSetDevice(device0));
    hipStream_t st = (hipStream_t)stream;
    GFS_HIP(hipStreamSynchronize(st));
    if (s->nstored == 0) return GF_OK;
    const int64_t per_chain = s->nstored * s->nwalkers;
    const size_t nbin3 = counts ? (size_t)nbins * nbins * nbins : 0;
    double* d_fr = nullptr;
    int32_t* d_st = nullptr;
    uint64_t* d_c = nullptr;
    hipError_t e = hipMalloc((void**)&d_fr, sizeof(double) * 3 * per_chain);
    if (e == hipSuccess && status) e = hipMalloc((void**)&d_st, sizeof(int32_t) * per_chain);
    if (e == hipSuccess && counts) e = hipMalloc((void**)&d_c, sizeof(uint64_t) * nbin3);
    int rc = GF_OK;
    // everything in order on the sampler's stream (the one the chain was written on): propagate, histogram,
    // copies back; the scratch buffers are reused chain after chain, one sync at the end
    for (int ch = 0; ch < s->nchains && e == hipSuccess && rc == GF_OK; ++ch) {
        gf_model* mc = models ? models[ch] : s->models ? s->models[ch] : s->model;
        const double* d_theta = s->d_chain + (size_t)ch * s->nstore_cap * s->nwalkers * s->ndim;
        rc = gf_model_propagate_on(mc, stream, d_theta, GF_LAYOUT_AOS, per_chain, d_fr, d_st);
        if (rc != GF_OK) break;
        if (counts) {
            e = hipMemsetAsync(d_c, 0, sizeof(uint64_t) * nbin3, st);
            if (e == hipSuccess) e = gf_launch_flavor_hist(d_fr, per_chain, nbins, (unsigned long long*)d_c, cus, st);
        }
        if (e == hipSuccess && fr)
            e = hipMemcpyAsync(fr + (size_t)ch * per_chain * 3, d_fr, sizeof(double) * 3 * per_chain, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && status)
            e = hipMemcpyAsync(status + (size_t)ch * per_chain, d_st, sizeof(int32_t) * per_chain, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && counts)
            e = hipMemcpyAsync(counts + (size_t)ch * nbin3, d_c, sizeof(uint64_t) * nbin3, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    if (e == hipSuccess) e = e2;
    if (d_fr) (void)hipFree(d_fr);
    if (d_st) (void)hipFree(d_st);
    if (d_c) (void)hipFree(d_c);
    if (rc != GF_OK) return rc;
    if (e != hipSuccess) return sfail(e, "gf_sampler_postprocess");
    return status ? gf_internal_check_overflow(device0, stream) : GF_OK;
}

}  // extern "C"
