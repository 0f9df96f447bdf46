// gf_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X): the per-walker physics of
// GolemFlavor's ensemble log-posterior.  fp64 throughout; no MFMA (3x3 matrices, HBM/VALU bound).
//
// Data movement.  theta arrives as emcee lays it out, AoS [n][ndim] fp64.  A wave owns 64
// consecutive walkers = one contiguous 512*ndim-byte span; it pulls that span with 16-byte
// lane-contiguous loads (1 KiB per wave-instruction, every byte of every 128-B line used once) into a
// wave-private LDS tile, then each lane reads back its own row.  Named parameters (s_12_2, dcp, ...)
// are then LDS reads at a *uniform* column index, so the sampled/fixed split costs no register
// indexing.  Waves never synchronise with each other (no s_barrier): the only ordering needed is
// within one wave, and LDS executes a wave's instructions in order.
//
// Reference formulas are cited per function (file:line under the reference tree); the algebraic forms
// are those of SURVEY.md Appendix A.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "gf_consts.h"

extern "C" const char* gf_internal_env(const char* name, int affects_results);   // gf_capi.hip: getenv with a record
#include "gf_launch.h"

#include "gf_device.hpp"

namespace {
using namespace gfdev;

// ---------------------------------------------------------------------------------------------
// lnprob for MODE_PRIOR_ONLY and MODE_SM_GAUSS.  One lane per walker, grid-stride over 64-walker
// wave tiles.
//
// Per-column prior constants {lo, hi, loc, 1/sigma} come from a 512-B LDS table (ds_read_b128
// broadcasts) rather than from kernel-argument SGPRs: with them in SGPRs the kernel needed > 102 scalar
// registers and spilled to VGPR lanes (52 v_readlane per tile in the first version, profiles/r01/a_*).
// AoS tiles are software-pipelined: the 16-B loads of the wave's next tile are issued before the
// current tile is evaluated, so a wave's HBM latency hides under its own fp64 work.

// Hot kernel: AoS theta, compile-time row length, FULL 64-walker tiles only (the launcher hands the
// ragged remainder to k_lnprob_sm_gen).  Keeping the generic staging out of this loop keeps its
// induction variables and bounds checks out of the register budget.
// theta is read once and lnprob written once per launch: nontemporal accesses keep the stream from
// displacing useful lines (measured +4..8 % on the full kernel, +15 % on a bare stream of this traffic mix).
// Compile-time A/B switches (tools/build_variants.sh; results in profiles/r01/ab_*): GF_NO_NT,
// GF_PREFETCH_DEPTH, GF_SM_WAVES_PER_EU, GF_BLOCKS_PER_CU, and the diagnostic builds GF_NO_LOADS (compute
// only) / GF_STRIP_COMPUTE (memory only).  The defaults below are the measured best.
#ifndef GF_NO_NT
#define GF_LOAD_THETA(p) __builtin_nontemporal_load(p)
#define GF_STORE_OUT(p, v) __builtin_nontemporal_store(v, p)
#else
#define GF_LOAD_THETA(p) (*(p))
#define GF_STORE_OUT(p, v) (*(p) = (v))
#endif
#ifndef GF_PREFETCH_DEPTH
#define GF_PREFETCH_DEPTH 1
#endif
#ifndef GF_SM_WAVES_PER_EU
#define GF_SM_WAVES_PER_EU 4
#endif
// Waves per SIMD the kernels are compiled for: four (128 VGPRs) wherever the instance fits -- the bench kernel uses 48 --; the
// instances that spill at four get the registers instead.  Measured (tools/bench_sm12.py, 16.8 M walkers, profiles/r03/
// ab_wide_sm_instances.txt): the 12-column flavor likelihood from AoS rows 1.38 ms at four waves (144-272 B of scratch per
// lane), 0.469 ms at three (32-112 B), 0.302 ms at two (none) = 0.16 / 0.47 / 0.72 of HBM peak; 12 prior columns from an SoA
// buffer 0.782 / 0.375 / 0.311 ms.  So: two waves for 12 columns with the likelihood or from SoA, three for the generic-column
// (SAMPLED = 0) likelihood instances from 6 / 7 columns on (32-96 B at four).
#ifndef GF_SM_WIDE_WAVES_PER_EU
#define GF_SM_WIDE_WAVES_PER_EU 2
#endif
template <int NDIM, int MODE, int SAMPLED, bool SOA>
constexpr int sm_waves_per_eu()
{
    constexpr int most = GF_SM_WAVES_PER_EU;
    if (NDIM >= 12 && (MODE == MODE_SM_GAUSS || SOA)) return most < GF_SM_WIDE_WAVES_PER_EU ? most : GF_SM_WIDE_WAVES_PER_EU;
    if (SAMPLED == 0 && MODE == MODE_SM_GAUSS && NDIM >= 6 && (SOA || NDIM >= 7)) return most < 3 ? most : 3;
    return most;
}
template <int NDIM, int MODE, int SAMPLED, bool WANT_FR>
__global__ __launch_bounds__(GF_BLOCK, (sm_waves_per_eu<NDIM, MODE, SAMPLED, false>())) void k_lnprob_sm_fast(const GfCommon c, const double* __restrict__ ptab,
                                                              const double* __restrict__ theta, int64_t nfull,
                                                              double* __restrict__ lnprob, double* __restrict__ fr_out,
                                                              int32_t* __restrict__ status)
{
    static_assert(NDIM > 0, "fast path needs a compile-time row length");
    constexpr int NV = GF_WAVE * NDIM / 2;                     // 16-B vectors in a tile
    constexpr int VPL = (NV + GF_WAVE - 1) / GF_WAVE;          // ... per lane
    constexpr bool EVEN = (NV % GF_WAVE) == 0;
    typedef double d2_t __attribute__((ext_vector_type(2)));   // native vector: stays in registers
    __shared__ __attribute__((aligned(16))) double tiles[GF_WAVES_PER_BLOCK][GF_WAVE * NDIM];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4];
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / GF_WAVE);        // wave-uniform -> SALU
    double* tile = tiles[wave];
    // Workgroups are dealt round-robin to the 8 XCDs of an MI355X; each XCD (own L2, own path to the fabric)
    // streams one contiguous eighth of the batch instead of every eighth tile group of all of it.  Measured within
    // the +-2 % run-to-run noise of the plain grid-stride order (GF_NO_XCD_MAP) on this pure stream -- there is no
    // reuse for an L2 to keep -- two interleaved A/B runs in profiles/r01/ab_xcd_contiguous.txt.
#ifndef GF_NO_XCD_MAP
    const int nx = (gridDim.x % 8 == 0) ? 8 : 1;
#else
    const int nx = 1;
#endif
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx;
    const int64_t tend = nfull * (xcd + 1) / nx;
    const int stride = (gridDim.x / nx) * GF_WAVES_PER_BLOCK;
    int64_t t = nfull * xcd / nx + (int64_t)lb * GF_WAVES_PER_BLOCK + wave;
    if (t >= tend) return;

    // PD tiles of this wave are in flight in registers (software pipeline of depth PD); a tile index past
    // the end is clamped to the wave's last valid tile (a redundant but harmless re-read at the tail).
    constexpr int PD = GF_PREFETCH_DEPTH;
    d2_t pre[PD][VPL];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const int64_t tp = (t + (int64_t)p * stride < tend) ? t + (int64_t)p * stride : t;
        const d2_t* src = reinterpret_cast<const d2_t*>(theta + tp * (GF_WAVE * NDIM));
#pragma unroll
        for (int j = 0; j < VPL; ++j)
            if (EVEN || j * GF_WAVE + lane < NV) pre[p][j] = GF_LOAD_THETA(src + j * GF_WAVE + lane);
    }
    // The lnprob store of tile t is issued at the top of iteration t+1, after the wait for tile t+1's
    // loads: with loads and a store pending together the compiler waits vmcnt(0) (mixed VMEM event types),
    // so a store issued at the end of an iteration would put its write-acknowledge on the next wait.
    // (Measured neutral on MI355X at 4 waves/SIMD -- other waves cover it -- kept because it is free.)
    double val_prev = 0.0;
    int64_t i_prev = -1;
    for (; t < tend; t += stride) {
        // oldest tile in flight: registers -> LDS; shift the pipeline; fetch the tile PD strides ahead
#pragma unroll
        for (int j = 0; j < VPL; ++j)
            if (EVEN || j * GF_WAVE + lane < NV) reinterpret_cast<d2_t*>(tile)[j * GF_WAVE + lane] = pre[0][j];
        if (i_prev >= 0) GF_STORE_OUT(lnprob + i_prev, val_prev);
#pragma unroll
        for (int p = 0; p + 1 < PD; ++p)
#pragma unroll
            for (int j = 0; j < VPL; ++j) pre[p][j] = pre[p + 1][j];
#ifndef GF_NO_LOADS
        const int64_t ta = t + (int64_t)PD * stride;
        const int64_t tn = (ta < tend) ? ta : t;
        const d2_t* src = reinterpret_cast<const d2_t*>(theta + tn * (GF_WAVE * NDIM));
#pragma unroll
        for (int j = 0; j < VPL; ++j)
            if (EVEN || j * GF_WAVE + lane < NV) pre[PD - 1][j] = GF_LOAD_THETA(src + j * GF_WAVE + lane);
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int64_t i = t * GF_WAVE + lane;
        double val, fr[3];
        int st;
#ifdef GF_STRIP_COMPUTE
        val = tile[lane * NDIM] + tile[lane * NDIM + NDIM - 1]; st = 0; fr[0] = fr[1] = fr[2] = val;
#else
        eval_walker<NDIM, MODE, SAMPLED, WANT_FR>(c, ctab, tile + lane * NDIM, NDIM, val, fr, st);
#endif
        val_prev = val;
        i_prev = i;
        if (WANT_FR) { fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2]; }
        if (status) status[i] = st;
        // the tile is rewritten by the same wave next iteration; keep its reads ahead of those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (i_prev >= 0) GF_STORE_OUT(lnprob + i_prev, val_prev);
}

// SoA theta ([ndim][n], one contiguous column per parameter): every lane reads its own walker's NDIM values with
// lane-contiguous 8-B loads (512 B per wave-instruction, each 128-B line used once) straight into registers -- no LDS
// tile, no transposition.  Same software pipeline as the AoS kernel: the next tile's loads are in flight while the
// current one is evaluated.  Full 64-walker tiles; the launcher hands the remainder to k_lnprob_sm_gen.  SAMPLED is 2
// (canonical columns: the row registers are used by position) or 0 for MODE_PRIOR_ONLY; posteriors that read named
// columns through a runtime index stay on the generic kernel (a register array cannot be indexed dynamically).
template <int NDIM, int MODE, int SAMPLED, bool WANT_FR>
__global__ __launch_bounds__(GF_BLOCK, (sm_waves_per_eu<NDIM, MODE, SAMPLED, true>())) void k_lnprob_sm_soa(const GfCommon c, const double* __restrict__ ptab,
                                                             const double* __restrict__ theta, int64_t n, int64_t nfull,
                                                             double* __restrict__ lnprob, double* __restrict__ fr_out,
                                                             int32_t* __restrict__ status)
{
    static_assert(NDIM > 0, "compile-time row length");
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4];
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / GF_WAVE);
    const int64_t stride = (int64_t)gridDim.x * GF_WAVES_PER_BLOCK;
    int64_t t = (int64_t)blockIdx.x * GF_WAVES_PER_BLOCK + wave;
    if (t >= nfull) return;
    double pre[NDIM];
#pragma unroll
    for (int d = 0; d < NDIM; ++d) pre[d] = GF_LOAD_THETA(theta + (int64_t)d * n + t * GF_WAVE + lane);
    for (; t < nfull; t += stride) {
        double row[NDIM];
#pragma unroll
        for (int d = 0; d < NDIM; ++d) row[d] = pre[d];
        const int64_t tn = (t + stride < nfull) ? t + stride : t;       // past the end: a harmless re-read of this tile
#pragma unroll
        for (int d = 0; d < NDIM; ++d) pre[d] = GF_LOAD_THETA(theta + (int64_t)d * n + tn * GF_WAVE + lane);
        const int64_t i = t * GF_WAVE + lane;
        double val, fr[3];
        int st;
        eval_walker<NDIM, MODE, SAMPLED, WANT_FR>(c, ctab, row, NDIM, val, fr, st);
        GF_STORE_OUT(lnprob + i, val);
        if (WANT_FR) { fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2]; }
        if (status) status[i] = st;
    }
}

#if defined(GF_ASM_PIPE) || defined(GF_EXPERIMENTAL_RING)
#include "../../tools/experiments/gf_sm_experiments.hpp"   // measured alternatives, not built by default
#endif


// Generic kernel: any layout, runtime row length, ragged tiles; walkers [first, n) of the batch.
template <int NDIM, int MODE>
__global__ __launch_bounds__(GF_BLOCK) void k_lnprob_sm_gen(const GfCommon c, const double* __restrict__ ptab,
                                                             const double* __restrict__ theta, int layout, int64_t first,
                                                             int64_t n, double* __restrict__ lnprob,
                                                             double* __restrict__ fr_out, int32_t* __restrict__ status)
{
    constexpr int ND = NDIM ? NDIM : GF_MAX_DIM;
    __shared__ __attribute__((aligned(16))) double tiles[GF_WAVES_PER_BLOCK][GF_WAVE * ND];
    __shared__ __attribute__((aligned(16))) double ctab[GF_MAX_DIM * 4];
    if (threadIdx.x < GF_MAX_DIM * 4) ctab[threadIdx.x] = ptab[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = threadIdx.x / GF_WAVE;
    const int ndim = NDIM ? NDIM : c.ndim;
    double* tile = tiles[wave];
    const int64_t ntiles = (n - first + GF_WAVE - 1) / GF_WAVE;
    const int64_t stride = (int64_t)gridDim.x * GF_WAVES_PER_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * GF_WAVES_PER_BLOCK + wave; t < ntiles; t += stride) {
        const int64_t w0 = first + t * GF_WAVE;
        stage_theta<NDIM>(theta, layout, n, w0, ndim, tile, lane);
        const int64_t i = w0 + lane;
        if (i < n) {
            double val, fr[3];
            int st;
            eval_walker<NDIM, MODE, 0, true>(c, ctab, tile + lane * ndim, ndim, val, fr, st);
            lnprob[i] = val;
            if (fr_out) { fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2]; }
            if (status) status[i] = st;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Chain post-processing: composition only (scripts/mc_unitary.py:189-193).
template <int NDIM>
__global__ __launch_bounds__(GF_BLOCK) void k_propagate_sm(const GfCommon c, const double* __restrict__ theta,
                                                            int layout, int64_t n, double* __restrict__ fr_out,
                                                            int32_t* __restrict__ status)
{
    __shared__ __attribute__((aligned(16))) double tiles[GF_WAVES_PER_BLOCK][GF_WAVE * (NDIM ? NDIM : GF_MAX_DIM)];
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int wave = threadIdx.x / GF_WAVE;
    const int ndim = NDIM ? NDIM : c.ndim;
    double* tile = tiles[wave];
    const int64_t ntiles = (n + GF_WAVE - 1) / GF_WAVE;
    const int64_t stride = (int64_t)gridDim.x * GF_WAVES_PER_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * GF_WAVES_PER_BLOCK + wave; t < ntiles; t += stride) {
        const int64_t w0 = t * GF_WAVE;
        stage_theta<NDIM>(theta, layout, n, w0, ndim, tile, lane);
        const int64_t i = w0 + lane;
        if (i < n) {
            double fr[3];
            sm_composition(c, tile + lane * ndim, fr);
            fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2];
            if (status) status[i] = (fr[0] != fr[0] || fr[1] != fr[1] || fr[2] != fr[2]) ? ST_NAN : ST_OK;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so draw i is a pure function of (seed, i).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input exclusive or in ONE instruction (v_bitop3_b32, truth table 0x96; gfx950): the compiler emits two v_xor_b32
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(m1 >> 32), c1, k0, 0x96);
        const uint32_t n1 = (uint32_t)m1;
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(m0 >> 32), c3, k1, 0x96);
        const uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo)
{
    // 53 random bits -> [0,1): (hi >> 5) * 2^26 + (lo >> 6), scaled by 2^-53
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// Composition of one Haar draw: u_to_fr(source, angles_to_u(angles)) (scripts/mc_unitary.py:189-193) with only the 2 x 3
// block of |U|^2 that the unit row and column sums leave independent -- w = P^T s = s2 + ds0 P_0. + ds1 P_1. (ds = s - s2),
// f_b = w2 + P_b0 (w0 - w2) + P_b1 (w1 - w2) for b = e, mu, f_tau = sum(w) - f_e - f_mu: ~60 fp64 instructions where
// pmns_abs2 + propagate take ~85 (the kernel is VALU-bound: the arithmetic around the two Philox blocks is a third of it).
__device__ __forceinline__ void haar_composition(double s12_2, double c13_4, double s23_2, double dcp, double s2, double ds0, double ds1,
                                                 double fr[3])
{
    const double c13_2 = fast_sqrt(c13_4);
    const double s13_2 = 1.0 - c13_2;
    const double c12_2 = 1.0 - s12_2;
    const double c23_2 = 1.0 - s23_2;
    const double a = s12_2 * c23_2, b = c12_2 * s23_2;
    const double e = c12_2 * c23_2, f = s12_2 * s23_2;
    const double j2 = 2.0 * fast_sqrt((a * b) * s13_2) * fast_cos_phase(dcp);
    const double p00 = c12_2 * c13_2, p01 = s12_2 * c13_2, p02 = s13_2;
    const double p10 = fma(b, s13_2, a) + j2, p11 = fma(f, s13_2, e) - j2, p12 = s23_2 * c13_2;
    const double w0 = fma(ds1, p10, fma(ds0, p00, s2));
    const double w1 = fma(ds1, p11, fma(ds0, p01, s2));
    const double w2 = fma(ds1, p12, fma(ds0, p02, s2));
    const double dw0 = w0 - w2, dw1 = w1 - w2;
    fr[0] = fma(p01, dw1, fma(p00, dw0, w2));
    fr[1] = fma(p11, dw1, fma(p10, dw0, w2));
    fr[2] = (((w0 + w1) + w2) - fr[0]) - fr[1];
}

// Haar draws: angles ~ U([0,1]^3 x [0,2pi]) (the flat prior of scripts/mc_unitary.py:35-40), then
// angles_to_u -> u_to_fr(source_ratio) (mc_unitary.py:189-193).  24 B written per draw.
__global__ __launch_bounds__(GF_BLOCK) void k_haar(const GfCommon c, uint64_t seed, int64_t first, int64_t n,
                                                   double* __restrict__ angles, double* __restrict__ fr_out)
{
    const int64_t stride = (int64_t)gridDim.x * GF_BLOCK;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const double isrc = fast_rcp(c.src_fixed_sum);
    const double s2 = c.src_fixed[2] * isrc, ds0 = fma(c.src_fixed[0], isrc, -s2), ds1 = fma(c.src_fixed[1], isrc, -s2);
    for (int64_t i = (int64_t)blockIdx.x * GF_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint64_t ctr = (uint64_t)(first + i);
        uint32_t a[4], b[4];
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, k0, k1, a);
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, 0u, k0, k1, b);
        const double s12_2 = u53(a[0], a[1]);
        const double c13_4 = u53(a[2], a[3]);
        const double s23_2 = u53(b[0], b[1]);
        const double dcp = 6.283185307179586 * u53(b[2], b[3]);
        double fr[3];
        haar_composition(s12_2, c13_4, s23_2, dcp, s2, ds0, ds1, fr);
        fr_out[3 * i] = fr[0]; fr_out[3 * i + 1] = fr[1]; fr_out[3 * i + 2] = fr[2];
        if (angles) {
            double4 v; v.x = s12_2; v.y = c13_4; v.z = s23_2; v.w = dcp;
            *reinterpret_cast<double4*>(angles + 4 * i) = v;
        }
    }
}

// Flavor-triangle histogram of a block of compositions: the reduction golemflavor/plot.py:365-370 does
// with np.histogramdd(frs, bins=(nb, nb, nb), range=((0,1),)*3): nb equal bins per axis on [0, 1], the
// last bin closed on the right, samples outside the cube (or NaN) dropped.  counts is [nb][nb][nb].
__global__ __launch_bounds__(GF_BLOCK) void k_flavor_hist(const double* __restrict__ fr, int64_t n, int nb,
                                                          unsigned long long* __restrict__ counts)
{
    const int64_t stride = (int64_t)gridDim.x * GF_BLOCK;
    const double scale = (double)nb;
    for (int64_t i = (int64_t)blockIdx.x * GF_BLOCK + threadIdx.x; i < n; i += stride) {
        int idx[3];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = fr[3 * i + a];
            ok = ok && (v >= 0.0) && (v <= 1.0);
            int b = (int)(v * scale);
            idx[a] = b >= nb ? nb - 1 : b;
        }
        if (ok) atomicAdd(counts + ((int64_t)idx[0] * nb + idx[1]) * nb + idx[2], 1ull);
    }
}

// Chain post-processing output rows: out[i] = (fr[i][0..3), theta[i][0..ndim)) -- the composition of a stored sample next to
// the sample itself; status[i] != 0 (the reference would have raised on that sample, scripts/mc_texture.py:216-221 through
// fr.py:398-399) turns the composition into NaN.  Pure data movement, one thread per output element.
__global__ __launch_bounds__(GF_BLOCK) void k_join_rows(const double* __restrict__ fr, const int32_t* __restrict__ status,
                                                        const double* __restrict__ theta, int ndim, int64_t n, double* __restrict__ out)
{
    const int w = 3 + ndim;
    const int64_t total = n * w;
    for (int64_t e = (int64_t)blockIdx.x * GF_BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * GF_BLOCK) {
        const int64_t i = e / w;
        const int col = (int)(e - i * w);
        double v;
        if (col < 3) v = (status && status[i] != 0) ? gf_nan() : fr[3 * i + col];
        else v = theta[i * ndim + (col - 3)];
        out[e] = v;
    }
}

// MultiNest's unit cube -> theta (golemflavor/mn.py:33-39): the scanned columns are mapped onto their ranges,
// theta_c = lo_c + (hi_c - lo_c) u, every other column keeps its current value.  One thread per (walker, column).
struct CubeMap {
    int32_t ndim, nscan;
    int32_t slot[GF_MAX_DIM];       // column -> index into the cube row, or -1
    double lo[GF_MAX_DIM], span[GF_MAX_DIM], base[GF_MAX_DIM];
};

__global__ __launch_bounds__(GF_BLOCK) void k_cube_to_theta(const CubeMap cm, const double* __restrict__ cube, int64_t n,
                                                            double* __restrict__ theta)
{
    const int64_t total = n * cm.ndim;
    for (int64_t i = (int64_t)blockIdx.x * GF_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * GF_BLOCK) {
        const int64_t w = i / cm.ndim;
        const int col = (int)(i - w * cm.ndim);
        const int sl = cm.slot[col];
        // product, then sum, each rounded (mn.py:36 is a Python expression): no fused multiply-add here
        theta[i] = sl >= 0 ? __dadd_rn(__dmul_rn(cm.span[col], cube[w * cm.nscan + sl]), cm.lo[col]) : cm.base[col];
    }
}

#ifndef GF_BLOCKS_PER_CU
#define GF_BLOCKS_PER_CU 8
#endif
inline int grid_for(int64_t work_items, int per_block, int cus)
{
    int64_t blocks = (work_items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)cus * GF_BLOCKS_PER_CU;   // 256-thread blocks per CU that keep the chip full
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int NDIM, int MODE>
hipError_t launch_lnprob_sm_nm(const GfCommon& c, const double* ptab, const double* theta, int layout, int64_t n,
                               double* lnprob, double* fr, int32_t* status, int cus, hipStream_t s)
{
    int64_t first = 0;
    if constexpr (NDIM != 0) {
        int64_t nfull = n / GF_WAVE;
        // a small ragged batch (emcee's half-ensembles) is one launch of the generic kernel, not fast + tail
        if (n <= 2048 && n % GF_WAVE != 0) nfull = 0;
        if (layout == 0 && nfull > 0) {
#ifdef GF_EXPERIMENTAL_RING
            static const bool use_ring = gf_internal_env("GF_SM_RING", 0) != nullptr;
#else
            constexpr bool use_ring = false;
#endif
            if (use_ring && (NDIM % 2) == 0 && c.idx_sm[0] == 0 && c.idx_sm[1] == 1 && c.idx_sm[2] == 2 && c.idx_sm[3] == 3 &&
                c.idx_src[0] == 4 && c.idx_src[1] == 5 && !fr) {
#ifdef GF_EXPERIMENTAL_RING
                if constexpr (NDIM >= 6 && (NDIM % 2) == 0) {
                    static int* d_err = nullptr;
                    if (!d_err) { (void)hipMalloc((void**)&d_err, sizeof(int)); (void)hipMemset(d_err, 0, sizeof(int)); }
                    const int64_t cap = (int64_t)cus * 2;
                    const int rgrid = (int)(nfull < cap ? nfull : cap);
                    hipLaunchKernelGGL((k_lnprob_sm_ring<NDIM, MODE, 2, false>), dim3(rgrid), dim3(512), 0, s, c, ptab, theta, nfull,
                                       lnprob, fr, status, d_err);
                    first = nfull * GF_WAVE;
                }
#endif
            } else {
            const int grid = grid_for(nfull * GF_WAVE, GF_BLOCK, cus);
            const bool sampled = c.idx_sm[0] >= 0 && c.idx_sm[1] >= 0 && c.idx_sm[2] >= 0 && c.idx_sm[3] >= 0 &&
                                 c.idx_src[0] >= 0 && c.idx_src[1] >= 0;
#define GF_GO(S, F) hipLaunchKernelGGL((k_lnprob_sm_fast<NDIM, MODE, S, F>), dim3(grid), dim3(GF_BLOCK), 0, s, c, ptab, theta, nfull, lnprob, fr, status)
            bool piped = false;
#ifdef GF_ASM_PIPE
            if constexpr (NDIM == 6 || NDIM == 4) {
                const bool canon6 = sampled && NDIM >= 6 && c.idx_sm[0] == 0 && c.idx_sm[1] == 1 && c.idx_sm[2] == 2 &&
                                    c.idx_sm[3] == 3 && c.idx_src[0] == 4 && c.idx_src[1] == 5;
                if (!fr && (canon6 || MODE == MODE_PRIOR_ONLY)) {
                    if (canon6) hipLaunchKernelGGL((k_lnprob_sm_pipe<NDIM, MODE, 2>), dim3(grid), dim3(GF_BLOCK), 0, s, c, ptab, theta, nfull, lnprob, status);
                    else hipLaunchKernelGGL((k_lnprob_sm_pipe<NDIM, MODE, 0>), dim3(grid), dim3(GF_BLOCK), 0, s, c, ptab, theta, nfull, lnprob, status);
                    piped = true;
                }
            }
#endif
            const bool canon = sampled && NDIM >= 6 && c.idx_sm[0] == 0 && c.idx_sm[1] == 1 && c.idx_sm[2] == 2 &&
                               c.idx_sm[3] == 3 && c.idx_src[0] == 4 && c.idx_src[1] == 5;
            if (piped)        { }
            else if (canon)   { if (fr) GF_GO(2, true); else GF_GO(2, false); }
            else if (sampled) { if (fr) GF_GO(1, true); else GF_GO(1, false); }
            else              { if (fr) GF_GO(0, true); else GF_GO(0, false); }
#undef GF_GO
            first = nfull * GF_WAVE;
            }
        }
    }
    if constexpr (NDIM != 0) {
        // SoA: the register kernel for the posteriors that read their columns by position
        int64_t nfull = n / GF_WAVE;
        if (n <= 2048 && n % GF_WAVE != 0) nfull = 0;
        const bool canon = NDIM >= 6 && c.idx_sm[0] == 0 && c.idx_sm[1] == 1 && c.idx_sm[2] == 2 && c.idx_sm[3] == 3 &&
                           c.idx_src[0] == 4 && c.idx_src[1] == 5;
        if (layout == 1 && nfull > 0 && (MODE == MODE_PRIOR_ONLY || canon)) {
            const int grid = grid_for(nfull * GF_WAVE, GF_BLOCK, cus);
#define GF_GOS(S, F) hipLaunchKernelGGL((k_lnprob_sm_soa<NDIM, MODE, S, F>), dim3(grid), dim3(GF_BLOCK), 0, s, c, ptab, theta, n, nfull, lnprob, fr, status)
            if (MODE == MODE_PRIOR_ONLY) { if (fr) GF_GOS(0, true); else GF_GOS(0, false); }
            else                         { if (fr) GF_GOS(2, true); else GF_GOS(2, false); }
#undef GF_GOS
            first = nfull * GF_WAVE;
        }
    }
    if (first < n) {
        const int grid = grid_for(n - first, GF_BLOCK, cus);
        hipLaunchKernelGGL((k_lnprob_sm_gen<NDIM, MODE>), dim3(grid), dim3(GF_BLOCK), 0, s, c, ptab, theta, layout, first, n,
                           lnprob, fr, status);
    }
    return hipGetLastError();
}

template <int NDIM>
hipError_t launch_lnprob_sm_n(const GfCommon& c, const double* ptab, const double* theta, int layout, int64_t n,
                              double* lnprob, double* fr, int32_t* status, int cus, hipStream_t s)
{
    if (c.mode == MODE_PRIOR_ONLY)
        return launch_lnprob_sm_nm<NDIM, MODE_PRIOR_ONLY>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
    return launch_lnprob_sm_nm<NDIM, MODE_SM_GAUSS>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
}

template <int NDIM>
hipError_t launch_propagate_sm_n(const GfCommon& c, const double* theta, int layout, int64_t n, double* fr,
                                 int32_t* status, int cus, hipStream_t s)
{
    const int grid = grid_for(n, GF_BLOCK, cus);
    hipLaunchKernelGGL((k_propagate_sm<NDIM>), dim3(grid), dim3(GF_BLOCK), 0, s, c, theta, layout, n, fr, status);
    return hipGetLastError();
}

}  // namespace

hipError_t gf_launch_lnprob_sm(const GfCommon& c, const double* ptab, const double* theta, int layout, int64_t n,
                               double* lnprob, double* fr, int32_t* status, int cus, hipStream_t s)
{
    switch (c.ndim) {
    case 2: return launch_lnprob_sm_n<2>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);   // examples/tutorial.ipynb
    case 4: return launch_lnprob_sm_n<4>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
    case 6: return launch_lnprob_sm_n<6>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
    case 7: return launch_lnprob_sm_n<7>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
    case 12: return launch_lnprob_sm_n<12>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
    default: return launch_lnprob_sm_n<0>(c, ptab, theta, layout, n, lnprob, fr, status, cus, s);
    }
}

hipError_t gf_launch_cube_to_theta(const GfCommon& c, int nscan, const int32_t* cols, const double* base, const double* cube,
                                   int64_t n, double* theta, int cus, hipStream_t s)
{
    CubeMap cm;
    cm.ndim = c.ndim; cm.nscan = nscan;
    for (int d = 0; d < GF_MAX_DIM; ++d) { cm.slot[d] = -1; cm.lo[d] = 0.0; cm.span[d] = 0.0; cm.base[d] = d < c.ndim ? base[d] : 0.0; }
    for (int k = 0; k < nscan; ++k) {
        cm.slot[cols[k]] = k;
        cm.lo[cols[k]] = c.lo[cols[k]];
        cm.span[cols[k]] = c.hi[cols[k]] - c.lo[cols[k]];            // mn.py:36 (hi - lo) * cube + lo
    }
    const int grid = grid_for(n * c.ndim, GF_BLOCK, cus);
    hipLaunchKernelGGL(k_cube_to_theta, dim3(grid), dim3(GF_BLOCK), 0, s, cm, cube, n, theta);
    return hipGetLastError();
}

hipError_t gf_launch_propagate_sm(const GfCommon& c, const double* theta, int layout, int64_t n, double* fr,
                                  int32_t* status, int cus, hipStream_t s)
{
    switch (c.ndim) {
    case 4: return launch_propagate_sm_n<4>(c, theta, layout, n, fr, status, cus, s);
    case 6: return launch_propagate_sm_n<6>(c, theta, layout, n, fr, status, cus, s);
    default: return launch_propagate_sm_n<0>(c, theta, layout, n, fr, status, cus, s);
    }
}

hipError_t gf_launch_join_rows(const double* fr, const int32_t* status, const double* theta, int ndim, int64_t n, double* out,
                               int cus, hipStream_t s)
{
    const int grid = grid_for(n * (3 + ndim), GF_BLOCK, cus);
    hipLaunchKernelGGL(k_join_rows, dim3(grid), dim3(GF_BLOCK), 0, s, fr, status, theta, ndim, n, out);
    return hipGetLastError();
}

hipError_t gf_launch_flavor_hist(const double* fr, int64_t n, int nb, unsigned long long* counts, int cus, hipStream_t s)
{
    const int grid = grid_for(n, GF_BLOCK, cus);
    hipLaunchKernelGGL(k_flavor_hist, dim3(grid), dim3(GF_BLOCK), 0, s, fr, n, nb, counts);
    return hipGetLastError();
}

hipError_t gf_launch_haar(const GfCommon& c, uint64_t seed, int64_t first, int64_t n, double* angles, double* fr,
                          int cus, hipStream_t s)
{
    const int grid = grid_for(n, GF_BLOCK, cus);
    hipLaunchKernelGGL(k_haar, dim3(grid), dim3(GF_BLOCK), 0, s, c, seed, first, n, angles, fr);
    return hipGetLastError();
}
