// gf_device.hpp -- device-side building blocks shared by the kernels (gf_kernels.hip, gf_bsm.hip).
// Reference formulas are cited per function (file:line under the reference tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf_consts.h"

#define GF_WAVE 64
#define GF_BLOCK 256
#define GF_WAVES_PER_BLOCK (GF_BLOCK / GF_WAVE)

#ifdef GF_COS_LITERALS            // A/B switch: polynomial coefficients as literals instead of kernarg SGPRs
#define GF_COSC(c) nullptr
#else
#define GF_COSC(c) (c).cosc
#endif

namespace gfdev {

constexpr int ST_OK = 0, ST_OUT_OF_PRIOR = 1, ST_NON_UNITARY = 2, ST_NAN = 3;
constexpr int MODE_PRIOR_ONLY = 0, MODE_SM_GAUSS = 1, MODE_BSM_GAUSS = 2;

__device__ __forceinline__ double gf_inf() { return __longlong_as_double(0x7ff0000000000000LL); }
__device__ __forceinline__ double gf_nan() { return __longlong_as_double(0x7ff8000000000000LL); }

// ---------------------------------------------------------------------------------------------
// fp64 building blocks.  The compiler's sqrt()/division/cos() lower to IEEE-complete sequences
// (input scaling for denormals, class fix-ups, Payne-Hanek argument reduction) that cost 2x the
// instructions this path needs; the forms below keep <= 1 ulp on the domain the physics feeds them and
// return NaN outside it.  Parity is checked end to end against the long-double oracle (1e-10 bar,
// ~1e-15 observed).

// sqrt for x in [0, ~1e300): v_rsq_f64 seed + two Newton-Raphson (Goldschmidt) refinements, no denormal
// scaling.  x == 0 -> 0; x < 0 or NaN -> NaN.
__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}

// 1/x for finite normal x: v_rcp_f64 seed + two Newton-Raphson steps.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// library sin/cos for huge or non-finite arguments, out of line (Payne-Hanek needs ~100 registers)
static __device__ __attribute__((noinline)) void sincos_cold(double x, double* sn, double* cs) { sincos(x, sn, cs); }

// sin and cos of x, |x| < 2^20 * pi/2: three-step Cody-Waite reduction with FMA (33-bit pieces of pi/2,
// the published fdlibm split), then the fdlibm minimax kernels on [-pi/4, pi/4].  Absolute error
// <= ~2e-16.  Larger |x| (never a physical phase) give NaN.
__device__ __forceinline__ void fast_sincos(double x, double* sn, double* cs)
{
    // beyond the reduction's range (never a physical phase): NaN -> GF_ST_NAN.  gf_model_create refuses models
    // whose phase columns could get here (GF_PHASE_MAX), so that the library's Payne-Hanek path -- ~100 VGPRs and
    // a scratch frame for every kernel that can reach it -- stays out of the hot kernels' register budget.
    if (!(fabs(x) < 1.6e6)) { *sn = gf_nan(); *cs = gf_nan(); return; }
    const double fn = rint(x * 6.36619772367581382433e-01);          // x * 2/pi
    double r = fma(-fn, 1.57079632673412561417e+00, x);              // pio2_1 (33 bits)
    r = fma(-fn, 6.07710050630396597660e-11, r);                     // pio2_2 (33 bits)
    r = fma(-fn, 2.02226624879595063154e-21, r);                     // pio2_2t: the rest of pi/2
    const int q = (int)fn;
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    // quadrant: (sin, cos)(x) = (s, c), (c, -s), (-s, -c), (-c, s) for q mod 4 = 0, 1, 2, 3
    const double ss = (q & 1) ? c : s;
    const double cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ double fast_cos(double x)
{
    double s, c;
    fast_sincos(x, &s, &c);
    return c;
}

// Cold, out-of-line general cosine (keeps its literals out of the hot loop's register budget).
static __device__ __attribute__((noinline)) double cos_general(double x) { return fast_cos(x); }

// cos(x) for a CP phase.  Every paramset of the reference boxes dcp into [0, 2 pi]
// (scripts/fr.py:41, examples/inference.ipynb:253), so the fast path folds that interval onto
// u = |x - pi| - pi/2 in [-pi/2, pi/2] where cos x = sin u, and evaluates one odd minimax polynomial
// (9 terms, fitted in 50-digit arithmetic: approximation error 3e-19, end-to-end abs error <= 2.8e-16).
// Any other x takes cos_general.
__device__ __forceinline__ double fast_cos_phase(double x, const double* k = nullptr)
{
    const double u = fabs(x - 3.141592653589793) - 1.5707963267948966;
    if (!(fabs(u) <= 1.5707963277948966)) return cos_general(x);     // pi/2 + 1e-9; NaN goes here too
    const double z = u * u;
    double q;
    if (k) {                               // coefficients from kernel-argument SGPRs (GfCommon::cosc)
        q = fma(z, k[0], k[1]);
        q = fma(z, q, k[2]);
        q = fma(z, q, k[3]);
        q = fma(z, q, k[4]);
        q = fma(z, q, k[5]);
        q = fma(z, q, k[6]);
        q = fma(z, q, k[7]);
    } else {
        q = fma(z, 2.7117413873509064e-15, -7.641995277350052e-13);
        q = fma(z, q, 1.605889634387573e-10);
        q = fma(z, q, -2.505210587009456e-08);
        q = fma(z, q, 2.75573191979119e-06);
        q = fma(z, q, -0.00019841269841110079);
        q = fma(z, q, 0.008333333333332799);
        q = fma(z, q, -0.16666666666666657);
    }
    return fma(u * z, q, u);
}

// Minimax / fdlibm coefficients of sincos_small and fast_acos in constant memory: read with uniform
// addresses they arrive by scalar loads and feed v_fma_f64 as its one scalar operand.  As literals the
// compiler keeps each in a VGPR pair across the BSM bin loop and copies it (v_mov_b64) in front of every
// Horner step (v_fmac overwrites its addend): 22 extra instructions per bin, ~50 VGPRs.
__constant__ double GF_KTAB[26] = {
    // [0..7]  sin: odd minimax, highest order first
    2.7117413873509064e-15, -7.641995277350052e-13, 1.605889634387573e-10, -2.505210587009456e-08,
    2.75573191979119e-06, -0.00019841269841110079, 0.008333333333332799, -0.16666666666666657,
    // [8..15] cos: even minimax, highest order first (the constant term 1 is applied last)
    4.695137892341541e-14, -1.1468837412893726e-11, 2.0876733673699156e-09, -2.755731905787651e-07,
    2.4801587300891894e-05, -0.0013888888888887252, 0.04166666666666665, -0.5,
    // [16..21] asin numerator pS5..pS0, [22..25] denominator qS4..qS1 (fdlibm e_asin.c)
    3.47933107596021167570e-05, 7.91534994289814532176e-04, -4.00555345006794114027e-02, 2.01212532134862925881e-01,
    -3.25565818622400915405e-01, 1.66666666666666657415e-01,
    7.70381505559019352791e-02, -6.88283971605453293030e-01, 2.02094576023350569471e+00, -2.40339491173441421878e+00};

// sin and cos of phi in [0, 1.1] (the cubic solver's angle acos(x)/3 <= pi/3): no argument reduction;
// odd / even minimax polynomials fitted in 50-digit arithmetic (abs error <= 2.3e-16 each).
__device__ __forceinline__ void sincos_small(double u, double* sn, double* cs)
{
    const double* k = GF_KTAB;
    const double z = u * u;
    double q = fma(z, k[0], k[1]);
    q = fma(z, q, k[2]);
    q = fma(z, q, k[3]);
    q = fma(z, q, k[4]);
    q = fma(z, q, k[5]);
    q = fma(z, q, k[6]);
    q = fma(z, q, k[7]);
    *sn = fma(u * z, q, u);
    double c = fma(z, k[8], k[9]);
    c = fma(z, c, k[10]);
    c = fma(z, c, k[11]);
    c = fma(z, c, k[12]);
    c = fma(z, c, k[13]);
    c = fma(z, c, k[14]);
    c = fma(z, c, k[15]);
    *cs = fma(z, c, 1.0);
}

// acos(x), x in [-1, 1]: the fdlibm scheme (asin(s) = s + s R(s^2), R = z p(z)/q(z); |x| > 1/2 goes through
// s = sqrt((1-|x|)/2)) written branch-free.  Abs error a few 1e-16 rad.
__device__ __forceinline__ double fast_acos(double x)
{
    const double* k = GF_KTAB + 16;
    const double ax = fabs(x);
    const bool big = ax > 0.5;
    const double z = big ? fma(-0.5, ax, 0.5) : x * x;
    const double s = big ? fast_sqrt(z) : ax;
    double p = fma(z, k[0], k[1]);
    p = fma(z, p, k[2]);
    p = fma(z, p, k[3]);
    p = fma(z, p, k[4]);
    p = fma(z, p, k[5]);
    double q = fma(z, k[6], k[7]);
    q = fma(z, q, k[8]);
    q = fma(z, q, k[9]);
    q = fma(z, q, 1.0);
    const double r = (z * p) * fast_rcp(q);
    const double as = fma(s, r, s);                                  // asin(s)
    const double a = big ? as + as : 1.5707963267948966 - as;        // acos(|x|)
    return x < 0.0 ? 3.141592653589793 - a : a;
}

// the same for an argument that rounding may have pushed a hair beyond [-1, 1] (the cubic's R / Q^(3/2)): taken as +-1.  One
// max on z where a clamp of x costs a min and a max: |x| > 1 only shows up in the big branch, as a negative z.
__device__ __forceinline__ double fast_acos_clamped(double x)
{
    const double* k = GF_KTAB + 16;
    const double ax = fabs(x);
    const bool big = ax > 0.5;
    const double z = big ? fmax(fma(-0.5, ax, 0.5), 0.0) : x * x;
    const double s = big ? fast_sqrt(z) : ax;
    double p = fma(z, k[0], k[1]);
    p = fma(z, p, k[2]);
    p = fma(z, p, k[3]);
    p = fma(z, p, k[4]);
    p = fma(z, p, k[5]);
    double q = fma(z, k[6], k[7]);
    q = fma(z, q, k[8]);
    q = fma(z, q, k[9]);
    q = fma(z, q, 1.0);
    const double r = (z * p) * fast_rcp(q);
    const double as = fma(s, r, s);                                  // asin(s)
    const double a = big ? as + as : 1.5707963267948966 - as;        // acos(|x|)
    return x < 0.0 ? 3.141592653589793 - a : a;
}

// 10^x for the NP scale (fr.py:380 np.power(10., logLam)), once per walker.  The library pow(10, x) is ~200 instructions
// behind a call; for |x| < 300 (every scale a paramset of the reference can hold: -72 ... -20) this is 22: n = rint(x log2 10),
// r = x - n log10(2) by a two-constant Cody-Waite reduction (the high part has 33 bits: n * hi is exact), 10^r as the degree-14
// Taylor polynomial of exp(r ln 10) on |r ln 10| <= 0.347 (truncation 4e-18), times 2^n.  <= 2 ulp against 50-digit values on
// 20 000 arguments; the value enters the Hamiltonian linearly, the parity bar is 1e-10.  (The x87-faithful arbitration has
// its own, correctly rounded 10^x: gf_x87.hpp.)  Anything else, NaN included, takes the library call.
static __device__ __attribute__((noinline)) double pow10_cold(double x) { return pow(10.0, x); }
#ifdef GF_POW_LIBRARY           // A/B switch (tools/build_variants.sh): the library call for every argument
__device__ __forceinline__ double pow10_scale(double x) { return pow10_cold(x); }
#else
// (constants in constant memory for the reason given at GF_KTAB: as literals they would sit in 36 VGPRs across the bin loop)
__constant__ double GF_P10TAB[18] = {
    3.321928094887362, 0.30102999560767785, 5.630334806675098e-11,      // log2(10); log10(2): high 33 bits, the rest
    1.3508629476223687e-06, 8.213412535439387e-06, 4.6371516642572196e-05, 0.00024166672554424694,   // ln(10)^k / k!, k = 14 ... 0
    0.0011544997789984348, 0.00501392883377544, 0.019597694626478524, 0.06808936507443707, 0.2069958486968681,
    0.5393829291955814, 1.171255148912267, 2.034678592293476, 2.650949055239199, 2.302585092994046, 1.0};
__device__ __forceinline__ double pow10_scale(double x)
{
    if (!(fabs(x) < 300.0)) return pow10_cold(x);
    const double* k = GF_P10TAB;
    const double n = rint(x * k[0]);
    double r = fma(-n, k[1], x);
    r = fma(-n, k[2], r);
    double p = k[3];
#pragma unroll
    for (int j = 4; j < 18; ++j) p = fma(p, r, k[j]);
    return ldexp(p, (int)n);
}
#endif

// ---------------------------------------------------------------------------------------------
// Stage one wave's 64 x ndim block of theta into its LDS tile (row-major [64][ndim]).
// AoS: the block is contiguous in memory -> 16-B vector loads, lane-contiguous.
// SoA: each lane loads its own ndim values (8-B, lane-contiguous per column).
template <int NDIM>
// `n` is the batch size (the SoA column stride); rows [w0, min(n_stage, w0 + 64)) are staged (n_stage < 0: n).
__device__ __forceinline__ void stage_theta(const double* __restrict__ theta, int layout, int64_t n,
                                            int64_t w0, int ndim_rt, double* tile, int lane, int64_t n_stage = -1)
{
    const int ndim = NDIM ? NDIM : ndim_rt;
    const int64_t nlim = n_stage < 0 ? n : n_stage;
    const int64_t rows = (nlim - w0 < GF_WAVE) ? (nlim - w0) : GF_WAVE;
    if (layout == 0) {
        const int64_t count = rows * ndim;                    // doubles in this wave's span
        const double* src = theta + w0 * ndim;                // 512*ndim-byte aligned relative to theta
        const int nvec = (GF_WAVE * ndim + 1) / 2;            // 16-B vectors in a full tile
#pragma unroll
        for (int j = 0; j < (NDIM ? (NDIM + 1) / 2 : (GF_MAX_DIM / 2)); ++j) {
            const int v = j * GF_WAVE + lane;
            if (!NDIM && v >= nvec) break;
            const int64_t d = 2 * (int64_t)v;
            if (d + 1 < count) {
                const double2 x = *reinterpret_cast<const double2*>(src + d);
                *reinterpret_cast<double2*>(tile + d) = x;
            } else if (d < count) {
                tile[d] = src[d];
            }
        }
    } else {
        if (lane < rows) {
#pragma unroll
            for (int d = 0; d < (NDIM ? NDIM : GF_MAX_DIM); ++d) {
                if (!NDIM && d >= ndim) break;
                tile[lane * ndim + d] = theta[(int64_t)d * n + w0 + lane];
            }
        }
    }
    // same-wave LDS write -> read: in-order in hardware; keep the compiler from reordering.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// golemflavor/llh.py:65-91 lnprior.  Closed box (NaN fails), then sum of truncated-normal logpdfs:
// scipy: ((-z^2/2 - log sqrt(2pi)) - log_mass) - log sigma, z = (x - loc)/sigma; the three constants
// are pre-summed on the host (GfCommon::prior_const).  UNIFORM columns have inv_sigma = 0.
template <int NDIM>
__device__ __forceinline__ bool lnprior(const GfCommon& c, const double* row, double& lp)
{
    const int ndim = NDIM ? NDIM : c.ndim;
    bool inbox = true;
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < (NDIM ? NDIM : GF_MAX_DIM); ++d) {
        if (!NDIM && d >= ndim) break;
        const double x = row[d];
        inbox = inbox && (x >= c.lo[d]) && (x <= c.hi[d]);
        const double z = (x - c.loc[d]) * c.inv_sigma[d];
        acc = fma(-0.5 * z, z, acc);
    }
    lp = acc + c.prior_const;
    return inbox;
}

// ---------------------------------------------------------------------------------------------
// |U_ai|^2 of the PMNS matrix from (s12^2, c13^4, s23^2, delta): golemflavor/fr.py:116-162 in the
// algebraic form of SURVEY.md A.2.  The decoherent propagation only needs the moduli, which depend on
// delta through cos(delta) alone:
//   |Ue1|^2 = c12^2 c13^2, |Ue2|^2 = s12^2 c13^2, |Ue3|^2 = s13^2,
//   |Um3|^2 = s23^2 c13^2, |Ut3|^2 = c23^2 c13^2,
//   |Um1|^2 = s12^2 c23^2 + c12^2 s23^2 s13^2 + 2 J cos d,   |Um2|^2 = c12^2 c23^2 + s12^2 s23^2 s13^2 - 2 J cos d,
//   |Ut1|^2 = s12^2 s23^2 + c12^2 c23^2 s13^2 - 2 J cos d,   |Ut2|^2 = c12^2 s23^2 + s12^2 c23^2 s13^2 + 2 J cos d,
// J = s12 c12 s23 c23 s13 = sqrt(s12^2 c12^2 s23^2 c23^2 s13^2)   (all five factors are >= 0: the
// reference takes the angles in [0, pi/2] via asin/acos of a square root, fr.py:145-147).
__device__ __forceinline__ void pmns_abs2(double s12_2, double c13_4, double s23_2, double dcp, double p[3][3],
                                          const double* cosc = nullptr)
{
    const double c13_2 = fast_sqrt(c13_4);
    const double s13_2 = 1.0 - c13_2;
    const double c12_2 = 1.0 - s12_2;
    const double c23_2 = 1.0 - s23_2;
    const double a = s12_2 * c23_2, b = c12_2 * s23_2;
    const double e = c12_2 * c23_2, f = s12_2 * s23_2;
    const double j2 = 2.0 * fast_sqrt((a * b) * s13_2) * fast_cos_phase(dcp, cosc);
    p[0][0] = c12_2 * c13_2;
    p[0][1] = s12_2 * c13_2;
    p[0][2] = s13_2;
    p[1][0] = fma(b, s13_2, a) + j2;
    p[1][1] = fma(f, s13_2, e) - j2;
    p[1][2] = s23_2 * c13_2;
    // (the tau row |Ut1|^2 = e s13^2 + f - 2 J cos d, |Ut2|^2 = a s13^2 + b + 2 J cos d, |Ut3|^2 = c23^2 c13^2 is not formed: the
    // columns of |U|^2 sum to one and `propagate` uses that)
}

// golemflavor/fr.py:82-113 angles_to_fr: (sin^4 phi, cos 2psi) -> composition.  sin^2(acos(c)/2) =
// (1-c)/2 exactly, so no trigonometry is needed.
__device__ __forceinline__ void angles_to_fr(double sphi4, double c2psi, double f[3])
{
    const double sphi2 = fast_sqrt(sphi4);
    const double spsi2 = 0.5 * (1.0 - c2psi);
    const double cpsi2 = 1.0 - spsi2;
    f[0] = fabs(sphi2 * cpsi2);
    f[1] = fabs(sphi2 * spsi2);
    f[2] = fabs(1.0 - sphi2);
}

// golemflavor/fr.py:502-536 u_to_fr: out_b = sum_a sum_i |U_ai|^2 |U_bi|^2 src_a / sum(src) = (P P^T s)_b with P = |U|^2 and
// s = src / sum(src).  The rows and columns of P sum to one, so its e and mu rows carry everything:
//   w = P^T s = s_tau + (s_e - s_tau) P_e. + (s_mu - s_tau) P_mu.          (6 FMAs)
//   out_b = w_3 + P_b1 (w_1 - w_3) + P_b2 (w_2 - w_3),  b = e, mu         (2 + 4)
//   out_tau = (w_1 + w_2 + w_3) - out_e - out_mu                           (4)
// 16 instructions behind the normalisation where the two full 3x3 matrix-vector products took 18 plus the tau row of P (5).
// `p`: rows 0 and 1 only are read.
__device__ __forceinline__ void propagate(const double p[3][3], const double src[3], double src_sum, double out[3])
{
    const double inv = fast_rcp(src_sum);
    const double s2 = src[2] * inv;
    const double ds0 = fma(src[0], inv, -s2), ds1 = fma(src[1], inv, -s2);
    const double w0 = fma(ds1, p[1][0], fma(ds0, p[0][0], s2));
    const double w1 = fma(ds1, p[1][1], fma(ds0, p[0][1], s2));
    const double w2 = fma(ds1, p[1][2], fma(ds0, p[0][2], s2));
    const double dw0 = w0 - w2, dw1 = w1 - w2;
    out[0] = fma(p[0][1], dw1, fma(p[0][0], dw0, w2));
    out[1] = fma(p[1][1], dw1, fma(p[1][0], dw0, w2));
    out[2] = (((w0 + w1) + w2) - out[0]) - out[1];
}

// golemflavor/llh.py:32-54 multi_gaussian = log(mvn.pdf) + offset.  scipy evaluates
// logpdf = -0.5 (3 log 2pi + log_pdet + maha) and then exp(); the reference takes log() of that, so
// below the fp64 underflow wall the value is quantised (subnormal band, logpdf in (-745.13, -708.40))
// or -inf.  Emulated: y = exp(logpdf) / 2^-1074 rounded to an integer count of subnormal ulps.
// Cold path, deliberately not inlined: exp()/log() bring ~25 fp64 literals whose materialisation the
// compiler would otherwise hoist out of the tile loop and keep live in (scarce) scalar registers.
static __device__ __attribute__((noinline)) double log_of_exp_band(double x)
{
    const double HI = 744.4400719213812, LO = 4.422444340918698e-14;   // 1074 ln 2 = HI + LO
    const double t = x + HI;                            // exact (both multiples of 2^-43, |t| < 64)
    const double y = exp(t) * (1.0 + LO);
    const double k = rint(y);
    if (!(k >= 1.0)) return (x != x) ? x : -gf_inf();   // underflows to zero -> log(0) = -inf
    return (log(k) - HI) - LO;
}

__device__ __forceinline__ double log_of_exp(double x, bool live = true)
{
    // exp(x) is a normal number above the threshold: log(exp(x)) == x to 1 ulp.  `live` = false lanes
    // (walkers already rejected by the box prior) never take the cold path.
    if (!live || x >= -708.3964185322641) return x;
    return log_of_exp_band(x);
}

__device__ __forceinline__ double gauss_llh(const GfCommon& c, const double fr[3], bool live = true)
{
    // maha = sum ((fr - bf) / smearing)^2;  logpdf = -0.5 (c0 + maha) = fma(-0.5/smearing^2, |fr-bf|^2, -0.5 c0)
    const double d0 = fr[0] - c.bf[0];
    const double d1 = fr[1] - c.bf[1];
    const double d2 = fr[2] - c.bf[2];
    const double r2 = fma(d2, d2, fma(d1, d1, d0 * d0));
    const double logpdf = fma(c.gauss_mh, r2, c.gauss_k);
    return log_of_exp(logpdf, live) + c.offset;
}

__device__ __forceinline__ double pick(const double* row, int idx, double fixed)
{
    return idx >= 0 ? row[idx] : fixed;                 // idx is wave-uniform (kernel argument)
}

// SM measured composition for one walker: examples/inference.ipynb:316-328
__device__ __forceinline__ void sm_composition(const GfCommon& c, const double* row, double fr[3])
{
    double p[3][3];
    pmns_abs2(pick(row, c.idx_sm[0], c.sm_fixed[0]), pick(row, c.idx_sm[1], c.sm_fixed[1]),
              pick(row, c.idx_sm[2], c.sm_fixed[2]), pick(row, c.idx_sm[3], c.sm_fixed[3]), p);
    double src[3], src_sum;
    if (c.idx_src[0] >= 0) {
        angles_to_fr(row[c.idx_src[0]], row[c.idx_src[1]], src);
        src_sum = (src[0] + src[1]) + src[2];
    } else if (c.idx_src_x >= 0) {
        // scripts/mc_x.py:186-190: srcs = normalize_fr((x, 1 - x, 0)) = (x, 1 - x, 0) / float(sum), then u_to_fr
        // divides by the sum of that again (fr.py:535)
        const double x = row[c.idx_src_x], y = 1.0 - x;
        const double inv = fast_rcp((x + y) + 0.0);
        src[0] = x * inv; src[1] = y * inv; src[2] = 0.0;
        src_sum = (src[0] + src[1]) + src[2];
    } else {
        src[0] = c.src_fixed[0]; src[1] = c.src_fixed[1]; src[2] = c.src_fixed[2];
        src_sum = c.src_fixed_sum;
    }
    propagate(p, src, src_sum, fr);
}


// one walker: box + priors from the LDS constant table, then the mode's likelihood.  Branch-free: the
// likelihood of an out-of-box walker is computed and discarded (a wave runs it anyway if any lane is
// inside), which keeps the LDS reads and the fp64 chain in one straight-line block.
// SAMPLED = 1: every mixing parameter and both source angles are columns of theta (drops the fixed-value
// selects and their scalar constants); 2: additionally in the canonical order s12, c13, s23, dcp, src1,
// src2 = columns 0..5 (the notebook posterior), so no named re-reads from LDS; 0: general.
template <int NDIM, int MODE, int SAMPLED, bool WANT_FR>
__device__ __forceinline__ void eval_walker(const GfCommon& c, const double* ctab, const double* row, int ndim_rt,
                                            double& val, double fr[3], int& st)
{
    const int ndim = NDIM ? NDIM : ndim_rt;
    bool inbox = true;
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < (NDIM ? NDIM : GF_MAX_DIM); ++d) {
        if (!NDIM && d >= ndim) break;
        const double x = row[d];
        const double2 lh = *reinterpret_cast<const double2*>(ctab + 4 * d);       // lo, hi
        const double2 ls = *reinterpret_cast<const double2*>(ctab + 4 * d + 2);   // loc, 1/sigma
        inbox = inbox & (x >= lh.x) & (x <= lh.y);                                 // llh.py:74-78 (NaN fails)
        const double z = (x - ls.x) * ls.y;
        acc = fma(-0.5 * z, z, acc);                                               // llh.py:81-90
#ifdef GF_DIM_BARRIER
        if ((d % GF_DIM_BARRIER) == GF_DIM_BARRIER - 1) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    const double lp = acc + c.prior_const;
    double v;
    if (MODE == MODE_PRIOR_ONLY) {
        v = lp + c.flat_llh;                             // mc_unitary.py:131,139
        if (WANT_FR) fr[0] = fr[1] = fr[2] = gf_nan();
    } else {
        double f[3];
        if (SAMPLED == 2) {                              // canonical columns 0..5: reuse the row registers
            double p[3][3], src[3];
            pmns_abs2(row[0], row[1], row[2], row[3], p, GF_COSC(c));
            angles_to_fr(row[4], row[5], src);
            propagate(p, src, (src[0] + src[1]) + src[2], f);
        } else if (SAMPLED == 1) {
            double p[3][3], src[3];
            pmns_abs2(row[c.idx_sm[0]], row[c.idx_sm[1]], row[c.idx_sm[2]], row[c.idx_sm[3]], p, GF_COSC(c));
            angles_to_fr(row[c.idx_src[0]], row[c.idx_src[1]], src);
            propagate(p, src, (src[0] + src[1]) + src[2], f);
        } else {
            sm_composition(c, row, f);
        }
        v = lp + gauss_llh(c, f, inbox);                 // ipynb:364
        if (WANT_FR) {
            fr[0] = inbox ? f[0] : gf_nan(); fr[1] = inbox ? f[1] : gf_nan(); fr[2] = inbox ? f[2] : gf_nan();
        }
    }
    val = inbox ? v : -gf_inf();                         // llh.py:78 / ipynb:360-361
    st = inbox ? ((v != v) ? ST_NAN : ST_OK) : ST_OUT_OF_PRIOR;
}


}  // namespace gfdev
