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
// everything of the emulated chain inline in this translation unit: the out-of-line division / square root / sine series of
// gf_x87.hpp's default cost k_uni_resolve a call frame (168 B of scratch per lane) and 5 % of its time (profiles/r03/ab_arbitration.txt)
#define GFX87_INLINE_ALL
#include "gf_x87.hpp"

namespace {
using namespace gfx87;

constexpr int TEX_NONE = 4;
constexpr int ST_NON_UNITARY = 2;
constexpr int UNI_BLOCK = 128;
#ifndef GF_UNI_WAVES
#define GF_UNI_WAVES 2                                // waves per SIMD k_uni_resolve is compiled for (~245 VGPRs)
#endif
#ifndef GF_UNI_BLOCKS_PER_CU
#define GF_UNI_BLOCKS_PER_CU (GF_UNI_WAVES * 4 * 64 / UNI_BLOCK)    // what is resident at once: later blocks would find the queue empty
#endif

// `ndim`: the row stride of an AoS block (the sampler's parked proposals are rows of GF_PEND_STRIDE doubles)
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

// ---- three lanes per walker -----------------------------------------------------------------------------------------------
// The chain of gf_x87.hpp (angles_to_u, sandwich, bin_residual / cardano_residual: the serial statement of the reference's
// arithmetic, which the host build of that header checks against the CPU's x87 unit) distributed over the three lanes
// r = 0, 1, 2 of a group WITHOUT changing a single operation or its order: every quantity below is computed by exactly the
// expression the serial chain uses, only by the lane that owns it -- lane r owns row r of the 3x3 matrices, the r-th term of
// the three-term sums (tr H^2, det), eigenvalue r and eigenvector r, two of the six entries of |X X^+| -- and what the other
// lanes need travels through a 400-byte slot of LDS per group.  The scalar part (cubic coefficients, the arccosine) runs
// redundantly on all three.  Matrices live in registers (a lane's rows) and LDS (H, then X): no scratch, and a walker's
// critical path is ~2.5x shorter than on one lane.  tests/test_gpu_unitarity_r3.py compares the residuals with the serial
// chain's, bit for bit.
constexpr int GRP = 3;                               // lanes per walker
constexpr int GRP_PER_WAVE = 64 / GRP;               // 21 (lane 63 idles)
constexpr int GRP_DOUBLES = 50;                      // M[9] (36 doubles) + ex[3] (12) + 2 of padding: 400 B, LDS bank step 36
struct Grp {
    cx87* M;                                         // [9] H (row-major), later X; during the set-up: exchange space
    cx87* ex;                                        // [3] exchange
    int r;
};

__device__ __forceinline__ void grp_sync()
{
    // the lanes of a group sit in one wave: its LDS operations execute in order; the fences keep the compiler from moving
    // this lane's accesses across the point where another lane's data is expected
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// row r of angles_to_u (fr.py:116-162; gf_x87.hpp angles_to_u).  Lane r evaluates "its" angle (theta12, theta13, theta23);
// asin and acos share their one expensive step (dd_asin_small), so the three lanes run the same instructions.
__device__ __forceinline__ void grp_angles_to_u_row(const Grp& g, const double ang[4], cx87 urow[3])
{
    const int r = g.r;
    // (selects, not ang[r]: a run-time index would put the array into scratch)
    x87 a = x_sqrt(x_from(r == 0 ? ang[0] : (r == 1 ? ang[1] : ang[2])));   // sqrt(s12^2) | c13^2 = sqrt(c13^4) | sqrt(s23^2)   fr.py:141,145-147
    if (r == 1) a = x_sqrt(a);                                        // sqrt(c13^2)
    // x_asin(a) / x_acos(a) for a >= 0, as dd_asin / dd_acos spell them out
    const dd da = as_dd(a);
    const bool small = da.hi <= 0.72;
    const dd arg = small ? da : dd_cofunc(da);
    const dd as = dd_asin_small(arg);
    const bool complement = (r == 1) == small;                        // asin: beyond 0.72; acos: up to 0.72
    const x87 t = round64(complement ? dd_sub(dd_pio2(), as) : as);
    x87 sn, cs, sd, cd;
    x_sincos(t, sn, cs);                                              // fr.py:149-154
    x_sincos(x_from(ang[3]), sd, cd);                                 // exp(+-i dcp) = (cos, +-sin)
    g.ex[r] = c_make(sn, cs);
    grp_sync();
    const cx87 e12 = g.ex[0], e13 = g.ex[1], e23 = g.ex[2];
    grp_sync();
    const x87 s12 = e12.re, c12 = e12.im, s13 = e13.re, c13 = e13.im, s23 = e23.re, c23 = e23.im;
    const cx87 em = c_make(cd, x_neg(sd)), ep = c_make(cd, sd);
    const cx87 s13em = c_scale(s13, em);                              // p2[0][2]
    const cx87 ms13ep = c_scale(x_neg(s13), ep);                      // p2[2][0]
    const x87 zero = x_from(0.0);
    // T = p1 . p2, row r
    const x87 fa = r == 1 ? s23 : c23, fb = r == 1 ? c23 : x_neg(s23);
    cx87 T0 = c_scale(fa, ms13ep), T1 = c_make(fb, zero), T2 = c_make(x_mul(fa, c13), zero);
    if (r == 0) { T0 = c_make(c13, zero); T1 = c_zero(); T2 = s13em; }
    // u = T . p3
    const x87 ms12 = x_neg(s12);
    urow[0] = c_add(c_scale(c12, T0), c_scale(ms12, T1));
    urow[1] = c_add(c_scale(s12, T0), c_scale(c12, T1));
    urow[2] = T2;
}

__device__ __forceinline__ void grp_load_row(const double* hi, const double* lo, int r, cx87 urow[3])
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = 3 * r + j;
        const x87 re = {hi[2 * k], lo[2 * k]}, im = {hi[2 * k + 1], lo[2 * k + 1]};
        urow[j] = c_make(re, im);
    }
}

// row r of U diag(0, w1, w2) U^+ (gf_x87.hpp sandwich): (diag . U^+)[1][j] and [2][j] come from lane j
__device__ __forceinline__ void grp_sandwich_row(const Grp& g, const cx87 urow[3], double w1, double w2, cx87 out[3])
{
    const x87 xw1 = x_from(w1), xw2 = x_from(w2);
    g.M[g.r] = c_scale(xw1, c_conj(urow[1]));
    g.M[3 + g.r] = c_scale(xw2, c_conj(urow[2]));
    grp_sync();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const cx87 t1 = g.M[j], t2 = g.M[3 + j];
        out[j] = c_add(c_mul(urow[1], t1), c_mul(urow[2], t2));
    }
    grp_sync();
}

// the walker's part of fr.py:380-399: row r of hsm and hnp
__device__ __forceinline__ void grp_walker_terms(const Grp& g, const GfCommon& c, const GfBsm& tb, const double* __restrict__ theta,
                                                            int layout, int64_t n, int64_t i, cx87 hs[3], cx87 hn[3], int stride = 0)
{
    const int ndim = stride ? stride : c.ndim;                          // row stride of the block
    cx87 urow[3];
    if (c.idx_sm[0] >= 0) {                                             // fr.py:425-431: all six from theta, or none
        double ang[4];
        for (int q = 0; q < 4; ++q) ang[q] = row_value(theta, layout, n, ndim, i, c.idx_sm[q]);
        grp_angles_to_u_row(g, ang, urow);
    } else {
        grp_load_row(tb.smu_hi, tb.smu_lo, g.r, urow);                  // fr.py:435 NUFIT_U (or the fixed angles)
    }
    const double m21 = c.idx_mass[0] >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_mass[0]) : c.mass_fixed[0];
    const double m3x = c.idx_mass[1] >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_mass[1]) : c.mass_fixed[1];
    grp_sandwich_row(g, urow, m21, m3x, hs);                            // fr.py:383-386 (before the 1/2E factor)
    if (tb.texture == TEX_NONE && c.idx_mm[0] >= 0) {                   // fr.py:378, 390
        double ang[4];
        for (int q = 0; q < 4; ++q) ang[q] = row_value(theta, layout, n, ndim, i, c.idx_mm[q]);
        grp_angles_to_u_row(g, ang, urow);
    } else {
        grp_load_row(tb.npu_hi, tb.npu_lo, g.r, urow);
    }
    const double ll = c.idx_scale >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_scale) : c.scale_fixed;
    const double sc2 = cr_pow10(ll);                                    // fr.py:380 np.power(10., sc2), fp64, correctly rounded
    const double sc1 = sc2 / 100.0;                                     // fr.py:381
    grp_sandwich_row(g, urow, sc1, sc2, hn);                            // fr.py:391-394 (before the E^(d-3) factor)
}

// One energy bin (gf_x87.hpp bin_residual + cardano_residual, fr.py:170-237 and 489-494) on the group's three lanes; every
// lane returns the same residual.
__device__ __forceinline__ double grp_bin_residual(const Grp& g, const cx87 hs[3], const cx87 hn[3], double pre, double epow,
                                                   long long* tick = nullptr)      // tick: diagnostics (k_uni_debug_group)
{
    const int r = g.r;
    cx87* M = g.M;
    // bin_residual: the power of two that brings the larger diagonal entry to magnitude one
    {
        double* exd = reinterpret_cast<double*>(g.ex);
        const double hsd = r == 0 ? hs[0].re.hi : (r == 1 ? hs[1].re.hi : hs[2].re.hi);      // the diagonal entry of this lane's row
        const double hnd = r == 0 ? hn[0].re.hi : (r == 1 ? hn[1].re.hi : hn[2].re.hi);      // (selects: no run-time register index)
        exd[2 * r] = fabs(pre * hsd);
        exd[2 * r + 1] = fabs(epow * hnd);
    }
    grp_sync();
    double big = 0.0;
    {
        const double* exd = reinterpret_cast<const double*>(g.ex);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double a = exd[2 * i], b = exd[2 * i + 1];
            big = a > big ? a : big;
            big = b > big ? b : big;
        }
    }
    grp_sync();
    double p2 = 1.0;
    if (big > 0.0 && big < 1.7976931348623157e308) {
        const int e = (int)((x_bits(big) >> 52) & 0x7ff) - 1023;
        int k = -e;
        k = k > 1000 ? 1000 : (k < -1000 ? -1000 : k);
        p2 = x_from_bits((int64_t)(k + 1023) << 52);
    }
    const x87 xp = x_from(pre * p2), xe = x_from(epow * p2);
#pragma unroll
    for (int j = 0; j < 3; ++j) M[3 * r + j] = c_add(c_scale(xp, hs[j]), c_scale(xe, hn[j]));      // fr.py:386, 394-395
    grp_sync();
    if (tick) tick[0] = clock64();
    // cardano_residual
    const x87 two = x_from(2.0), three = x_from(3.0), nine = x_from(9.0), n27 = x_from(27.0);
    const cx87 tr = c_add(c_add(M[0], M[4]), M[8]);
    // tr H^2: lane r forms (H^2)_rr
    {
        cx87 sacc = c_mul(M[3 * r + 0], M[0 + r]);
        sacc = c_add(sacc, c_mul(M[3 * r + 1], M[3 + r]));
        sacc = c_add(sacc, c_mul(M[3 * r + 2], M[6 + r]));
        g.ex[r] = sacc;
    }
    grp_sync();
    const cx87 tr2 = c_add(c_add(g.ex[0], g.ex[1]), g.ex[2]);
    grp_sync();
    // det (fr.py:77-79): lane r forms h[r][0] * (h[p][1] h[q][2] - h[q][1] h[p][2]), (p, q) the two other rows in order
    {
        const int p = r == 0 ? 1 : 0, q = r == 2 ? 1 : 2;
        g.ex[r] = c_mul(M[3 * r], c_sub(c_mul(M[3 * p + 1], M[3 * q + 2]), c_mul(M[3 * q + 1], M[3 * p + 2])));
    }
    grp_sync();
    const cx87 det = c_add(c_sub(g.ex[0], g.ex[1]), g.ex[2]);
    grp_sync();
    // products the eigenvectors share: lane 0 h10 h02, lane 1 h21 h10, lane 2 h12 h20
    {
        const int i0 = r == 0 ? 3 : (r == 1 ? 7 : 5), i1 = r == 0 ? 2 : (r == 1 ? 3 : 6);
        g.ex[r] = c_mul(M[i0], M[i1]);
    }
    grp_sync();
    const cx87 h10h02 = g.ex[0], h21h10 = g.ex[1], h12h20 = g.ex[2];
    if (tick) tick[1] = clock64();
    const cx87 a = c_neg(tr);                                                           // fr.py:204
    const cx87 a2 = c_mul(tr, tr);                                                      // = a a, bit for bit: (-x)(-y) is x y
    const cx87 b = c_scale(GFX_X87_HALF, c_sub(a2, tr2));                               // fr.py:205
    const cx87 c = c_neg(det);                                                          // fr.py:206
    const cx87 Q = c_scale(GFX_X87_NINTH, c_sub(a2, c_scale(three, b)));                // fr.py:208
    const cx87 R = c_scale(GFX_X87_54TH,
                           c_add(c_sub(c_scale(two, c_mul(a, a2)), c_mul(c_scale(nine, a), b)), c_scale(n27, c)));   // fr.py:209
    // the two complex square roots of the bin, sqrt(Q^3) (fr.py:210) and sqrt(Q) (fr.py:212-214), in ONE pass: lanes 0 and 1 take
    // the first, lane 2 the second, and they meet in the exchange slot
    const cx87 Q3 = c_mul(Q, c_mul(Q, Q));
    g.ex[r] = c_sqrt_pos(r == 2 ? Q : Q3);
    grp_sync();
    const cx87 sqQ3 = g.ex[0], sq = g.ex[2];
    grp_sync();
    const cx87 theta = c_acos_near_real(c_div(R, sqQ3));                                // fr.py:210
    if (tick) tick[2] = clock64();
    const cx87 m2sq = c_scale(x_neg(two), sq);
    const cx87 third_a = c_scale(GFX_X87_THIRD, a);
    const x87 pi = {3.141592653589793, 1.22514845490862e-16};                           // np.arccos(np.float128(-1)), fr.py:24
    const x87 twopi = x_mul(two, pi);
    // eigenvalue r: theta, theta - 2 pi, theta + 2 pi  (fr.py:212-214)
    x87 are = theta.re;
    if (r == 1) are = x_sub(theta.re, twopi);
    if (r == 2) are = x_add(theta.re, twopi);
    const cx87 E = c_sub(c_mul(m2sq, c_cos_near_real(c_div_real(c_make(are, theta.im), three))), third_a);
    if (tick) tick[3] = clock64();
    // eigenvector r (fr.py:216-236)
    const cx87 A = c_sub(c_mul(M[5], c_sub(M[0], E)), h10h02);
    const cx87 B = c_sub(c_mul(M[6], c_sub(M[4], E)), h21h10);
    const cx87 C = c_sub(c_mul(M[3], c_sub(M[8], E)), h12h20);
    const cx87 AB = c_mul(A, B), AC = c_mul(A, C), BC = c_mul(B, C);
    const x87 ab = c_abs(AB), ac = c_abs(AC), bc = c_abs(BC);
    const x87 N = x_sqrt(x_add(x_add(x_mul(ab, ab), x_mul(ac, ac)), x_mul(bc, bc)));   // fr.py:228-230
    // fr.py:232-236: complex / real is x * (1 / d) in numpy (c_div_real): the reciprocal once for the three components
    const x87 rn = x_div(x_from(1.0), N);
    const cx87 cbc = c_mul(c_conj(B), C);
    const cx87 x0 = c_make(x_mul(cbc.re, rn), x_mul(cbc.im, rn));
    const cx87 x1 = c_make(x_mul(AC.re, rn), x_mul(AC.im, rn));
    const cx87 x2 = c_make(x_mul(AB.re, rn), x_mul(AB.im, rn));
    if (tick) tick[4] = clock64();
    grp_sync();                                                                         // every lane is done with H
    M[0 + r] = x0; M[3 + r] = x1; M[6 + r] = x2;                                        // column r of X
    grp_sync();
    // f = |X X^+| (fr.py:489): lane 0 -> f00, f01; lane 1 -> f02, f11; lane 2 -> f12, f22
    {
        x87* exx = reinterpret_cast<x87*>(g.ex);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = 2 * r + e;                                                  // 0..5 = (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
            const int i = idx < 3 ? 0 : (idx < 5 ? 1 : 2);
            const int j = idx < 3 ? idx : (idx < 5 ? idx - 2 : 2);
            cx87 sacc = c_mul(M[3 * i + 0], c_conj(M[3 * j + 0]));
            sacc = c_add(sacc, c_mul(M[3 * i + 1], c_conj(M[3 * j + 1])));
            sacc = c_add(sacc, c_mul(M[3 * i + 2], c_conj(M[3 * j + 2])));
            exx[idx] = c_abs(sacc);
        }
    }
    grp_sync();
    double res;
    {
        const x87* exx = reinterpret_cast<const x87*>(g.ex);
        const x87 f00 = exx[0], f01 = exx[1], f02 = exx[2], f11 = exx[3], f12 = exx[4], f22 = exx[5];
        const x87 trf = x_add(x_add(f00, f11), f22);
        const x87 sum = x_add(x_add(x_add(x_add(f00, f01), x_add(f02, f01)), x_add(x_add(f11, f12), x_add(f02, f12))), f22);
        const double rt = fabs(x_to_double(x_sub(trf, three))), rs = fabs(x_to_double(x_sub(sum, three)));
        res = rt > rs ? rt : rs;
        if (!(res == res) || !(rt == rt) || !(rs == rs)) res = INFINITY;
    }
    grp_sync();                                                                         // M and ex are free again
    return res;
}

// ---- nine lanes per walker ---------------------------------------------------------------------------------------------
// The same chain once more, for the one place where a walker's LATENCY is what the caller waits for (the device sampler's
// settle step: a half-step stands still until the slowest parked proposal has its verdict).  Lane l = 3 r + c of a group owns
// ENTRY (r, c) of every 3x3 matrix: the four angle functions, the eighteen entries of the two sandwiches, the nine products
// of tr H^2, the nine of the determinant and the eigenvectors' shared terms, the three components of each eigenvector and
// the six entries of |X X^+| are each one lane's work instead of a row's; rows / columns meet in the group's LDS slot.
// Still every operation is the serial chain's, on the same operands in the same order: the residual is the serial chain's bit
// for bit (tests/test_gpu_unitarity_r3.py).  Seven groups per wave; not used for the bulk path, where the three-lane form
// wastes fewer lanes on the scalar part.
constexpr int G9 = 9;
constexpr int G9_PER_WAVE = 64 / G9;                 // 7 (lane 63 idles)
constexpr int G9_DOUBLES = 4 * 36 + 50;              // four 3x3 complex arrays + a small exchange area; 1552 B, LDS bank step 194
struct Grp9 {
    cx87 *A0, *A1, *A2, *A3;                         // [9] each
    double* sm;                                      // [48] small exchanges
    int r, c, l;
};

// entry (r, c) of angles_to_u (gf_x87.hpp angles_to_u, fr.py:116-162)
__device__ __forceinline__ cx87 g9_angles_to_u(const Grp9& g, double ang0, double ang1, double ang2, double ang3)
{
    const int r = g.r, c = g.c;
    // lanes (0,0), (1,0), (2,0) evaluate theta12, theta13, theta23; lane (0,1) the phase; the others repeat lane (0,0)'s work
    const int which = c == 0 ? r : (g.l == 1 ? 3 : 0);
    // (four scalars, not an array: the compiler turned the selects over ang[] into an indexed load from a stack copy -- scratch)
    const double a0 = which == 1 ? ang1 : (which == 2 ? ang2 : ang0);
    x87 a = x_sqrt(x_from(a0));                                       // fr.py:141,145-147
    if (which == 1) a = x_sqrt(a);
    const dd da = as_dd(a);
    const bool small = da.hi <= 0.72;
    const dd arg = small ? da : dd_cofunc(da);
    const dd as = dd_asin_small(arg);
    const bool complement = (which == 1) == small;
    x87 t = round64(complement ? dd_sub(dd_pio2(), as) : as);
    if (which == 3) t = x_from(ang3);
    x87 sn, cs;
    x_sincos(t, sn, cs);                                              // fr.py:149-154; exp(+-i dcp) = (cos, +-sin)
    cx87* ex = g.A0;
    if (c == 0 || g.l == 1) ex[which] = c_make(sn, cs);
    grp_sync();
    const cx87 e12 = ex[0], e13 = ex[1], e23 = ex[2], ed = ex[3];
    grp_sync();
    const x87 s12 = e12.re, c12 = e12.im, s13 = e13.re, c13 = e13.im, s23 = e23.re, c23 = e23.im, sd = ed.re, cd = ed.im;
    const cx87 em = c_make(cd, x_neg(sd)), ep = c_make(cd, sd);
    const cx87 s13em = c_scale(s13, em);                              // p2[0][2]
    const cx87 ms13ep = c_scale(x_neg(s13), ep);                      // p2[2][0]
    const x87 zero = x_from(0.0);
    // T = p1 . p2, row r (as grp_angles_to_u_row)
    const x87 fa = r == 1 ? s23 : c23, fb = r == 1 ? c23 : x_neg(s23);
    cx87 T0 = c_scale(fa, ms13ep), T1 = c_make(fb, zero), T2 = c_make(x_mul(fa, c13), zero);
    if (r == 0) { T0 = c_make(c13, zero); T1 = c_zero(); T2 = s13em; }
    // u = T . p3, entry c
    const x87 ga = c == 0 ? c12 : s12, gb = c == 0 ? x_neg(s12) : c12;
    const cx87 v = c_add(c_scale(ga, T0), c_scale(gb, T1));
    return c == 2 ? T2 : v;
}

__device__ __forceinline__ cx87 g9_load(const double* hi, const double* lo, int l)
{
    const x87 re = {hi[2 * l], lo[2 * l]}, im = {hi[2 * l + 1], lo[2 * l + 1]};
    return c_make(re, im);
}

// entry (r, c) of U diag(0, w1, w2) U^+ (gf_x87.hpp sandwich)
__device__ __forceinline__ cx87 g9_sandwich(const Grp9& g, cx87 u, double w1, double w2)
{
    const x87 xw1 = x_from(w1), xw2 = x_from(w2);
    cx87* U = g.A0;
    U[g.l] = u;
    grp_sync();
    const cx87 ui1 = U[3 * g.r + 1], ui2 = U[3 * g.r + 2], uj1 = U[3 * g.c + 1], uj2 = U[3 * g.c + 2];
    grp_sync();
    const cx87 t1 = c_scale(xw1, c_conj(uj1));                        // (diag . U^+)[1][j]
    const cx87 t2 = c_scale(xw2, c_conj(uj2));
    return c_add(c_mul(ui1, t1), c_mul(ui2, t2));
}

__device__ __forceinline__ void g9_walker_terms(const Grp9& g, const GfCommon& c, const GfBsm& tb, const double* __restrict__ theta,
                                                int layout, int64_t n, int64_t i, cx87& hs, cx87& hn, int stride = 0)
{
    const int ndim = stride ? stride : c.ndim;
    cx87 u;
    if (c.idx_sm[0] >= 0) {
        u = g9_angles_to_u(g, row_value(theta, layout, n, ndim, i, c.idx_sm[0]), row_value(theta, layout, n, ndim, i, c.idx_sm[1]),
                           row_value(theta, layout, n, ndim, i, c.idx_sm[2]), row_value(theta, layout, n, ndim, i, c.idx_sm[3]));
    } else {
        u = g9_load(tb.smu_hi, tb.smu_lo, g.l);
    }
    const double m21 = c.idx_mass[0] >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_mass[0]) : c.mass_fixed[0];
    const double m3x = c.idx_mass[1] >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_mass[1]) : c.mass_fixed[1];
    hs = g9_sandwich(g, u, m21, m3x);
    if (tb.texture == TEX_NONE && c.idx_mm[0] >= 0) {
        u = g9_angles_to_u(g, row_value(theta, layout, n, ndim, i, c.idx_mm[0]), row_value(theta, layout, n, ndim, i, c.idx_mm[1]),
                           row_value(theta, layout, n, ndim, i, c.idx_mm[2]), row_value(theta, layout, n, ndim, i, c.idx_mm[3]));
    } else {
        u = g9_load(tb.npu_hi, tb.npu_lo, g.l);
    }
    const double ll = c.idx_scale >= 0 ? row_value(theta, layout, n, ndim, i, c.idx_scale) : c.scale_fixed;
    const double sc2 = cr_pow10(ll);
    const double sc1 = sc2 / 100.0;
    hn = g9_sandwich(g, u, sc1, sc2);
}

// one energy bin on the nine lanes; every lane returns the same residual
__device__ __forceinline__ double g9_bin_residual(const Grp9& g, cx87 hs, cx87 hn, double pre, double epow)
{
    const int r = g.r, c = g.c, l = g.l;
    cx87 *M = g.A0, *P = g.A1, *S = g.A2, *X = g.A3;
    double* sm = g.sm;
    if (r == c) { sm[2 * r] = fabs(pre * hs.re.hi); sm[2 * r + 1] = fabs(epow * hn.re.hi); }
    grp_sync();
    double big = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double a = sm[i]; big = a > big ? a : big; }
    grp_sync();
    double p2 = 1.0;
    if (big > 0.0 && big < 1.7976931348623157e308) {
        const int e = (int)((x_bits(big) >> 52) & 0x7ff) - 1023;
        int k = -e;
        k = k > 1000 ? 1000 : (k < -1000 ? -1000 : k);
        p2 = x_from_bits((int64_t)(k + 1023) << 52);
    }
    const x87 xp = x_from(pre * p2), xe = x_from(epow * p2);
    M[l] = c_add(c_scale(xp, hs), c_scale(xe, hn));                                      // fr.py:386, 394-395
    grp_sync();
    const x87 two = x_from(2.0), three = x_from(3.0), nine = x_from(9.0), n27 = x_from(27.0);
    const cx87 tr = c_add(c_add(M[0], M[4]), M[8]);
    // the nine products of tr H^2 -- entry (r, c) forms h[r][c] h[c][r] -- and, in a second pass, the six inner products of the
    // determinant (lanes 0-2: h[p][1] h[q][2], lanes 3-5: h[q][1] h[p][2], (p, q) the rows other than l mod 3 in order) and the
    // three products the eigenvectors share (lane 6: h10 h02, 7: h21 h10, 8: h12 h20)
    P[l] = c_mul(M[3 * r + c], M[3 * c + r]);
    {
        const int rr = l < 3 ? l : (l < 6 ? l - 3 : 0);
        const int p = rr == 0 ? 1 : 0, q = rr == 2 ? 1 : 2;
        int i0 = l < 3 ? 3 * p + 1 : 3 * q + 1, i1 = l < 3 ? 3 * q + 2 : 3 * p + 2;
        if (l == 6) { i0 = 3; i1 = 2; }
        if (l == 7) { i0 = 7; i1 = 3; }
        if (l == 8) { i0 = 5; i1 = 6; }
        S[l] = c_mul(M[i0], M[i1]);
    }
    grp_sync();
    const cx87 h10h02 = S[6], h21h10 = S[7], h12h20 = S[8];
    {
        // row r of tr H^2 and term r of the determinant (the three lanes of a row do the same; one of them hands it on)
        const cx87 srow = c_add(c_add(P[3 * r], P[3 * r + 1]), P[3 * r + 2]);
        const cx87 drow = c_mul(M[3 * r], c_sub(S[r], S[3 + r]));
        cx87* ex = reinterpret_cast<cx87*>(sm);
        if (c == 0) { ex[r] = srow; ex[3 + r] = drow; }
    }
    grp_sync();
    cx87 tr2, det;
    {
        const cx87* ex = reinterpret_cast<const cx87*>(sm);
        tr2 = c_add(c_add(ex[0], ex[1]), ex[2]);
        det = c_add(c_sub(ex[3], ex[4]), ex[5]);                                          // fr.py:77-79
    }
    grp_sync();
    const cx87 a = c_neg(tr);                                                           // fr.py:204
    const cx87 a2 = c_mul(tr, tr);                                                      // = a a, bit for bit
    const cx87 b = c_scale(GFX_X87_HALF, c_sub(a2, tr2));                               // fr.py:205
    const cx87 cc = c_neg(det);                                                         // fr.py:206
    const cx87 Q = c_scale(GFX_X87_NINTH, c_sub(a2, c_scale(three, b)));                // fr.py:208
    const cx87 R = c_scale(GFX_X87_54TH,
                           c_add(c_sub(c_scale(two, c_mul(a, a2)), c_mul(c_scale(nine, a), b)), c_scale(n27, cc)));   // fr.py:209
    const cx87 Q3 = c_mul(Q, c_mul(Q, Q));
    {
        cx87* ex = reinterpret_cast<cx87*>(sm);
        const cx87 root = c_sqrt_pos(r == 2 ? Q : Q3);                                   // both square roots of the bin in one pass
        if (c == 0) ex[r] = root;
    }
    grp_sync();
    cx87 sqQ3, sq;
    { const cx87* ex = reinterpret_cast<const cx87*>(sm); sqQ3 = ex[0]; sq = ex[2]; }
    grp_sync();
    const cx87 theta = c_acos_near_real(c_div(R, sqQ3));                                // fr.py:210
    const cx87 m2sq = c_scale(x_neg(two), sq);
    const cx87 third_a = c_scale(GFX_X87_THIRD, a);
    const x87 pi = {3.141592653589793, 1.22514845490862e-16};
    const x87 twopi = x_mul(two, pi);
    // eigenvalue k = r (fr.py:212-214); the three lanes of a row are the components A, B, C of eigenvector k (fr.py:216-226)
    x87 are = theta.re;
    if (r == 1) are = x_sub(theta.re, twopi);
    if (r == 2) are = x_add(theta.re, twopi);
    const cx87 E = c_sub(c_mul(m2sq, c_cos_near_real(c_div_real(c_make(are, theta.im), three))), third_a);
    {
        const cx87 mi = c == 0 ? M[5] : (c == 1 ? M[6] : M[3]);
        const cx87 mjj = c == 0 ? M[0] : (c == 1 ? M[4] : M[8]);
        const cx87 sp = c == 0 ? h10h02 : (c == 1 ? h21h10 : h12h20);
        P[l] = c_sub(c_mul(mi, c_sub(mjj, E)), sp);                                     // A | B | C of eigenvector r
    }
    grp_sync();
    const cx87 A = P[3 * r], B = P[3 * r + 1], C = P[3 * r + 2];
    // lane c = 0: AB, 1: AC, 2: BC (fr.py:228-230), then the component it owns: x2 = AB / N, x1 = AC / N, x0 = conj(B) C / N
    const cx87 prod = c_mul(c == 2 ? B : A, c == 0 ? B : C);
    const cx87 cbc = c_mul(c_conj(B), C);
    {
        x87* exx = reinterpret_cast<x87*>(sm);
        exx[l] = c_abs(prod);
    }
    grp_sync();
    x87 N;
    {
        const x87* exx = reinterpret_cast<const x87*>(sm);
        const x87 ab = exx[3 * r], ac = exx[3 * r + 1], bc = exx[3 * r + 2];
        N = x_sqrt(x_add(x_add(x_mul(ab, ab), x_mul(ac, ac)), x_mul(bc, bc)));
    }
    const x87 rn = x_div(x_from(1.0), N);                                               // fr.py:232-236: x * (1 / d)
    const cx87 comp = c == 2 ? cbc : prod;
    X[3 * (2 - c) + r] = c_make(x_mul(comp.re, rn), x_mul(comp.im, rn));                // column r of X
    grp_sync();
    // f = |X X^+| (fr.py:489): lanes 0..5 take (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    {
        const int idx = l < 6 ? l : 5;
        const int i = idx < 3 ? 0 : (idx < 5 ? 1 : 2);
        const int j = idx < 3 ? idx : (idx < 5 ? idx - 2 : 2);
        cx87 sacc = c_mul(X[3 * i + 0], c_conj(X[3 * j + 0]));
        sacc = c_add(sacc, c_mul(X[3 * i + 1], c_conj(X[3 * j + 1])));
        sacc = c_add(sacc, c_mul(X[3 * i + 2], c_conj(X[3 * j + 2])));
        x87* exx = reinterpret_cast<x87*>(sm) + 12;                                     // (the norms above may still be being read)
        if (l < 6) exx[idx] = c_abs(sacc);
    }
    grp_sync();
    double res;
    {
        const x87* exx = reinterpret_cast<const x87*>(sm) + 12;
        const x87 f00 = exx[0], f01 = exx[1], f02 = exx[2], f11 = exx[3], f12 = exx[4], f22 = exx[5];
        const x87 trf = x_add(x_add(f00, f11), f22);
        const x87 sum = x_add(x_add(x_add(x_add(f00, f01), x_add(f02, f01)), x_add(x_add(f11, f12), x_add(f02, f12))), f22);
        const double rt = fabs(x_to_double(x_sub(trf, three))), rs = fabs(x_to_double(x_sub(sum, three)));
        res = rt > rs ? rt : rs;
        if (!(res == res) || !(rt == rt) || !(rs == rs)) res = INFINITY;
    }
    grp_sync();
    return res;
}

// The two teams behind one face, so that the kernels below are written once.
struct Team3 {
    static constexpr int LANES = GRP, PER_WAVE = GRP_PER_WAVE, DOUBLES = GRP_DOUBLES;
    Grp g;
    cx87 hs[3], hn[3];
    __device__ __forceinline__ void init(double* base, int lane_in_group) { g.M = reinterpret_cast<cx87*>(base); g.ex = reinterpret_cast<cx87*>(base + 36); g.r = lane_in_group; }
    __device__ __forceinline__ bool leader() const { return g.r == 0; }
    __device__ __forceinline__ void terms(const GfCommon& c, const GfBsm& tb, const double* theta, int layout, int64_t n, int64_t i, int stride)
    { grp_walker_terms(g, c, tb, theta, layout, n, i, hs, hn, stride); }
    __device__ __forceinline__ double bin(double pre, double epow, long long* tick = nullptr) { return grp_bin_residual(g, hs, hn, pre, epow, tick); }
};
struct Team9 {
    static constexpr int LANES = G9, PER_WAVE = G9_PER_WAVE, DOUBLES = G9_DOUBLES;
    Grp9 g;
    cx87 hs, hn;
    __device__ __forceinline__ void init(double* base, int lane_in_group)
    {
        g.A0 = reinterpret_cast<cx87*>(base); g.A1 = g.A0 + 9; g.A2 = g.A1 + 9; g.A3 = g.A2 + 9; g.sm = base + 144;
        g.l = lane_in_group; g.r = lane_in_group / 3; g.c = lane_in_group - 3 * g.r;
    }
    __device__ __forceinline__ bool leader() const { return g.l == 0; }
    __device__ __forceinline__ void terms(const GfCommon& c, const GfBsm& tb, const double* theta, int layout, int64_t n, int64_t i, int stride)
    { g9_walker_terms(g, c, tb, theta, layout, n, i, hs, hn, stride); }
    __device__ __forceinline__ double bin(double pre, double epow, long long* = nullptr) { return g9_bin_residual(g, hs, hn, pre, epow); }
};

// Fan-out of a short queue.  A walker's bins are evaluated one after the other by one group -- right for throughput, but a
// queue with fewer walkers than the grid has groups leaves most of the GPU idle behind the critical path of the walker with
// the most bins (nine bins: ~0.4 ms).  So with `count` walkers and `groups` groups in the grid every walker is cut into
// F = min(GF_UNI_MAX_FANOUT, groups / count) parts: part j takes the walker's j-th, (j + F)-th, ... undecided bin, counted from
// the highest energy, and builds the walker's terms itself (redundant work on otherwise idle lanes).  A long queue has F = 1.
#ifndef GF_UNI_MAX_FANOUT
#define GF_UNI_MAX_FANOUT 20
#endif
__device__ __forceinline__ unsigned int uni_fanout(unsigned int count, unsigned int groups)
{
    if (count == 0u) return 1u;
    unsigned int f = groups / count;
    f = f < 1u ? 1u : f;
    return f > (unsigned int)GF_UNI_MAX_FANOUT ? (unsigned int)GF_UNI_MAX_FANOUT : f;
}
// the bins of `mask` whose rank from the top is part, part + fan, ...
__device__ __forceinline__ unsigned long long uni_part_mask(unsigned long long mask, unsigned int part, unsigned int fan)
{
    if (fan <= 1u) return mask;
    unsigned long long out = 0ull;
    unsigned int ord = 0;
    for (unsigned long long rest = mask; rest != 0ull; ++ord) {
        const int k = 63 - __clzll((long long)rest);
        rest &= ~(1ull << k);
        if (ord % fan == part) out |= 1ull << k;
    }
    return out;
}

// Three lanes = one walker at a time.  A group fetches a walker from the queue (one atomic per wave and round, shared out by
// rank among the groups that need one), builds its Hamiltonian terms, then takes its undecided bins from the highest energy
// down -- one bin per round of the wave -- until one fails (the walker is non-unitary: fr.py:398-399 raises on the first
// failing energy) or none is left; then it fetches the next walker.  Every wave leaves when the queue is exhausted and all its
// groups have finished their walker: the exit condition is reached whatever the other waves do.
template <class Team>
__global__ __launch_bounds__(UNI_BLOCK, GF_UNI_WAVES) void k_uni_resolve(const GfCommon* __restrict__ cp, const GfBsm* __restrict__ tbp,
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
        if (fan > 1u && mask != 0ull && status[wi] == ST_NON_UNITARY) mask = 0ull;    // another part of this walker has failed already
        if (mask != 0ull) {
            const int k = 63 - __clzll((long long)mask);                // the highest undecided energy first: the likeliest to fail
            mask &= ~(1ull << k);
            const double res = tm.bin(tbp->inv2e[k], tbp->epow[k]);
            if (!(res < 1e-7)) {                                        // fr.py:493-494 (NaN raises too)
                if (lead) {
                    status[wi] = ST_NON_UNITARY;
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
    for (;;) {
        const bool need = !exhausted && !mine;
        const unsigned long long nb = __ballot(need && lead);
        if (nb != 0ull) {
            unsigned int base = 0;
            const int leader = __ffsll((long long)nb) - 1;
            if (lane == leader) base = atomicAdd(&uq->head, (unsigned int)__popcll(nb));
            base = (unsigned int)__shfl((int)base, leader);
            if (need) {
                const unsigned int idx = base + (unsigned int)__popcll(nb & ((1ull << (grp * Team::LANES)) - 1ull));
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
                    // (gf_sampler.hip stretch_body, after proposal_lnprob)
                    if (failed) atomicOr(&s.ctl[2 * t + 1], 1u);
                    __threadfence();
                    const unsigned int before = atomicAdd(&s.ctl[2 * t], 1u);
                    if (before == parts - 1u) {
                        __threadfence();
                        const bool bad = atomicOr(&s.ctl[2 * t + 1], 0u) != 0u;
                        s.ctl[2 * t] = 0u;                              // zero between uses
                        s.ctl[2 * t + 1] = 0u;
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
