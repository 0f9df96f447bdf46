// gf_unitarity_teams.hpp -- the reference's unitarity chain (golemflavor/fr.py:116-162, 170-237, 461-499) in emulated x87
// arithmetic, distributed over THREE or NINE lanes per walker (bit for bit the serial chain of gf_x87.hpp), behind one face
// (`Team3`, `Team9`), plus the fan-out helpers of a short queue.  Included by gf_unitarity.hip (the arbitration kernels of the bulk
// path and of the grid sampler) and by gf_sampler.hip (the per-chain persistent sampler settles its own parked proposals inside
// its workgroup).  Everything lives in the including translation unit's anonymous namespace.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf_consts.h"
// everything of the emulated chain inline: the out-of-line division / square root / sine series of gf_x87.hpp's default cost
// k_uni_resolve a call frame (168 B of scratch per lane) and 5 % of its time (profiles/r03/ab_arbitration.txt)
#ifndef GFX87_INLINE_ALL
#define GFX87_INLINE_ALL
#endif
#include "gf_x87.hpp"

namespace {
using namespace gfx87;

constexpr int UT_TEX_NONE = 4;
constexpr int UT_NON_UNITARY = 2;
constexpr int UNI_BLOCK = 128;
#ifndef GF_UNI_WAVES
#define GF_UNI_WAVES 2                                // waves per SIMD k_uni_resolve is compiled for (~245 VGPRs)
#endif
#ifndef GF_UNI_RESOLVE_WAVES
#define GF_UNI_RESOLVE_WAVES GF_UNI_WAVES             // the bulk arbitration kernel alone (throughput), apart from the settle step (latency)
#endif
#ifndef GF_UNI_BLOCKS_PER_CU
#define GF_UNI_BLOCKS_PER_CU (GF_UNI_RESOLVE_WAVES * 4 * 64 / UNI_BLOCK)    // what is resident at once: later blocks would find the queue empty
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
    if (tb.texture == UT_TEX_NONE && c.idx_mm[0] >= 0) {                   // fr.py:378, 390
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
    if (tb.texture == UT_TEX_NONE && c.idx_mm[0] >= 0) {                   // fr.py:378, 390
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
    const cx87 a2 = c_sqr(tr);                                                          // = a a, bit for bit: (-x)(-y) is x y; a square: gf_x87.hpp c_sqr
    const cx87 b = c_times2(c_sub(a2, tr2), 0.5);                                       // fr.py:205 (halving is exact)
    const cx87 c = c_neg(det);                                                          // fr.py:206
    const cx87 Q = c_scale(GFX_X87_NINTH, c_sub(a2, c_scale(three, b)));                // fr.py:208
    const cx87 R = c_scale(GFX_X87_54TH,
                           c_add(c_sub(c_times2(c_mul(a, a2), 2.0), c_mul(c_scale(nine, a), b)), c_scale(n27, c)));   // fr.py:209
    // the two complex square roots of the bin, sqrt(Q^3) (fr.py:210) and sqrt(Q) (fr.py:212-214), in ONE pass: lanes 0 and 1 take
    // the first, lane 2 the second, and they meet in the exchange slot
    const cx87 Q3 = c_mul(Q, c_sqr(Q));
    g.ex[r] = c_sqrt_pos(r == 2 ? Q : Q3);
    grp_sync();
    const cx87 sqQ3 = g.ex[0], sq = g.ex[2];
    grp_sync();
    const cx87 theta = c_acos_near_real(c_div(R, sqQ3));                                // fr.py:210
    if (tick) tick[2] = clock64();
    const cx87 m2sq = c_times2(sq, -2.0);
    const cx87 third_a = c_scale(GFX_X87_THIRD, a);
    const x87 pi = {3.141592653589793, 1.22514845490862e-16};                           // np.arccos(np.float128(-1)), fr.py:24
    const x87 twopi = x_mul(two, pi);
    // eigenvalue r: theta, theta - 2 pi, theta + 2 pi  (fr.py:212-214)
    x87 are = theta.re;
    if (r == 1) are = x_sub(theta.re, twopi);
    if (r == 2) are = x_add(theta.re, twopi);
    const cx87 E = c_sub(c_mul(m2sq, c_cos_near_real(c_scale(GFX_X87_THIRD, c_make(are, theta.im)))), third_a);
    if (tick) tick[3] = clock64();
    // eigenvector r (fr.py:216-236)
    const cx87 A = c_sub(c_mul(M[5], c_sub(M[0], E)), h10h02);
    const cx87 B = c_sub(c_mul(M[6], c_sub(M[4], E)), h21h10);
    const cx87 C = c_sub(c_mul(M[3], c_sub(M[8], E)), h12h20);
    // B C and conj(B) C (fr.py:232) are made of the SAME four products: conj(B) = (Br, -Bi), and (-Bi) Ci = -(Bi Ci) exactly
    const x87 brcr = x_mul(B.re, C.re), bici = x_mul(B.im, C.im), brci = x_mul(B.re, C.im), bicr = x_mul(B.im, C.re);
    const cx87 AB = c_mul(A, B), AC = c_mul(A, C), BC = c_make(x_sub(brcr, bici), x_add(brci, bicr));
    const x87 ab = c_abs(AB), ac = c_abs(AC), bc = c_abs(BC);
    const x87 N = x_sqrt(x_add(x_add(x_mul(ab, ab), x_mul(ac, ac)), x_mul(bc, bc)));   // fr.py:228-230
    // fr.py:232-236: complex / real is x * (1 / d) in numpy (c_div_real): the reciprocal once for the three components
    const x87 rn = x_div(x_from(1.0), N);
    const cx87 cbc = c_make(x_add(brcr, bici), x_sub(brci, bicr));                      // c_mul(c_conj(B), C)
    const cx87 x0 = c_make(x_mul(cbc.re, rn), x_mul(cbc.im, rn));
    const cx87 x1 = c_make(x_mul(AC.re, rn), x_mul(AC.im, rn));
    const cx87 x2 = c_make(x_mul(AB.re, rn), x_mul(AB.im, rn));
    if (tick) tick[4] = clock64();
    grp_sync();                                                                         // every lane is done with H
    M[0 + r] = x0; M[3 + r] = x1; M[6 + r] = x2;                                        // column r of X
    grp_sync();
    // f = |X X^+| (fr.py:489): lane r -> the diagonal entry f_rr first (all three lanes together: a sum of squares, its modulus
    // itself -- gf_x87.hpp c_norm2), then one entry above the diagonal: lane 0 -> f01, lane 1 -> f02, lane 2 -> f12
    {
        x87* exx = reinterpret_cast<x87*>(g.ex);                                        // 0..5 = (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
        {
            x87 d = c_norm2(M[3 * r + 0]);
            d = x_add(d, c_norm2(M[3 * r + 1]));
            d = x_add(d, c_norm2(M[3 * r + 2]));
            exx[r == 0 ? 0 : (r == 1 ? 3 : 5)] = d;
        }
        {
            const int i = r == 2 ? 1 : 0, j = r == 0 ? 1 : 2;
            cx87 sacc = c_mul(M[3 * i + 0], c_conj(M[3 * j + 0]));
            sacc = c_add(sacc, c_mul(M[3 * i + 1], c_conj(M[3 * j + 1])));
            sacc = c_add(sacc, c_mul(M[3 * i + 2], c_conj(M[3 * j + 2])));
            exx[r == 0 ? 1 : (r == 1 ? 2 : 4)] = c_abs(sacc);
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
    if (tb.texture == UT_TEX_NONE && c.idx_mm[0] >= 0) {
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
    const cx87 a2 = c_sqr(tr);                                                          // = a a, bit for bit; a square: gf_x87.hpp c_sqr
    const cx87 b = c_times2(c_sub(a2, tr2), 0.5);                                       // fr.py:205 (halving is exact)
    const cx87 cc = c_neg(det);                                                         // fr.py:206
    const cx87 Q = c_scale(GFX_X87_NINTH, c_sub(a2, c_scale(three, b)));                // fr.py:208
    const cx87 R = c_scale(GFX_X87_54TH,
                           c_add(c_sub(c_times2(c_mul(a, a2), 2.0), c_mul(c_scale(nine, a), b)), c_scale(n27, cc)));   // fr.py:209
    const cx87 Q3 = c_mul(Q, c_sqr(Q));
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
    const cx87 m2sq = c_times2(sq, -2.0);
    const cx87 third_a = c_scale(GFX_X87_THIRD, a);
    const x87 pi = {3.141592653589793, 1.22514845490862e-16};
    const x87 twopi = x_mul(two, pi);
    // eigenvalue k = r (fr.py:212-214); the three lanes of a row are the components A, B, C of eigenvector k (fr.py:216-226)
    x87 are = theta.re;
    if (r == 1) are = x_sub(theta.re, twopi);
    if (r == 2) are = x_add(theta.re, twopi);
    const cx87 E = c_sub(c_mul(m2sq, c_cos_near_real(c_scale(GFX_X87_THIRD, c_make(are, theta.im)))), third_a);
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
}  // namespace
