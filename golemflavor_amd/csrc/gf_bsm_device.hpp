// gf_bsm_device.hpp -- device functions of the BSM (texture) branch, shared by gf_bsm.hip (batch
// evaluation) and gf_sampler.hip (device-resident stretch move).  See gf_bsm.hip for the numerics notes.
#pragma once
#include "gf_device.hpp"

// Floating-point contraction only where the source writes one expression (a * b + c), not across statements: with the
// compiler's default (`fast`) the back end fuses or not depending on how often a product is used elsewhere, and the
// <CHECK_UNI = true> / <false> instances, the lanes-per-walker variants and the sampler's copy of this code would round
// differently in the last bit.  They are required to agree bit for bit (tests).  Restored at the end of the file.
#pragma clang fp contract(on)

namespace gfdev {

constexpr int TEX_NONE = 4;
// The unitarity residual of the reference (fr.py:489-494, threshold 1e-7) is x87 rounding noise, amplified by the
// cancellations of the eigenvector formula.  Three tiers decide a (walker, energy bin) pair:
//  1. the weight `a` of the SM term in the trace-normalised Hamiltonian H' = a S' + t N'.  The amplification grows like
//     1/a (the SM term is what lifts the zero eigenvalue of the NP term): over 30 000 pairs of every operator dimension
//     and texture incl. random NP angles the 80-bit residual never exceeds 1.3e-19 / a (tools/uni_weight_bound.py,
//     profiles/r02/uni_weight_bound.txt).  a >= uni_a_ok (2e-11; residual < 7e-9): unitary, nothing is evaluated --
//     the whole low-scale bulk of a posterior costs nothing extra;
//  2. else the same eigenvector form in fp64, whose residual `rr` is the noise 2^11 times louder: an ESTIMATE.  Where
//     fp64 still resolves the SM term (a >= uni_a_lin = 1e-16) it is within [-2.2, +2.1] decades of the 80-bit value
//     (180 000 walkers, tools/uni_estimate_spread.py); below that it errs by up to four decades either way.  rr below
//     uni_lo (uni_lo_nl in the unresolved regime): unitary; rr above uni_hi in the resolved regime: not unitary;
//  3. everything else is queued for the x87-faithful evaluation of gf_unitarity.hip, which decides.
// The device sampler, which needs the verdict inside the kernel, stops at tier 2 (rr against the scaled threshold).
constexpr double UNI_EST_SCALE = 2048.0;
constexpr double UNI_THRESHOLD = 1e-7 * UNI_EST_SCALE;    // estimate-only verdict
// doubles of LDS a group of LPW lanes sharing one walker needs: [nb][3] compositions + 3 unitarity accumulators per lane
#define GF_FGRP_PER_BIN 2                              // (f_e, f_mu) of a bin; the third follows from the unit sum
#define GF_FGRP_DOUBLES(nb, lpw) (GF_FGRP_PER_BIN * (nb) + 3 * (lpw))

struct UniAcc {
    double est_max;             // largest estimate over the bins tier 1 did not clear
    double clear_max;           // ... over those of them where the estimate may condemn (a >= uni_a_lin)
    unsigned long long amb;     // bins whose estimate reaches uni_lo
    double a_min;               // smallest SM weight over the bins (UNI_DEFER: what tier 1 needs, the rest runs later)
};

// What goes to the x87 arbitration for a walker with undecided bins: those bins AND every bin above the lowest of them.
// The amplification of the rounding noise grows with the energy (the SM weight a falls from bin to bin), so a bin that
// tier 2 would acquit on its own estimate while a LOWER bin is undecided owes that to an unusually quiet fp64 evaluation
// (or the chain's residual to an unusually loud x87 one): over 2.4 M walkers of eight (dimension, texture) cases the 21
// walkers whose status differed from the verdict of the chain run on every bin all had exactly that shape -- one failing
// bin, residual 1.03e-7 ... 1.42e-7, next to undecided ones (profiles/r03/tier2_pairs.txt).  The arbitration takes a
// walker's bins from the top down and stops at the first failure, so a failing walker costs nothing more; a unitary one
// has its top bins undecided anyway.  (The bins tier 1 clears have the largest a: they lie below the lowest undecided bin.)
__device__ __forceinline__ unsigned long long uni_arbitration_mask(unsigned long long amb, const GfBsm* __restrict__ tb)
{
    const int nbins = tb->nbins;
    if (amb == 0ull || tb->uni_own_bins_only) return amb;
    const unsigned long long lowest = amb & (~amb + 1ull);
    const unsigned long long all = nbins >= 64 ? ~0ull : (1ull << nbins) - 1ull;
    return all & ~(lowest - 1ull);
}

// How much of the unitarity verdict a call of flux_average / bin_moduli carries besides the values:
constexpr int UNI_NONE = 0;     // values only
constexpr int UNI_INLINE = 1;   // values + tiers 1 and 2 (small batches, the device sampler)
constexpr int UNI_DEFER = 2;    // values + the smallest SM weight: the evaluation kernel of a large batch queues the few
                                // walkers tier 1 does not clear for k_bsm_tier2 instead of making every wave run tier 2
constexpr int UNI_ONLY = 3;     // tiers 1 and 2 without the values (k_bsm_tier2)

struct Herm3 {          // 3x3 Hermitian: real diagonal + the three upper off-diagonals
    double d0, d1, d2;
    double r01, i01, r02, i02, r12, i12;
};

// m = w1 * a a^+ + w2 * b b^+ for complex 3-vectors a, b given as (re, im) triples.
__device__ __forceinline__ Herm3 rank2(double w1, const double ar[3], const double ai[3],
                                       double w2, const double br[3], const double bi[3])
{
    Herm3 m;
    m.d0 = w1 * fma(ar[0], ar[0], ai[0] * ai[0]) + w2 * fma(br[0], br[0], bi[0] * bi[0]);
    m.d1 = w1 * fma(ar[1], ar[1], ai[1] * ai[1]) + w2 * fma(br[1], br[1], bi[1] * bi[1]);
    m.d2 = w1 * fma(ar[2], ar[2], ai[2] * ai[2]) + w2 * fma(br[2], br[2], bi[2] * bi[2]);
    // (a a^+)_pq = a_p conj(a_q)
    m.r01 = w1 * fma(ar[0], ar[1], ai[0] * ai[1]) + w2 * fma(br[0], br[1], bi[0] * bi[1]);
    m.i01 = w1 * fma(ai[0], ar[1], -ar[0] * ai[1]) + w2 * fma(bi[0], br[1], -br[0] * bi[1]);
    m.r02 = w1 * fma(ar[0], ar[2], ai[0] * ai[2]) + w2 * fma(br[0], br[2], bi[0] * bi[2]);
    m.i02 = w1 * fma(ai[0], ar[2], -ar[0] * ai[2]) + w2 * fma(bi[0], br[2], -br[0] * bi[2]);
    m.r12 = w1 * fma(ar[1], ar[2], ai[1] * ai[2]) + w2 * fma(br[1], br[2], bi[1] * bi[2]);
    m.i12 = w1 * fma(ai[1], ar[2], -ar[1] * ai[2]) + w2 * fma(bi[1], br[2], -br[1] * bi[2]);
    return m;
}

// Columns 1 and 2 of the mixing matrix for (s12^2, c13^4, s23^2, delta): fr.py:116-162, SURVEY A.2.
// Once (twice for texture NONE) per walker against 20 bin diagonalisations: kept out of line so that its
// literals and temporaries do not inflate the bin loop's register allocation.
// Inlined since round 2 (+5-7 % on the bulk kernel, and no call frame: scratch 0); -DGF_MIX_NOINLINE restores the out-of-line
// call of round 1 (A/B in profiles/r02/ab_bsm_variants.txt)
#ifndef GF_MIX_NOINLINE
#define GF_MIX_ATTR __forceinline__
#else
#define GF_MIX_ATTR __attribute__((noinline))
#endif
static __device__ GF_MIX_ATTR void mixing_cols12(double s12_2, double c13_4, double s23_2, double dcp,
                                              double c1r[3], double c1i[3], double c2r[3], double c2i[3])
{
    const double c13_2 = fast_sqrt(c13_4);
    const double s12 = fast_sqrt(s12_2), c12 = fast_sqrt(1.0 - s12_2);
    const double c13 = fast_sqrt(c13_2), s13 = fast_sqrt(1.0 - c13_2);
    const double s23 = fast_sqrt(s23_2), c23 = fast_sqrt(1.0 - s23_2);
    double sd, cd;
    fast_sincos(dcp, &sd, &cd);
    // column 1: (s12 c13, c12 c23 - s12 s23 s13 e^{id}, -c12 s23 - s12 c23 s13 e^{id})
    const double t1 = s12 * s23 * s13, t2 = s12 * c23 * s13;
    c1r[0] = s12 * c13;            c1i[0] = 0.0;
    c1r[1] = fma(-t1, cd, c12 * c23);  c1i[1] = -t1 * sd;
    c1r[2] = fma(-t2, cd, -c12 * s23); c1i[2] = -t2 * sd;
    // column 2: (s13 e^{-id}, s23 c13, c23 c13)
    c2r[0] = s13 * cd;             c2i[0] = -s13 * sd;
    c2r[1] = s23 * c13;            c2i[1] = 0.0;
    c2r[2] = c23 * c13;            c2i[2] = 0.0;
}

// ---- per-walker invariants -------------------------------------------------------------------------
// H(E) = u S + v N (u = 1/2E, v = E^(d-3)) divided by its trace is the one-parameter family
//     H' = a S' + t N',   S' = S / tr S,  N' = N / tr N,  a = u tr S / (u tr S + v tr N),  t = 1 - a,
// so everything the bin needs that is polynomial in the entries of H' is a polynomial in (a, t) whose
// coefficients depend on the walker only:
//     b   = sum of principal 2x2 minors = a^2 b(S') + a t b(S',N') + t^2 b(N')            (fr.py:205)
//     det = a t (a m1 + t m2),  m1 = tr(adj(S') N'),  m2 = tr(adj(N') S')                  (fr.py:206)
//           (det S' = det N' = 0: both terms carry a zero eigenvalue, diag(0, ., .) in fr.py:383-393)
//     h_aa = a S'_aa + t N'_aa,   sum_{b != a} |h_ab|^2 = a^2 qS_a + a t qX_a + t^2 qN_a.
// These sums are at least as well conditioned as the products of entries they replace (b and det are sums
// of non-negative terms for positive semi-definite S', N') and cost ~40 fewer fp64 instructions per bin.
struct BinInv {
    double trS, trN;
    double tau;                 // trN / trS: a bin's SM weight is a = 1 / (1 + rho_k tau)
    double bS, bSN, bN, m1, m2;
    double sd0, sd1, nd0, nd1;
    double qS0, qX0, qN0, qS1, qX1, qN1;
    // the same quadratics in the one variable a (t = 1 - a), Horner form x0 + a (x1 + a x2), for the value loop: b, the two
    // off-diagonal sums and (linear) the two diagonal entries.  Absolute accuracy relative to the unit trace is all the
    // eigenvector-eigenvalue identity asks of them (gf_bsm.hip, numerics note); det keeps its factored form a t (a m1 + t m2).
    double b1, b2;              // b  = bN  + a (b1 + a b2)
    double o01, o02, o11, o12;  // os = qN + a (o.1 + a o.2)
    double e0, e1;              // d  = nd + a e.
};

__device__ __forceinline__ Herm3 scaled(const Herm3& x, double s)
{
    Herm3 y;
    y.d0 = x.d0 * s; y.d1 = x.d1 * s; y.d2 = x.d2 * s;
    y.r01 = x.r01 * s; y.i01 = x.i01 * s; y.r02 = x.r02 * s; y.i02 = x.i02 * s; y.r12 = x.r12 * s; y.i12 = x.i12 * s;
    return y;
}

__device__ __forceinline__ double abs2(double re, double im) { return fma(re, re, im * im); }

// sum of the principal 2x2 minors of a Hermitian 3x3
__device__ __forceinline__ double minor_sum(const Herm3& x)
{
    return (fma(x.d0, x.d1, -abs2(x.r01, x.i01)) + fma(x.d0, x.d2, -abs2(x.r02, x.i02))) + fma(x.d1, x.d2, -abs2(x.r12, x.i12));
}

// tr(adj(X) Y) for Hermitian X, Y: the coefficient of x^2 y in det(x X + y Y)
__device__ __forceinline__ double tr_adj(const Herm3& X, const Herm3& Y)
{
    const double a00 = fma(X.d1, X.d2, -abs2(X.r12, X.i12));
    const double a11 = fma(X.d0, X.d2, -abs2(X.r02, X.i02));
    const double a22 = fma(X.d0, X.d1, -abs2(X.r01, X.i01));
    const double a01r = fma(X.r02, X.r12, X.i02 * X.i12) - X.r01 * X.d2;      // X02 conj(X12) - X01 X22
    const double a01i = fma(X.i02, X.r12, -X.r02 * X.i12) - X.i01 * X.d2;
    const double a02r = fma(X.r01, X.r12, -X.i01 * X.i12) - X.r02 * X.d1;     // X01 X12 - X02 X11
    const double a02i = fma(X.r01, X.i12, X.i01 * X.r12) - X.i02 * X.d1;
    const double a12r = fma(X.r02, X.r01, X.i02 * X.i01) - X.d0 * X.r12;      // X02 conj(X01) - X00 X12
    const double a12i = fma(X.i02, X.r01, -X.r02 * X.i01) - X.d0 * X.i12;
    const double diag = fma(a00, Y.d0, fma(a11, Y.d1, a22 * Y.d2));
    const double off = fma(a01r, Y.r01, a01i * Y.i01) + fma(a02r, Y.r02, a02i * Y.i02) + fma(a12r, Y.r12, a12i * Y.i12);
    return fma(2.0, off, diag);
}

// S, N -> normalised Sn, Nn and the invariants
__device__ __forceinline__ void bin_invariants(const Herm3& S, const Herm3& N, Herm3& Sn, Herm3& Nn, BinInv& w)
{
    w.trS = (S.d0 + S.d1) + S.d2;
    w.trN = (N.d0 + N.d1) + N.d2;
    const double itS = fast_rcp(w.trS);
    w.tau = w.trN * itS;
    Sn = scaled(S, itS);
    Nn = scaled(N, fast_rcp(w.trN));
    w.bS = minor_sum(Sn);
    w.bN = minor_sum(Nn);
    const double x01 = fma(Sn.r01, Nn.r01, Sn.i01 * Nn.i01);                 // Re(S_ab conj(N_ab))
    const double x02 = fma(Sn.r02, Nn.r02, Sn.i02 * Nn.i02);
    const double x12 = fma(Sn.r12, Nn.r12, Sn.i12 * Nn.i12);
    w.bSN = (fma(Sn.d0, Nn.d1, Sn.d1 * Nn.d0) + fma(Sn.d0, Nn.d2, Sn.d2 * Nn.d0)) + fma(Sn.d1, Nn.d2, Sn.d2 * Nn.d1) -
            2.0 * ((x01 + x02) + x12);
    w.m1 = tr_adj(Sn, Nn);
    w.m2 = tr_adj(Nn, Sn);
    w.sd0 = Sn.d0; w.sd1 = Sn.d1; w.nd0 = Nn.d0; w.nd1 = Nn.d1;
    const double s01 = abs2(Sn.r01, Sn.i01), s02 = abs2(Sn.r02, Sn.i02), s12 = abs2(Sn.r12, Sn.i12);
    const double n01 = abs2(Nn.r01, Nn.i01), n02 = abs2(Nn.r02, Nn.i02), n12 = abs2(Nn.r12, Nn.i12);
    w.qS0 = s01 + s02; w.qS1 = s01 + s12;
    w.qN0 = n01 + n02; w.qN1 = n01 + n12;
    w.qX0 = 2.0 * (x01 + x02); w.qX1 = 2.0 * (x01 + x12);
    w.b1 = fma(-2.0, w.bN, w.bSN);  w.b2 = (w.bS - w.bSN) + w.bN;
    w.o01 = fma(-2.0, w.qN0, w.qX0); w.o02 = (w.qS0 - w.qX0) + w.qN0;
    w.o11 = fma(-2.0, w.qN1, w.qX1); w.o12 = (w.qS1 - w.qX1) + w.qN1;
    w.e0 = w.sd0 - w.nd0; w.e1 = w.sd1 - w.nd1;
}

// One energy bin: eigenvalues of the trace-normalised H by the trigonometric cubic solution
// (fr.py:204-214 with a = -1), moduli by the eigenvector-eigenvalue identity for the 2x2 block
// (alpha, i) in {e, mu} x {0, 1}; the remaining five follow from the unit row and column sums of |U|^2.
// Optionally the reference's eigenvector form for the unitarity status.
template <int UNI_MODE>
// `rho` = E^(d-3) / (1 / 2E) of the bin (GfBsm::rho).  Output: the 2x2 block p[0..1][0..1] only; the other five entries follow
// from the unit row and column sums where the caller needs them.
__device__ __forceinline__ void bin_moduli(const BinInv& w, const Herm3& Sn, const Herm3& Nn, double rho,
                                           double p[3][3], UniAcc& acc, int kbin, const GfBsm* __restrict__ tb)
{
    constexpr bool CHECK_UNI = UNI_MODE == UNI_ONLY;                 // the values and the estimate never share a loop body
    // H / tr H = a S' + t N' with a = u trS / (u trS + v trN) = 1 / (1 + rho tau), t = 1 - a = rho tau a: both to full relative
    // accuracy at either end (a -> 1e-20 at the top of a scale range, t -> 1e-20 at its bottom)
    const double rt = rho * w.tau;
    const double a = fast_rcp(rt + 1.0), t = rt * a;
    if (UNI_MODE == UNI_ONLY && !(a < tb->uni_a_ok)) return;          // tier 1 clears this bin: nothing to evaluate
    const double at = a * t;
    double b;
    if (CHECK_UNI) {                                                  // the estimate keeps the arithmetic its bands were calibrated on
        const double aa = a * a, tt = t * t;
        b = fma(aa, w.bS, fma(at, w.bSN, tt * w.bN));
    } else {
        b = fma(a, fma(a, w.b2, w.b1), w.bN);
    }
    const double det = at * fma(a, w.m1, t * w.m2);
    const double Q = fma(-3.0, b, 1.0) * (1.0 / 9.0);                 // (a^2 - 3b)/9 with a = -tr = -1
    const double R = (fma(9.0, b, -2.0) - 27.0 * det) * (1.0 / 54.0); // (2a^3 - 9ab + 27c)/54, c = -det
#ifndef GF_NO_RSQ_CUBE          // 1 / Q^(3/2) from the square root's own reciprocal instead of a second division (+1 %, A/B as above)
    double sq, x;
    {
        const double y = __builtin_amdgcn_rsq(Q);
        double g = Q * y, h = 0.5 * y;
        const double r0 = fma(-h, g, 0.5);
        g = fma(g, r0, g);
        h = fma(h, r0, h);
        const double d = fma(-g, g, Q);
        g = fma(d, h, g);
        h = fma(fma(-h, g, 0.5), h, h);                    // 1 / (2 sqrt Q), refined once more
        sq = Q == 0.0 ? 0.0 : g;
        const double ih = h + h;                          // 1 / sqrt Q
        x = R * ((ih * ih) * ih);
    }
#else
    const double sq = fast_sqrt(Q);
    double x = R * fast_rcp(Q * sq);
#endif
    // (rounding can leave |x| a hair above one: fast_acos_clamped takes that as one)
    const double phi = fast_acos_clamped(x) * (1.0 / 3.0);
    double sp, cp;
    sincos_small(phi, &sp, &cp);
    const double m2 = -2.0 * sq;
    const double HS3 = 0.8660254037844386;                            // sqrt(3)/2
    // E_k = m2 cos(phi + {0, -2pi/3, +2pi/3}) + 1/3 (fr.py:212-214) = mc + 1/3, -mc/2 +- mb + 1/3 with mc = m2 cos phi,
    // mb = m2 (sqrt(3)/2) sin phi; the eigenvalue gaps in closed form (no cancellation near a level crossing):
    // E0 - E1 = 3/2 mc - mb, E0 - E2 = 3/2 mc + mb, E1 - E2 = 2 mb
    const double mc = m2 * cp, mb = m2 * (HS3 * sp);
    const double E0 = mc + 1.0 / 3.0;
    const double eb = fma(-0.5, mc, 1.0 / 3.0);
    const double E1 = eb + mb, E2 = eb - mb;
    const double g01 = fma(1.5, mc, -mb), g02 = fma(1.5, mc, mb), g12 = mb + mb;
    const double r = fast_rcp((g01 * g02) * g12);
    const double inv0 = g12 * r;                                      // 1 / ((E0 - E1)(E0 - E2))
    const double inv1 = -(g02 * r);                                   // 1 / ((E1 - E2)(E1 - E0))
    double d0, d1, os0, os1;
    if (CHECK_UNI) {
        const double aa = a * a, tt = t * t;
        d0 = fma(a, w.sd0, t * w.nd0); d1 = fma(a, w.sd1, t * w.nd1);
        os0 = fma(aa, w.qS0, fma(at, w.qX0, tt * w.qN0));
        os1 = fma(aa, w.qS1, fma(at, w.qX1, tt * w.qN1));
    } else {
        d0 = fma(a, w.e0, w.nd0); d1 = fma(a, w.e1, w.nd1);
        os0 = fma(a, fma(a, w.o02, w.o01), w.qN0);
        os1 = fma(a, fma(a, w.o12, w.o11), w.qN1);
    }
    p[0][0] = fma(d0 - E1, d0 - E2, os0) * inv0;
    p[0][1] = fma(d0 - E2, d0 - E0, os0) * inv1;
    p[1][0] = fma(d1 - E1, d1 - E2, os1) * inv0;
    p[1][1] = fma(d1 - E2, d1 - E0, os1) * inv1;

    if (CHECK_UNI && a < tb->uni_a_ok) {
        // fr.py:216-236 in fp64, then fr.py:489-494.  h10 = conj(h01) etc.
        const double E[3] = {E0, E1, E2};
        const double d2 = fma(a, Sn.d2, t * Nn.d2);
        const double r01 = fma(a, Sn.r01, t * Nn.r01), i01 = fma(a, Sn.i01, t * Nn.i01);
        const double r02 = fma(a, Sn.r02, t * Nn.r02), i02 = fma(a, Sn.i02, t * Nn.i02);
        const double r12 = fma(a, Sn.r12, t * Nn.r12), i12 = fma(a, Sn.i12, t * Nn.i12);
        // tr|XX^+| is the sum of the squared norms of the normalised eigenvectors: 3 up to one rounding whatever
        // A, B, C are, so only the off-diagonal sum (fr.py:491) carries the signal; a NaN shows up there too.
        double f01r = 0, f01i = 0, f02r = 0, f02i = 0, f12r = 0, f12i = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double e0 = d0 - E[k], e1 = d1 - E[k], e2 = d2 - E[k];
            // A = h12 (h00 - E) - h10 h02 ; h10 h02 = conj(h01) h02
            const double Ar = fma(r12, e0, -fma(r01, r02, i01 * i02));
            const double Ai = fma(i12, e0, -fma(r01, i02, -i01 * r02));
            // B = h20 (h11 - E) - h21 h10 ; h20 = conj(h02), h21 h10 = conj(h12) conj(h01)
            const double Br = fma(r02, e1, -fma(r12, r01, -i12 * i01));
            const double Bi = fma(-i02, e1, fma(r12, i01, i12 * r01));
            // C = h10 (h22 - E) - h12 h20 ; h12 h20 = h12 conj(h02)
            const double Cr = fma(r01, e2, -fma(r12, r02, i12 * i02));
            const double Ci = fma(-i01, e2, -fma(i12, r02, -r12 * i02));
            const double a2 = fma(Ar, Ar, Ai * Ai), b2 = fma(Br, Br, Bi * Bi), c2 = fma(Cr, Cr, Ci * Ci);
            const double S = fma(a2, b2, fma(a2, c2, b2 * c2));
            const double invS = fast_rcp(S);
            // conj(A) conj(B) = (Ar Br - Ai Bi) - i (Ar Bi + Ai Br)
            const double abr = fma(Ar, Br, -Ai * Bi), abi = -fma(Ar, Bi, Ai * Br);
            // (XX^+)_01 += conj(A)conj(B) |C|^2 / S
            f01r = fma(abr * c2, invS, f01r); f01i = fma(abi * c2, invS, f01i);
            // (XX^+)_02 += conj(A) conj(B)^2 C / S = (conj(A)conj(B)) * (conj(B) C) / S
            const double bcr = fma(Br, Cr, Bi * Ci), bci = fma(Br, Ci, -Bi * Cr);   // conj(B) C
            f02r = fma(fma(abr, bcr, -abi * bci), invS, f02r);
            f02i = fma(fma(abr, bci, abi * bcr), invS, f02i);
            // (XX^+)_12 += |A|^2 C conj(B) / S
            f12r = fma(a2 * bcr, invS, f12r); f12i = fma(a2 * bci, invS, f12i);
        }
        const double off = fast_sqrt(fma(f01r, f01r, f01i * f01i)) + fast_sqrt(fma(f02r, f02r, f02i * f02i)) +
                           fast_sqrt(fma(f12r, f12r, f12i * f12i));
        double rr = 2.0 * off;                                         // |sum|XX^+| - 3| with the trace at 3
        if (rr != rr) rr = gf_inf();                                   // NaN fails the reference's test too
        acc.est_max = fmax(acc.est_max, rr);
        const bool resolved = a >= tb->uni_a_lin;                      // fp64 still sees the SM term
        if (resolved) acc.clear_max = fmax(acc.clear_max, rr);
        if (rr >= (resolved ? tb->uni_lo : tb->uni_lo_nl)) acc.amb |= 1ull << kbin;   // not safe from the estimate
    }
}

// The two terms of a walker's Hamiltonian as the evaluation kernel leaves them for k_bsm_tier2: [S (9) | N (9)]
constexpr int GF_SN_DOUBLES = 18;
__device__ __forceinline__ void store_sn(const Herm3& S, const Herm3& N, double* __restrict__ o)
{
    double2* o2 = reinterpret_cast<double2*>(o);                      // rows are 16-B aligned (18 doubles = 144 B)
    o2[0] = make_double2(S.d0, S.d1);   o2[1] = make_double2(S.d2, S.r01);  o2[2] = make_double2(S.i01, S.r02);
    o2[3] = make_double2(S.i02, S.r12); o2[4] = make_double2(S.i12, N.d0);  o2[5] = make_double2(N.d1, N.d2);
    o2[6] = make_double2(N.r01, N.i01); o2[7] = make_double2(N.r02, N.i02); o2[8] = make_double2(N.r12, N.i12);
}
__device__ __forceinline__ void load_sn(const double* __restrict__ o, Herm3& S, Herm3& N)
{
    const double2* o2 = reinterpret_cast<const double2*>(o);
    const double2 q0 = o2[0], q1 = o2[1], q2 = o2[2], q3 = o2[3], q4 = o2[4], q5 = o2[5], q6 = o2[6], q7 = o2[7], q8 = o2[8];
    S.d0 = q0.x; S.d1 = q0.y; S.d2 = q1.x; S.r01 = q1.y; S.i01 = q2.x; S.r02 = q2.y; S.i02 = q3.x; S.r12 = q3.y; S.i12 = q4.x;
    N.d0 = q4.y; N.d1 = q5.x; N.d2 = q5.y; N.r01 = q6.x; N.i01 = q6.y; N.r02 = q7.x; N.i02 = q7.y; N.r12 = q8.x; N.i12 = q8.y;
}

// Tiers 1-2 of one walker from its normalised Hamiltonian terms: k_bsm_tier2 (terms stored by the evaluation kernel) and
// flux_average<UNI_INLINE> (ahead of its value loop) run this same code on the same bin_invariants output, hence produce
// the same estimates bit for bit.  Keeping the estimate out of the value loop's body is what keeps the evaluation kernels
// free of scratch: the two loops need their registers one after the other, not together.  `sub`, `lpw`: this lane takes
// the bins sub, sub + lpw, ...
__device__ __forceinline__ void tier2_bins(const GfBsm* __restrict__ tb, const BinInv& w, const Herm3& Sn, const Herm3& Nn, UniAcc& acc,
                                           int sub = 0, int lpw = 1)
{
    // a = al / (al + be) clearly at or above uni_a_ok <=> be < skip_be al: tier 1 clears the bin (bin_moduli decides
    // the borderline itself)
    const double skip_be = (1.0 - tb->uni_a_ok) * fast_rcp(tb->uni_a_ok) * (1.0 - 1e-9);
    const int nb = tb->nbins;
    for (int k = sub; k < nb; k += lpw) {
        const double rho = tb->rho[k];
        if (rho * w.tau < skip_be) continue;
        double p[3][3];
        bin_moduli<UNI_ONLY>(w, Sn, Nn, rho, p, acc, k, tb);
    }
}
__device__ __forceinline__ void tier2_from_sn(const GfBsm* __restrict__ tb, const Herm3& S, const Herm3& N, UniAcc& acc)
{
    Herm3 Sn, Nn;
    BinInv w;
    bin_invariants(S, N, Sn, Nn, w);
    tier2_bins(tb, w, Sn, Nn, acc);
}

// The two terms of a walker's Hamiltonian before the energy factors: S = U diag(0, m21, m3x) U^+ (fr.py:383-386) and
// N = U~ diag(0, sc1, sc2) U~^+ (fr.py:380-393).
__device__ __forceinline__ void hamiltonian_terms(const GfCommon& c, const GfBsm* __restrict__ tb, const double* ttab,
                                                  const double* row, Herm3& S, Herm3& N)
{
    // SM part, per walker: U diag(0, m21, m3x) U^+ = m21 u1 u1^+ + m3x u2 u2^+   (fr.py:383-386)
    double c1r[3], c1i[3], c2r[3], c2i[3];
    mixing_cols12(pick(row, c.idx_sm[0], c.sm_fixed[0]), pick(row, c.idx_sm[1], c.sm_fixed[1]),
                  pick(row, c.idx_sm[2], c.sm_fixed[2]), pick(row, c.idx_sm[3], c.sm_fixed[3]), c1r, c1i, c2r, c2i);
    S = rank2(pick(row, c.idx_mass[0], c.mass_fixed[0]), c1r, c1i,
                          pick(row, c.idx_mass[1], c.mass_fixed[1]), c2r, c2i);
    // NP part, per walker: sc1 T1 + sc2 T2, sc2 = 10^logLam, sc1 = sc2/100   (fr.py:380-393)
    const double sc2 = pow10_scale(pick(row, c.idx_scale, c.scale_fixed));
    const double sc1 = sc2 / 100.0;
    if (tb->texture == TEX_NONE) {
        mixing_cols12(pick(row, c.idx_mm[0], c.mm_fixed[0]), pick(row, c.idx_mm[1], c.mm_fixed[1]),
                      pick(row, c.idx_mm[2], c.mm_fixed[2]), pick(row, c.idx_mm[3], c.mm_fixed[3]), c1r, c1i, c2r, c2i);
        N = rank2(sc1, c1r, c1i, sc2, c2r, c2i);
    } else {
        // ttab (LDS): the 18 entries of T1, T2 that a Hermitian 3x3 needs, laid out {t1, t2} pairs
        N.d0 = fma(sc1, ttab[0], sc2 * ttab[1]);
        N.d1 = fma(sc1, ttab[2], sc2 * ttab[3]);
        N.d2 = fma(sc1, ttab[4], sc2 * ttab[5]);
        N.r01 = fma(sc1, ttab[6], sc2 * ttab[7]);   N.i01 = fma(sc1, ttab[8], sc2 * ttab[9]);
        N.r02 = fma(sc1, ttab[10], sc2 * ttab[11]); N.i02 = fma(sc1, ttab[12], sc2 * ttab[13]);
        N.r12 = fma(sc1, ttab[14], sc2 * ttab[15]); N.i12 = fma(sc1, ttab[16], sc2 * ttab[17]);
    }
}

// flux_averaged_BSMu for one walker (fr.py:403-458).  Returns the normalised composition and the worst
// unitarity residual over the bins.
//
// LPW > 1 (device sampler on small ensembles): LPW adjacent lanes of one wave hold the SAME walker; each
// computes the walker's invariants (redundantly, bit-identical) and the bins k = sub, sub + LPW, ...; the
// per-bin compositions meet in LDS (`fgrp`: GF_FGRP_DOUBLES(nb, LPW) doubles, private to the lane group) and
// every lane then runs the same in-order weighted sum, so the result is bitwise the LPW = 1 result on all LPW
// lanes.  The walker's critical path drops from nb bins to ceil(nb / LPW).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_terms`: called once the Hamiltonian terms are built, i.e. behind the last out-of-line call of the prologue (10^x)
// and ahead of the bin loop -- where k_bsm starts the LDS-DMA copy of its next tile (a callee's entry waits for every
// outstanding memory operation, so a copy started earlier would be waited for at once).
template <int UNI_MODE, int LPW = 1, class Hook = NoHook>
__device__ __forceinline__ void flux_average(const GfCommon& c, const GfBsm* __restrict__ tb, const double* ttab,
                                             const double* row, double fr[3], UniAcc& acc, int sub = 0,
                                             double* fgrp = nullptr, double* sn_out = nullptr, Hook after_terms = Hook())
{
    Herm3 S, N;
    hamiltonian_terms(c, tb, ttab, row, S, N);
    after_terms();
    // the smallest SM weight over the bins, a_k = 1 / (1 + (v_k / u_k) trN / trS), from the largest v_k / u_k of the table
    // (the traces as bin_invariants forms them)
    if (UNI_MODE == UNI_DEFER || UNI_MODE == UNI_INLINE) {
        const double trS = (S.d0 + S.d1) + S.d2, trN = (N.d0 + N.d1) + N.d2;
        acc.a_min = fast_rcp(fma(tb->rho_max, trN * fast_rcp(trS), 1.0));
        // tier 1 does not clear this walker: leave its two Hamiltonian terms for k_bsm_tier2 (18 doubles; the stores
        // retire behind the bin loop), which then needs neither theta nor the per-walker prologue.  Ahead of
        // bin_invariants: S and N die there, and the kernel's register peak is there too.
        if (sn_out && acc.a_min < tb->uni_a_ok) store_sn(S, N, sn_out);
    }
    Herm3 Sn, Nn;
    BinInv w;
    bin_invariants(S, N, Sn, Nn, w);
    // source_flux[k] = source_ratio * E_k^gamma (fr.py:416-419) enters u_to_fr only through
    // src / sum(src) (fr.py:535): the E^gamma factor cancels, so the spectral index has no effect.
    const double isrc = fast_rcp(c.src_fixed_sum);
    const double s2 = c.src_fixed[2] * isrc;
    const double ds0 = fma(c.src_fixed[0], isrc, -s2), ds1 = fma(c.src_fixed[1], isrc, -s2);
    // fr.py:451 u_to_fr: f = P P^T s with P = |U|^2, s = src / sum(src).  P's rows and columns sum to one, so only its 2x2 block
    // is needed: w = P^T s = s2 + ds0 P_0. + ds1 P_1. (ds = s - s2), f_b = w2 + P_b0 (w0 - w2) + P_b1 (w1 - w2) for b = e, mu, and
    // f_tau = (sum s) - f_e - f_mu -- which makes the tau row of the width-weighted sum (sum of the widths) - a0 - a1: 18
    // instructions per bin where the full 3x3 products took 31.
    double a0 = 0.0, a1 = 0.0;
    const int nb = tb->nbins;
    for (int k = (LPW > 1 ? sub : 0); k < nb; k += LPW) {
        double p[3][3];
        bin_moduli<UNI_MODE>(w, Sn, Nn, tb->rho[k], p, acc, k, tb);
        const double p02 = (1.0 - p[0][0]) - p[0][1], p12 = (1.0 - p[1][0]) - p[1][1];
        const double w0 = fma(ds1, p[1][0], fma(ds0, p[0][0], s2));
        const double w1 = fma(ds1, p[1][1], fma(ds0, p[0][1], s2));
        const double w2 = fma(ds1, p12, fma(ds0, p02, s2));
        const double dw0 = w0 - w2, dw1 = w1 - w2;
        const double f0 = fma(p[0][1], dw1, fma(p[0][0], dw0, w2));
        const double f1 = fma(p[1][1], dw1, fma(p[1][0], dw0, w2));
        if (LPW > 1) {
            fgrp[GF_FGRP_PER_BIN * k] = f0; fgrp[GF_FGRP_PER_BIN * k + 1] = f1;
        } else {
            const double wk = tb->weight[k];
            a0 = fma(f0, wk, a0); a1 = fma(f1, wk, a1);                         // fr.py:454
        }
    }
    // UNI_INLINE, second phase: tier 2 for the walkers tier 1 does not clear (none where the posterior lives).  The terms
    // are rebuilt from the row rather than kept: 36 registers live across the value loop would spill (the row offset goes
    // through an empty asm so that the compiler does not merge the two evaluations and keep them after all).  Tried and
    // worse: tier 2 ahead of the value loop (more spills), and as a noinline call (the callee saves ~200 registers).
    if (UNI_MODE == UNI_INLINE && __builtin_expect(acc.a_min < tb->uni_a_ok, 0)) {
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const double* row2 = row + opaque;
        Herm3 S2, N2;
        hamiltonian_terms(c, tb, ttab, row2, S2, N2);
        Herm3 Sn2, Nn2;
        BinInv w2;
        bin_invariants(S2, N2, Sn2, Nn2, w2);
        tier2_bins(tb, w2, Sn2, Nn2, acc, LPW > 1 ? sub : 0, LPW);
    }
    if (LPW > 1) {
        if (UNI_MODE == UNI_INLINE) {
            fgrp[GF_FGRP_PER_BIN * nb + sub] = acc.est_max;
            fgrp[GF_FGRP_PER_BIN * nb + LPW + sub] = acc.clear_max;
            fgrp[GF_FGRP_PER_BIN * nb + 2 * LPW + sub] = __longlong_as_double((long long)acc.amb);
        }
        // the lanes of a group sit in one wave: its LDS operations retire in order, the fence keeps the compiler honest
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = 0; k < nb; ++k) {
            const double wk = tb->weight[k];
            a0 = fma(fgrp[GF_FGRP_PER_BIN * k], wk, a0); a1 = fma(fgrp[GF_FGRP_PER_BIN * k + 1], wk, a1);
        }
        if (UNI_MODE == UNI_INLINE) {
#pragma unroll
            for (int j = 0; j < LPW; ++j) {
                acc.est_max = fmax(acc.est_max, fgrp[GF_FGRP_PER_BIN * nb + j]);
                acc.clear_max = fmax(acc.clear_max, fgrp[GF_FGRP_PER_BIN * nb + LPW + j]);
                acc.amb |= (unsigned long long)__double_as_longlong(fgrp[GF_FGRP_PER_BIN * nb + 2 * LPW + j]);
            }
        }
    }
    const double a2 = (tb->wsum - a0) - a1;                         // the tau row: (sum of the widths) sum(s) - a0 - a1, sum(s) = 1
    const double inv = fast_rcp((a0 + a1) + a2);                    // fr.py:457
    fr[0] = a0 * inv; fr[1] = a1 * inv; fr[2] = a2 * inv;
}

// box + priors from the LDS constant table (same layout as the SM kernels: {lo, hi, loc, 1/sigma} per
// column), branch-free
template <int NDIM>
__device__ __forceinline__ bool lnprior_tab(const double* ctab, const double* row, int ndim_rt, double prior_const,
                                            double& lp)
{
    const int ndim = NDIM ? NDIM : ndim_rt;
    bool inbox = true;
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < (NDIM ? NDIM : GF_MAX_DIM); ++d) {
        if (!NDIM && d >= ndim) break;
        const double x = row[d];
        const double2 lh = *reinterpret_cast<const double2*>(ctab + 4 * d);
        const double2 ls = *reinterpret_cast<const double2*>(ctab + 4 * d + 2);
        inbox = inbox & (x >= lh.x) & (x <= lh.y);
        const double z = (x - ls.x) * ls.y;
        acc = fma(-0.5 * z, z, acc);
    }
    lp = acc + prior_const;
    return inbox;
}


}  // namespace gfdev

#pragma clang fp contract(fast)
