// gf_x87.hpp -- x87 extended precision (64-bit significand, round to nearest even) emulated in pairs of
// doubles, and on top of it the reference's closed-form diagonalisation + unitarity test, operation by operation.
//
// Why.  golemflavor/fr.py computes in np.float128 / np.complex256 (fr.py:22-30), i.e. the x87 80-bit format, and
// raises "Matrix is not unitary!" when max(|tr f - 3|, |sum f - 3|) >= 1e-7 for f = |X X^+| (fr.py:461-499 via
// :398-399).  In exact arithmetic X is unitary; the number compared with 1e-7 is the rounding noise of the 64-bit
// significand, amplified by the cancellations of the eigenvector formula (fr.py:216-236).  A walker is therefore
// accepted or killed according to the *size of the x87 unit roundoff* and to the *order of the operations*: an fp64
// evaluation (2^11 times the noise) can only estimate the verdict.  The kernels use that estimate to sort walkers into
// clearly-unitary / clearly-not / undecided; the undecided (walker, bin) pairs come here.
//
// Representation: value = hi + lo, |lo| <= ulp(hi)/2, the sum having at most 64 significant bits.  Every operation is
// carried out in double-double arithmetic (error-free transformations with FMA, >= 2^-104 relative) and the result
// rounded to 64 bits, ties to even -- which is the correctly rounded x87 result except when the exact result lies
// within ~2^-104 of a rounding boundary (probability ~2^-39 per operation).  + - * / sqrt are checked bit for bit
// against native `long double` on the host (tests/x87/, tests/test_x87_emulation.py).  The transcendental functions
// of the chain (asinl, acosl, sinl, cosl; cacosl / ccosl / csqrtl near the real axis) are evaluated to ~2^-100 and
// rounded: correctly rounded results, where glibc / the x87 microcode are faithful (< 1 ulp) ones -- they agree in the
// large majority of calls and differ by one unit in the last place otherwise.  Exponent range: fp64's; the Hamiltonian
// is rescaled by an exact power of two first (the x87 format never over- or underflows on this path, so the reference's
// significands are unchanged by the rescaling).
//
// The operation order below is numpy's, pinned on the reference's own residuals (tests/golden G17; the CPU oracle
// reproduces them bit for bit): complex products as (ac - bd, ad + bc); complex / as Smith's algorithm with a real
// divisor promoted to (d, 0), i.e. x * (1 / d); a**3 = a * (a * a); np.dot accumulating from zero in index order;
// np.trace left to right; np.sum over nine entries as numpy's eight-way pairwise sum plus the ninth.
//
// Compiles for the device (hipcc) and for the host (tests/x87/x87_host.cpp, g++): the host build exists so that the
// arithmetic can be compared with the CPU's own x87 unit; the library itself never evaluates anything on the host.
#pragma once
#include <stdint.h>
#include <string.h>

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GFX_HD __host__ __device__
#else
#define GFX_HD
#endif
// The chain below is ~37 000 instructions when everything is inlined and every 3x3 loop unrolled: 150 KB of straight-line
// code, far more than the instruction cache holds, with every wave somewhere else in it.  Measured on the arbitration of a
// failing region (tools/ab_unitarity.sh, profiles/r02/ab_unitarity_code_size.txt): keeping the 3x3 loops rolled (GFX_ROLLED)
// and the large, rarely called pieces out of line (GFX_BIG: the sine/cosine series, division, square root) is 18 % faster
// (27 000 instructions).  Making functions of the small primitives as well (x_add, x_mul: 50-60 instructions each; 13 000
// instructions in all) is 25 % SLOWER than the all-inline build, and of the complex product (320 instructions) no faster:
// the call sequence and its register conventions cost what the instruction fetches save.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GFX87_INLINE_ALL)
#define GFX_BIG __attribute__((noinline))
#else
#define GFX_BIG inline
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GFX87_UNROLL_ALL)
#define GFX_ROLLED _Pragma("unroll 1")
#else
#define GFX_ROLLED
#endif

// The series of dd_sincos / dd_sin_small are the exception (round 4): rolled, every Horner step fetched its coefficient with a scalar
// load and waited for it -- ~30 round trips to the constant cache per call, one of them for a single fused multiply-add in the fp64
// tails.  They are out-of-line functions (one copy each), so unrolling them costs 2 KB of code and turns the table into literals.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GFX87_SERIES_ROLLED)
#define GFX_SERIES _Pragma("unroll")
#else
#define GFX_SERIES GFX_ROLLED
#endif

// A/B switches for the arbitration kernels (tools/build_variants.sh): the complex / scalar primitives out of line
#if defined(__HIP_DEVICE_COMPILE__) && defined(GFX87_CMUL_NOINLINE)
#define GFX_CMUL __attribute__((noinline))
#else
#define GFX_CMUL inline
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(GFX87_CADD_NOINLINE)
#define GFX_CADD __attribute__((noinline))
#else
#define GFX_CADD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(GFX87_X_NOINLINE)
#define GFX_XOP __attribute__((noinline))
#else
#define GFX_XOP inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

namespace gfx87 {

struct x87 { double hi, lo; };
struct cx87 { x87 re, im; };

GFX_HD inline double x_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
GFX_HD inline int64_t x_bits(double x) { int64_t b; memcpy(&b, &x, 8); return b; }
GFX_HD inline double x_from_bits(int64_t b) { double x; memcpy(&x, &b, 8); return x; }

GFX_HD inline void two_sum(double a, double b, double& s, double& e)
{
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
GFX_HD inline void quick_two_sum(double a, double b, double& s, double& e)   // |a| >= |b| or a == 0
{
    s = a + b;
    e = b - (s - a);
}
GFX_HD inline void two_prod(double a, double b, double& p, double& e)
{
    p = a * b;
    e = x_fma(a, b, -p);
}

// ---- double-double (unrounded) ---------------------------------------------------------------------
struct dd { double hi, lo; };

GFX_HD inline dd dd_norm(double a, double b) { dd r; quick_two_sum(a, b, r.hi, r.lo); return r; }
GFX_HD inline dd dd_from(double a) { dd r = {a, 0.0}; return r; }
GFX_HD inline dd dd_neg(dd a) { dd r = {-a.hi, -a.lo}; return r; }

GFX_HD inline dd dd_add(dd a, dd b)
{
    double s1, e1, s2, e2;
    two_sum(a.hi, b.hi, s1, e1);
    two_sum(a.lo, b.lo, s2, e2);
    e1 += s2;
    quick_two_sum(s1, e1, s1, e1);
    e1 += e2;
    return dd_norm(s1, e1);
}
GFX_HD inline dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
// a double plus a double-double: the heads exactly, the tail in fp64 (11 operations instead of dd_add's 20; where the sum cancels,
// the heads' sum is exact and small and the tail is added to it exactly)
GFX_HD inline dd dd_add_d(double a, dd b)
{
#if defined(GFX87_ROUND64_R3)
    return dd_add(dd_from(a), b);
#endif
    double s, e;
    two_sum(a, b.hi, s, e);
    e += b.lo;
    return dd_norm(s, e);
}

GFX_HD inline dd dd_mul(dd a, dd b)
{
    double p, e;
    two_prod(a.hi, b.hi, p, e);
    double t = a.hi * b.lo;
    t = x_fma(a.lo, b.hi, t);
    t = x_fma(a.lo, b.lo, t);
    e += t;
    return dd_norm(p, e);
}
GFX_HD inline dd dd_mul_d(dd a, double b)
{
    double p, e;
    two_prod(a.hi, b, p, e);
    e = x_fma(a.lo, b, e);
    return dd_norm(p, e);
}

// c * t + k for Horner's rule when |k| >= |c t| (the series below: every coefficient is at least twice the product it is added
// to) and the two do not cancel: the product is not normalised on its own, the heads are added by the three-operation sum that
// needs the larger operand first, the tails in plain fp64 -- 14 operations instead of dd_mul + dd_add's 29, the same ~2^-104.
GFX_HD inline dd dd_horner(dd c, dd t, dd k)
{
#if defined(GFX87_ROUND64_R3)
    return dd_add(dd_mul(c, t), k);
#endif
    double p, e;
    two_prod(c.hi, t.hi, p, e);
    double u = c.hi * t.lo;
    u = x_fma(c.lo, t.hi, u);
    const double s = k.hi + p;
    double r = p - (s - k.hi);
    r += (k.lo + e) + u;
    return dd_norm(s, r);
}

GFX_HD inline dd dd_div(dd a, dd b)
{
#if defined(GFX87_ROUND64_R3)
    const double q1 = a.hi / b.hi;
    dd r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.hi / b.hi;
    dd q = dd_norm(q1, q2);
    return dd_add(q, dd_from(q3));
#else
    // Round 4: ONE fp64 division (the reciprocal of the divisor's head) instead of three, and the remainder a - q1 b formed directly --
    // the heads' difference is exact (q1 b.hi is within an ulp of a.hi), the rest is of relative size 2^-52 and needs fp64 only.
    // q1 + q2 + q3 is the quotient to ~2^-104, as before; ~45 operations instead of ~130.
    const double r0 = 1.0 / b.hi;
    const double q1 = a.hi * r0;
    double p, e;
    two_prod(q1, b.hi, p, e);
    double r = (((a.hi - p) - e) + a.lo) - q1 * b.lo;              // a - q1 b
    const double q2 = r * r0;
    two_prod(q2, b.hi, p, e);
    r = ((r - p) - e) - q2 * b.lo;                                 // ... - q2 b
    const double q3 = r * r0;
    double s, t;
    quick_two_sum(q1, q2, s, t);
    return dd_norm(s, t + q3);
#endif
}
GFX_HD inline dd dd_div_d(dd a, double b)
{
    const double q1 = a.hi / b;
    double p, e;
    two_prod(q1, b, p, e);
    const double r = ((a.hi - p) - e) + a.lo;
    return dd_norm(q1, r / b);
}

GFX_HD inline dd dd_sqrt(dd a)
{
    if (a.hi <= 0.0) { dd z = {a.hi == 0.0 ? 0.0 : NAN, 0.0}; return z; }
    const double s0 = sqrt(a.hi);
    double p, e;
    two_prod(s0, s0, p, e);
    dd r;
#if defined(GFX87_ROUND64_R3)
    const dd d = dd_sub(a, dd_norm(p, e));
    two_sum(s0, d.hi / (2.0 * s0), r.hi, r.lo);
#else
    // a - s0^2: the heads' difference is exact (s0^2 is within an ulp of a.hi); the correction is below an ulp of s0
    const double d = ((a.hi - p) - e) + a.lo;
    quick_two_sum(s0, d / (2.0 * s0), r.hi, r.lo);
#endif
    return r;
}

// ---- rounding to a 64-bit significand, ties to even ---------------------------------------------------
// Every + - * of the chain ends here, and the arbitration kernels are bound by the number of instructions these primitives
// are made of (profiles/r04/x87_integer_emulation_probe.txt: ~57 dependent instructions per multiply-add).
// Round 3 made it branch-free; round 4 takes it down from ~21 instructions to 10 (GFX87_ROUND64_R3 keeps round 3's form for A/B):
//   * the rounding constant c = 1.5 * 2^(e-11) is put together in the HIGH WORD of hi's own bit pattern: exponent field minus 11
//     (minus 12 when the value lies in the binade below `hi`: hi a power of two and the tail of the other sign -- a zero tail of
//     the other sign may take that branch too, it rounds to itself on any grid), top mantissa bit set: 32-bit operations only;
//   * no special cases.  A signed maximum with zero turns the constant of everything below 2^-1011 (zeros included) into 0 or a
//     denormal, and adding and subtracting that changes nothing: zeros and tiny numbers pass as they are.  For inf / NaN the constant
//     is an ordinary number and the tail comes out NaN; round 3 returned (inf, 0) -- either way the residual ends as NaN, which the
//     chain reports as +inf, fr.py's "not unitary" (an inf can only come from a division by an exact zero: fr.py:210, :232-236);
//   * the pair is returned as (hi, rounded tail) WITHOUT the final renormalisation.  |tail| <= ulp(hi)/2 still holds (half an ulp is
//     on the rounding grid), the VALUE is the same 64-bit number; only a tail of exactly half an ulp is no longer folded into hi.
//     Nothing computes with the parts except through the error-free transformations, which do not care; x_ge compares by difference.
// Results are unchanged as VALUES on everything finite (tests/test_x87_emulation.py compares the two forms on 48 M operand pairs
// and whole chains; the device tests compare verdicts with the stored ones of round 3).
#if !defined(GFX87_ROUND64_R3)
GFX_HD inline x87 round64(dd v)
{
    const uint64_t b = (uint64_t)x_bits(v.hi);
    const uint32_t hw = (uint32_t)(b >> 32), lw = (uint32_t)b;
    const uint32_t tw = (uint32_t)((uint64_t)x_bits(v.lo) >> 32);
    uint32_t sx = hw ^ tw;                                                          // bit 31: the signs of hi and of the tail differ
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(sx));                          // (keeps the compiler from widening the sign test into a 64-bit xor + compare)
#endif
    const bool below = (((hw & 0x000fffffu) | lw) == 0u) & ((int32_t)sx < 0);
    int32_t ch = (int32_t)((hw & 0x7ff00000u) + (below ? (0x00080000u - (12u << 20)) : (0x00080000u - (11u << 20))));
    ch = ch > 0 ? ch : 0;
    const double c = x_from_bits((int64_t)((uint64_t)(uint32_t)ch << 32));
    x87 r;
    r.hi = v.hi;
    r.lo = (v.lo + c) - c;
    return r;
}
#else
GFX_HD inline x87 round64(dd v)
{
    const int64_t b = x_bits(v.hi);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    // the value's own binade: a negative tail under a power of two lies in the binade below
    const bool below = ((b & 0xfffffffffffffLL) == 0) & (v.lo != 0.0) & ((v.lo < 0.0) != (v.hi < 0.0));
    e -= below ? 1 : 0;
    // q = 2^(e-63) is the unit in the last place; adding c = 1.5 * 2^52 q = 1.5 * 2^(e-11) rounds the tail to a multiple of q,
    // ties to even.  c's bit pattern directly: exponent field e - 11 + 1023, top mantissa bit set
    const double c = x_from_bits(((int64_t)(e + 1012) << 52) | 0x0008000000000000LL);
    const double lr = (v.lo + c) - c;
    const double hs = v.hi + lr;
    const double ls = lr - (hs - v.hi);                                          // quick_two_sum(v.hi, lr)
    const bool special = (v.hi == 0.0) | !(fabs(v.hi) <= 1.7976931348623157e308);    // zero, inf, NaN: (hi, 0)
    const bool tiny = e < -1000;                                                 // out of the emulated range: keep the tail
    x87 r;
    r.hi = (special | tiny) ? v.hi : hs;
    r.lo = special ? 0.0 : (tiny ? v.lo : ls);
    return r;
}
#endif

GFX_HD inline dd as_dd(x87 a) { dd r = {a.hi, a.lo}; return r; }
GFX_HD inline x87 x_from(double a) { x87 r = {a, 0.0}; return r; }
GFX_HD inline x87 x_neg(x87 a) { x87 r = {-a.hi, -a.lo}; return r; }
GFX_HD inline x87 x_abs(x87 a) { return a.hi < 0.0 ? x_neg(a) : a; }
GFX_HD inline double x_to_double(x87 a) { return a.hi + a.lo; }
GFX_HD inline bool x_is_zero(x87 a) { return a.hi == 0.0; }
// a >= b by the sign of the difference: the heads' difference is exact when they are close, and decides alone when they are not
GFX_HD inline bool x_ge(x87 a, x87 b) { return (a.hi - b.hi) + (a.lo - b.lo) >= 0.0; }

// The sum of two x87 NUMBERS (tails of at most 11 bits) needs less than the sum of two general double-doubles: the heads are added
// exactly (two_sum), the tails and the heads' error in plain fp64.  a.lo + b.lo is exact unless the exponents lie more than
// 41 apart, and then its rounding error is below 2^-106 of the result -- dd_add's own error class (its `e1 += s2` rounds in the same
// place); under cancellation the heads' difference is exact and the tails' sum as well.  11 floating-point operations instead of 20.
GFX_HD inline dd dd_add_x(x87 a, x87 b)
{
    double s, e;
    two_sum(a.hi, b.hi, s, e);
    e += a.lo + b.lo;
    return dd_norm(s, e);
}
#if !defined(GFX87_ROUND64_R3)
GFX_HD GFX_XOP x87 x_add(x87 a, x87 b) { return round64(dd_add_x(a, b)); }
#else
GFX_HD GFX_XOP x87 x_add(x87 a, x87 b) { return round64(dd_add(as_dd(a), as_dd(b))); }
#endif
GFX_HD inline x87 x_sub(x87 a, x87 b) { return x_add(a, x_neg(b)); }      // dd_sub is dd_add of the negation: the same operations
GFX_HD GFX_XOP x87 x_mul(x87 a, x87 b) { return round64(dd_mul(as_dd(a), as_dd(b))); }
GFX_HD GFX_BIG x87 x_div(x87 a, x87 b) { return round64(dd_div(as_dd(a), as_dd(b))); }
GFX_HD GFX_BIG x87 x_sqrt(x87 a) { return round64(dd_sqrt(as_dd(a))); }
GFX_HD inline x87 x_scale2(x87 a, double p2) { x87 r = {a.hi * p2, a.lo * p2}; return r; }    // exact: p2 a power of two

// ---- transcendental functions, ~2^-100, for the (cold) chain ---------------------------------------------
// pi/2 to three doubles
#define GFX_PIO2_1 1.5707963267948966
#define GFX_PIO2_2 6.123233995736766e-17
#define GFX_PIO2_3 (-1.4973849048591698e-33)

// 1 / n! to double-double, n = 0 .. 29 (mpmath, 200 bits)
struct ddc { double hi, lo; };
#define GFX_INV_FACT_TABLE                                                                                         \
    {1.0, 0.0}, {1.0, 0.0}, {0.5, 0.0}, {0.16666666666666666, 9.25185853854297e-18},                               \
    {0.041666666666666664, 2.3129646346357427e-18}, {0.008333333333333333, 1.1564823173178714e-19},               \
    {0.001388888888888889, -5.300543954373577e-20}, {0.0001984126984126984, 1.7209558293420705e-22},              \
    {2.48015873015873e-05, 2.1511947866775882e-23}, {2.7557319223985893e-06, -1.858393274046472e-22},             \
    {2.755731922398589e-07, 2.3767714622250297e-23}, {2.505210838544172e-08, -1.448814070935912e-24},             \
    {2.08767569878681e-09, -1.20734505911326e-25}, {1.6059043836821613e-10, 1.2585294588752098e-26},              \
    {1.1470745597729725e-11, 2.0655512752830745e-28}, {7.647163731819816e-13, 7.03872877733453e-30},              \
    {4.779477332387385e-14, 4.399205485834081e-31}, {2.8114572543455206e-15, 1.6508842730861433e-31},             \
    {1.5619206968586225e-16, 1.1910679660273754e-32}, {8.22063524662433e-18, 2.2141894119604265e-34},             \
    {4.110317623312165e-19, 1.4412973378659527e-36}, {1.9572941063391263e-20, -1.3643503830087908e-36},           \
    {8.896791392450574e-22, -7.911402614872376e-38}, {3.868170170630684e-23, -8.843177655482344e-40},             \
    {1.6117375710961184e-24, -3.6846573564509766e-41}, {6.446950284384474e-26, -1.9330404233703465e-42},          \
    {2.4795962632247976e-27, -1.2953730964765229e-43}, {9.183689863795546e-29, 1.4303150396787322e-45},           \
    {3.279889237069838e-30, 1.5117542744029879e-46}, {1.1309962886447716e-31, 1.0498015412959506e-47}

// sin and cos of a double-double, |x| < ~1e4, to ~2^-102.  Round 3: the Maclaurin series by Horner's rule with the
// reciprocal factorials as double-double constants (round 2 divided every term by its index pair: two double-double
// divisions per order, 14 orders -- 2 100 instructions a call, and the chain makes eleven calls per (walker, bin) when the
// mixing angles are sampled).  The orders whose terms lie below 2^-52 of the leading one are summed in plain fp64 first:
// their rounding errors are below 2^-104 of the result.  |r| <= pi/4 + eps after the reduction; r^30 / 30! < 2^-117.
GFX_HD GFX_BIG void dd_sincos(dd x, dd& sn, dd& cs)
{
    static const ddc F[30] = {GFX_INV_FACT_TABLE};
    const double kf = nearbyint(x.hi * 0.6366197723675814);
    dd r = x;
    if (kf != 0.0) {
        double p, e;
        two_prod(kf, GFX_PIO2_1, p, e);
        r = dd_sub(r, dd_norm(p, e));
        two_prod(kf, GFX_PIO2_2, p, e);
        r = dd_sub(r, dd_norm(p, e));
        r = dd_sub(r, dd_from(kf * GFX_PIO2_3));
    }
    const dd t = dd_mul(r, r);
    // sin r = r S(t),  S = sum_k (-1)^k t^k / (2k+1)!,  k = 0 .. 14;   cos r = C(t),  C = sum_k (-1)^k t^k / (2k)!,  k = 0 .. 14
    // fp64 tails: S from k = 8 (t^8 / 17! < 2^-54), C from k = 9 (t^9 / 18! < 2^-58)
    double ts = F[29].hi, tc = -F[28].hi;
    GFX_SERIES
    for (int k = 13; k >= 8; --k) ts = x_fma(ts, t.hi, (k & 1) ? -F[2 * k + 1].hi : F[2 * k + 1].hi);
    tc = x_fma(tc, t.hi, F[26].hi);                                            // k = 13
    GFX_SERIES
    for (int k = 12; k >= 9; --k) tc = x_fma(tc, t.hi, (k & 1) ? -F[2 * k].hi : F[2 * k].hi);
    dd s = dd_from(ts), c = dd_from(tc);
    GFX_SERIES
    for (int k = 8; k >= 0; --k) {
        const ddc fc = F[2 * k];
        const dd ck = {(k & 1) ? -fc.hi : fc.hi, (k & 1) ? -fc.lo : fc.lo};
        c = dd_horner(c, t, ck);
        if (k < 8) {
            const ddc fs = F[2 * k + 1];
            const dd sk = {(k & 1) ? -fs.hi : fs.hi, (k & 1) ? -fs.lo : fs.lo};
            s = dd_horner(s, t, sk);
        }
    }
    s = dd_mul(s, r);
    const int q = (int)((long long)kf & 3);
    const dd ns = dd_neg(s), nc = dd_neg(c);
    sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? ns : nc));
    cs = q == 0 ? c : (q == 1 ? ns : (q == 2 ? nc : s));
}

// sin x for 0 <= |x| <= 0.81 without argument reduction (dd_asin_small: x = asin of at most 0.72): the odd series alone, half of
// dd_sincos's work and no quadrant to branch on.  x^31 / 31! < 2^-111 x at 0.81; the orders from x^17 on are below 2^-53 of the
// leading term and are summed in fp64 (rounding errors below 2^-106).
GFX_HD GFX_BIG dd dd_sin_small(dd x)
{
    static const ddc F[30] = {GFX_INV_FACT_TABLE};
    const dd t = dd_mul(x, x);
    double ts = F[29].hi;
    GFX_SERIES
    for (int k = 13; k >= 8; --k) ts = x_fma(ts, t.hi, (k & 1) ? -F[2 * k + 1].hi : F[2 * k + 1].hi);
    dd v = dd_from(ts);
    GFX_SERIES
    for (int k = 7; k >= 0; --k) {
        const ddc fs = F[2 * k + 1];
        const dd sk = {(k & 1) ? -fs.hi : fs.hi, (k & 1) ? -fs.lo : fs.lo};
        v = dd_horner(v, t, sk);
    }
    return dd_mul(v, x);
}

// asin for |t| <= ~0.75: one Newton step on sin from the fp64 value (error e -> e^2 tan / 2).  The step needs sin(th0) to full
// double-double accuracy but cos(th0) only as the divisor of a correction of relative size 1e-16: fp64, from the sine.
GFX_HD inline dd dd_asin_small(dd t)
{
    if (t.hi == 0.0) return t;
    const double th0 = asin(t.hi);
    const dd s = dd_sin_small(dd_from(th0));                          // th0 <= asin(0.75) = 0.85 at the very most
    const double c = sqrt(x_fma(-s.hi, s.hi, 1.0));                   // |s| <= 0.75: no cancellation to speak of
    const dd d = dd_sub(s, t);
    dd r;
    two_sum(th0, -(d.hi / c), r.hi, r.lo);
    return r;
}

GFX_HD inline dd dd_pio2() { dd r = {GFX_PIO2_1, GFX_PIO2_2}; return r; }
GFX_HD inline dd dd_pi() { dd r = {2.0 * GFX_PIO2_1, 2.0 * GFX_PIO2_2}; return r; }

// sqrt((1 - a)(1 + a)) for 0 <= a <= 1
GFX_HD inline dd dd_cofunc(dd a)
{
    return dd_sqrt(dd_mul(dd_add_d(1.0, dd_neg(a)), dd_add_d(1.0, a)));
}

GFX_HD inline dd dd_asin(dd s)      // s in [-1, 1]
{
    const bool neg = s.hi < 0.0;
    const dd a = neg ? dd_neg(s) : s;
    dd r;
    if (a.hi <= 0.72) r = dd_asin_small(a);
    else r = dd_sub(dd_pio2(), dd_asin_small(dd_cofunc(a)));
    return neg ? dd_neg(r) : r;
}

GFX_HD inline dd dd_acos(dd x)      // x in [-1, 1]
{
    const bool neg = x.hi < 0.0;
    const dd a = neg ? dd_neg(x) : x;
    if (a.hi <= 0.72) return dd_sub(dd_pio2(), dd_asin_small(x));
    const dd phi = dd_asin_small(dd_cofunc(a));
    return neg ? dd_sub(dd_pi(), phi) : phi;
}

// 10^x for an fp64 x, rounded to fp64 from a ~2^-95 evaluation: the correctly rounded value except within 2^-42 ulp of a
// midpoint.  fr.py:380 `np.power(10., logLam)` is an fp64 operation whose result every later step inherits; libm's pow
// is correctly rounded on all but ~0.1 % of arguments, so this reproduces the value a libm-backed numpy feeds the chain
// (a numpy with vectorised pow differs from libm's -- and from this -- by one ulp on ~5 % of arguments; tests/golden G17).
GFX_HD inline double cr_pow10(double x)
{
    if (!(fabs(x) < 300.0)) return pow(10.0, x);
    static const ddc F[30] = {GFX_INV_FACT_TABLE};
    const dd L = {3.321928094887362, 1.661617516973592e-16};          // log2(10)
    const dd LN2 = {0.6931471805599453, 2.3190468138462996e-17};
    const dd y = dd_mul_d(L, x);
    const double n = nearbyint(y.hi);
    const dd f = dd_add(y, dd_from(-n));                              // [-1/2, 1/2]
    const dd z = dd_mul(f, LN2);                                      // |z| <= 0.3466
    // exp z = sum z^k / k!, k = 0 .. 23 (z^24 / 24! < 2^-115): Horner's rule, the orders from 12 on (z^12 / 12! < 2^-47) in fp64
    double tail = F[23].hi;
    GFX_ROLLED
    for (int k = 22; k >= 12; --k) tail = x_fma(tail, z.hi, F[k].hi);
    dd sum = dd_from(tail);
    GFX_ROLLED
    for (int k = 11; k >= 0; --k) {
        const ddc fk = F[k];
        const dd ck = {fk.hi, fk.lo};
        sum = dd_horner(sum, z, ck);
    }
    return ldexp(sum.hi, (int)n);                                     // sum.hi = fl(sum): the nearest double
}

GFX_HD inline x87 x_asin(x87 a) { return round64(dd_asin(as_dd(a))); }
GFX_HD inline x87 x_acos(x87 a) { return round64(dd_acos(as_dd(a))); }
GFX_HD inline void x_sincos(x87 a, x87& s, x87& c)
{
    dd ds, dc;
    dd_sincos(as_dd(a), ds, dc);
    s = round64(ds);
    c = round64(dc);
}
// hypotl: sqrt(x^2 + y^2) to ~2^-100, rounded (glibc's is faithful; it only feeds well-conditioned quantities here)
GFX_HD inline x87 x_hypot(x87 a, x87 b)
{
    const dd da = as_dd(a), db = as_dd(b);
#if defined(GFX87_ROUND64_R3)
    return round64(dd_sqrt(dd_add(dd_mul(da, da), dd_mul(db, db))));
#else
    // the two squares are non-negative: their sum cancels nothing, the tails go in plain fp64 (11 operations instead of dd_add's 20)
    const dd p = dd_mul(da, da), q = dd_mul(db, db);
    double s, e;
    two_sum(p.hi, q.hi, s, e);
    e += p.lo + q.lo;
    return round64(dd_sqrt(dd_norm(s, e)));
#endif
}

// 1/2, 1/3, 1/9, 1/54 rounded to a 64-bit significand: what fr.py:205-209's `1./2`, `/ 3.`, `(1./9)`, `(1./54)` are in np.float128
// -- four divisions per energy bin that are the same every time (checked against x_div: tests/test_x87_emulation.py)
#define GFX_X87_HALF x87{0.5, 0.0}
#define GFX_X87_THIRD x87{0.3333333333333333, 1.8512752095189988e-17}
#define GFX_X87_NINTH x87{0.1111111111111111, 6.1663998560113065e-18}
#define GFX_X87_54TH x87{0.018518518518518517, 1.0282979979667206e-18}

// ---- complex arithmetic the way numpy does it for np.complex256 ------------------------------------------
GFX_HD inline cx87 c_make(x87 re, x87 im) { cx87 r = {re, im}; return r; }
GFX_HD inline cx87 c_zero() { cx87 r = {{0.0, 0.0}, {0.0, 0.0}}; return r; }
GFX_HD GFX_CADD cx87 c_add(cx87 a, cx87 b) { return c_make(x_add(a.re, b.re), x_add(a.im, b.im)); }
GFX_HD GFX_CADD cx87 c_sub(cx87 a, cx87 b) { return c_make(x_sub(a.re, b.re), x_sub(a.im, b.im)); }
GFX_HD inline cx87 c_neg(cx87 a) { return c_make(x_neg(a.re), x_neg(a.im)); }
GFX_HD inline cx87 c_conj(cx87 a) { return c_make(a.re, x_neg(a.im)); }
GFX_HD GFX_CMUL cx87 c_mul(cx87 a, cx87 b)
{
    return c_make(x_sub(x_mul(a.re, b.re), x_mul(a.im, b.im)), x_add(x_mul(a.re, b.im), x_mul(a.im, b.re)));
}
// Three places where the reference's operations have an EXACT outcome that needs no emulated arithmetic (round 4; the values are the
// chain's, bit for bit -- the three-lane chain uses them; the serial chain below (the statement the host build checks against the x87
// unit) and the nine-lane chain keep the long way unless GFX87_SHORT_A / _B / _C are defined, and the tests compare both ways):
//   z * z: numpy's (ac - bd, ad + bc) with c = a, d = b -- the two cross products are the same number and their sum is its double;
//   2 * z, -2 * z, z / 2 (fr.py:205's `1./2`): a doubling or halving is exact in any binary format;
//   z * conj(z): (a a - b (-b), a (-b) + b a) = (RN(a^2) + RN(b^2), -p + p = 0), so the diagonal of X X^+ is a sum of squares with an
//   imaginary part of exactly zero, and its modulus hypotl(s, 0) is s.
GFX_HD inline cx87 c_sqr(cx87 z)
{
    const x87 p = x_mul(z.re, z.im);
    return c_make(x_sub(x_mul(z.re, z.re), x_mul(z.im, z.im)), x_scale2(p, 2.0));
}
GFX_HD inline cx87 c_times2(cx87 z, double two) { return c_make(x_scale2(z.re, two), x_scale2(z.im, two)); }       // two = 2.0, -2.0 or 0.5
GFX_HD inline x87 c_norm2(cx87 z) { return x_add(x_mul(z.re, z.re), x_mul(z.im, z.im)); }                        // Re(z conj z); Im is exactly 0

// real * complex: numpy promotes the real to (r, 0); the products with the zero are exact, so this is a scaling
GFX_HD GFX_CADD cx87 c_scale(x87 r, cx87 a) { return c_make(x_mul(r, a.re), x_mul(r, a.im)); }
// numpy's complex division (Smith)
GFX_HD inline cx87 c_div(cx87 a, cx87 b)
{
    const x87 br = x_abs(b.re), bi = x_abs(b.im);
    const x87 one = x_from(1.0);
    if (x_ge(br, bi)) {
        if (x_is_zero(br) && x_is_zero(bi)) return c_make(x_div(a.re, br), x_div(a.im, br));
        const x87 rat = x_div(b.im, b.re);
        const x87 scl = x_div(one, x_add(b.re, x_mul(b.im, rat)));
        return c_make(x_mul(x_add(a.re, x_mul(a.im, rat)), scl), x_mul(x_sub(a.im, x_mul(a.re, rat)), scl));
    }
    const x87 rat = x_div(b.re, b.im);
    const x87 scl = x_div(one, x_add(b.im, x_mul(b.re, rat)));
    return c_make(x_mul(x_add(x_mul(a.re, rat), a.im), scl), x_mul(x_sub(x_mul(a.im, rat), a.re), scl));
}
// complex / real scalar = complex / (d, 0) in numpy: x * (1 / d)
GFX_HD inline cx87 c_div_real(cx87 a, x87 d)
{
    const x87 scl = x_div(x_from(1.0), d);
    return c_make(x_mul(a.re, scl), x_mul(a.im, scl));
}
GFX_HD inline x87 c_abs(cx87 a) { return x_hypot(a.re, a.im); }

// csqrtl for Re z > 0 (glibc's general branch: d = hypot, r = sqrt((d + re)/2), s = (im / r)/2)
GFX_HD inline cx87 c_sqrt_pos(cx87 z)
{
    const x87 d = x_hypot(z.re, z.im);
    const x87 r = x_sqrt(x_scale2(x_add(d, z.re), 0.5));
    const x87 s = x_scale2(x_div(z.im, r), 0.5);
    return c_make(r, s);
}

// cacosl(x + iy) next to the real axis (|y| << 1): acos(x) - i y / sqrt(1 - x^2); beyond the branch points (rounding
// can push the cubic's argument a hair past +-1) the real part sticks to 0 / pi and the imaginary part is -+ acosh|x|
GFX_HD inline cx87 c_acos_near_real(cx87 z)
{
    const dd x = as_dd(z.re);
    const dd ax = x.hi < 0.0 ? dd_neg(x) : x;
    const dd one = dd_from(1.0);
    const dd om = dd_add_d(1.0, dd_neg(ax));                          // 1 - |x|
    if (om.hi > 0.0) {
        const x87 re = round64(dd_acos(x));
        const dd den = dd_sqrt(dd_mul(om, dd_add_d(1.0, ax)));
        const x87 im = x_neg(round64(dd_div(as_dd(z.im), den)));
        return c_make(re, im);
    }
    // |x| >= 1: acosh(|x|) ~ sqrt(2 (|x| - 1)) (1 - (|x| - 1)/12)
    const dd t = dd_neg(om);
    const dd ah = dd_mul(dd_sqrt(dd_mul_d(t, 2.0)), dd_sub(one, dd_div_d(t, 12.0)));
    const x87 re = x.hi < 0.0 ? round64(dd_pi()) : x_from(0.0);
    x87 im = round64(ah);
    if (!(z.im.hi < 0.0)) im = x_neg(im);                            // Im acos(x + iy) has the sign of -y
    return c_make(re, im);
}

// ccosl(u + iv) next to the real axis: (cos u cosh v, -sin u sinh v) with cosh v = 1, sinh v = v to the last bit for
// |v| < 2^-32 (glibc returns exactly that)
GFX_HD inline cx87 c_cos_near_real(cx87 z)
{
    x87 s, c;
    x_sincos(z.re, s, c);
    return c_make(c, x_neg(x_mul(s, z.im)));
}

// ---- the chain ---------------------------------------------------------------------------------------------
// golemflavor/fr.py:116-162 angles_to_u, with the structural zeros and ones of p1, p2, p3 folded (a product with an
// exact 0 or 1 and a sum with an exact 0 round to themselves)
GFX_HD inline void angles_to_u(const double ang[4], cx87 u[3][3])
{
    const x87 s12_2 = x_from(ang[0]), c13_4 = x_from(ang[1]), s23_2 = x_from(ang[2]), dcp = x_from(ang[3]);
    const x87 c13_2 = x_sqrt(c13_4);                                  // fr.py:141
    const x87 t12 = x_asin(x_sqrt(s12_2));                            // fr.py:145-147
    const x87 t13 = x_acos(x_sqrt(c13_2));
    const x87 t23 = x_asin(x_sqrt(s23_2));
    x87 c12, s12, c13, s13, c23, s23, cd, sd;
    x_sincos(t12, s12, c12);                                          // fr.py:149-154
    x_sincos(t13, s13, c13);
    x_sincos(t23, s23, c23);
    x_sincos(dcp, sd, cd);                                            // exp(+-i dcp) = (cos, +-sin)
    const cx87 em = c_make(cd, x_neg(sd)), ep = c_make(cd, sd);
    const cx87 s13em = c_scale(s13, em);                              // p2[0][2]
    const cx87 ms13ep = c_scale(x_neg(s13), ep);                      // p2[2][0]
    const x87 zero = x_from(0.0);
    // T = p1 . p2
    cx87 T[3][3];
    T[0][0] = c_make(c13, zero);  T[0][1] = c_zero();                  T[0][2] = s13em;
    T[1][0] = c_scale(s23, ms13ep); T[1][1] = c_make(c23, zero);      T[1][2] = c_make(x_mul(s23, c13), zero);
    T[2][0] = c_scale(c23, ms13ep); T[2][1] = c_make(x_neg(s23), zero); T[2][2] = c_make(x_mul(c23, c13), zero);
    // u = T . p3
    const x87 ms12 = x_neg(s12);
    GFX_ROLLED
    for (int i = 0; i < 3; ++i) {
        u[i][0] = c_add(c_scale(c12, T[i][0]), c_scale(ms12, T[i][1]));
        u[i][1] = c_add(c_scale(s12, T[i][0]), c_scale(c12, T[i][1]));
        u[i][2] = T[i][2];
    }
}

// U diag(0, w1, w2) U^+ the way fr.py:383-386 / :391-393 evaluate it: np.dot(U, np.dot(diag, U^+))
GFX_HD inline void sandwich(const cx87 u[3][3], double w1, double w2, cx87 h[3][3])
{
    const x87 xw1 = x_from(w1), xw2 = x_from(w2);
    GFX_ROLLED
    for (int j = 0; j < 3; ++j) {
        const cx87 t1 = c_scale(xw1, c_conj(u[j][1]));                // (diag . U^+)[1][j]
        const cx87 t2 = c_scale(xw2, c_conj(u[j][2]));
        GFX_ROLLED
        for (int i = 0; i < 3; ++i) h[i][j] = c_add(c_mul(u[i][1], t1), c_mul(u[i][2], t2));
    }
}

// fr.py:170-237 cardano_eqn followed by fr.py:489-494: returns max(|tr f - 3|, |sum f - 3|); NaN -> +inf
GFX_HD inline double cardano_residual(const cx87 h[3][3])
{
    const x87 two = x_from(2.0), three = x_from(3.0), nine = x_from(9.0), n27 = x_from(27.0);
    const cx87 tr = c_add(c_add(h[0][0], h[1][1]), h[2][2]);
    cx87 tr2 = c_zero();
    {
        cx87 d[3];
        GFX_ROLLED
        for (int i = 0; i < 3; ++i) {
            cx87 s = c_mul(h[i][0], h[0][i]);
            s = c_add(s, c_mul(h[i][1], h[1][i]));
            s = c_add(s, c_mul(h[i][2], h[2][i]));
            d[i] = s;
        }
        tr2 = c_add(c_add(d[0], d[1]), d[2]);
    }
    const cx87 a = c_neg(tr);                                                           // fr.py:204
#if defined(GFX87_SHORT_A)
    const cx87 b = c_times2(c_sub(c_sqr(tr), tr2), 0.5);                                // fr.py:205 (halving is exact)
#else
    const cx87 b = c_scale(GFX_X87_HALF, c_sub(c_mul(tr, tr), tr2));                    // fr.py:205
#endif
    const cx87 det = c_add(c_sub(c_mul(h[0][0], c_sub(c_mul(h[1][1], h[2][2]), c_mul(h[2][1], h[1][2]))),
                                 c_mul(h[1][0], c_sub(c_mul(h[0][1], h[2][2]), c_mul(h[2][1], h[0][2])))),
                           c_mul(h[2][0], c_sub(c_mul(h[0][1], h[1][2]), c_mul(h[1][1], h[0][2]))));   // fr.py:77-79
    const cx87 c = c_neg(det);                                                          // fr.py:206
#if defined(GFX87_SHORT_A)
    const cx87 a2 = c_sqr(a);
#else
    const cx87 a2 = c_mul(a, a);
#endif
    const cx87 Q = c_scale(GFX_X87_NINTH, c_sub(a2, c_scale(three, b)));                // fr.py:208
#if defined(GFX87_SHORT_B)
    const cx87 R = c_scale(GFX_X87_54TH,
                           c_add(c_sub(c_times2(c_mul(a, a2), 2.0), c_mul(c_scale(nine, a), b)), c_scale(n27, c)));   // fr.py:209
#else
    const cx87 R = c_scale(GFX_X87_54TH,
                           c_add(c_sub(c_scale(two, c_mul(a, a2)), c_mul(c_scale(nine, a), b)), c_scale(n27, c)));   // fr.py:209
#endif
#if defined(GFX87_SHORT_A)
    const cx87 theta = c_acos_near_real(c_div(R, c_sqrt_pos(c_mul(Q, c_sqr(Q)))));      // fr.py:210
#else
    const cx87 theta = c_acos_near_real(c_div(R, c_sqrt_pos(c_mul(Q, c_mul(Q, Q)))));   // fr.py:210
#endif
    const cx87 sq = c_sqrt_pos(Q);
#if defined(GFX87_SHORT_B)
    const cx87 m2sq = c_times2(sq, -2.0);
#else
    const cx87 m2sq = c_scale(x_neg(two), sq);
#endif
    const cx87 third_a = c_scale(GFX_X87_THIRD, a);
    const x87 pi = {3.141592653589793, 1.22514845490862e-16};                           // np.arccos(np.float128(-1)), fr.py:24
    const x87 twopi = x_mul(two, pi);
    cx87 E[3];
    E[0] = c_sub(c_mul(m2sq, c_cos_near_real(c_div_real(theta, three))), third_a);                                      // fr.py:212
    E[1] = c_sub(c_mul(m2sq, c_cos_near_real(c_div_real(c_make(x_sub(theta.re, twopi), theta.im), three))), third_a);   // fr.py:213
    E[2] = c_sub(c_mul(m2sq, c_cos_near_real(c_div_real(c_make(x_add(theta.re, twopi), theta.im), three))), third_a);   // fr.py:214
    cx87 x[3][3];
    const cx87 h10h02 = c_mul(h[1][0], h[0][2]), h21h10 = c_mul(h[2][1], h[1][0]), h12h20 = c_mul(h[1][2], h[2][0]);
    GFX_ROLLED
    for (int k = 0; k < 3; ++k) {
        const cx87 A = c_sub(c_mul(h[1][2], c_sub(h[0][0], E[k])), h10h02);             // fr.py:216-218
        const cx87 B = c_sub(c_mul(h[2][0], c_sub(h[1][1], E[k])), h21h10);             // fr.py:220-222
        const cx87 C = c_sub(c_mul(h[1][0], c_sub(h[2][2], E[k])), h12h20);             // fr.py:224-226
        const cx87 AB = c_mul(A, B), AC = c_mul(A, C), BC = c_mul(B, C);
        const x87 ab = c_abs(AB), ac = c_abs(AC), bc = c_abs(BC);
        const x87 N = x_sqrt(x_add(x_add(x_mul(ab, ab), x_mul(ac, ac)), x_mul(bc, bc)));   // fr.py:228-230
        x[0][k] = c_div_real(c_mul(c_conj(B), C), N);                                   // fr.py:232-236
        x[1][k] = c_div_real(AC, N);
        x[2][k] = c_div_real(AB, N);
    }
    // fr.py:489: f = |x x^+|; |p_ji| = |p_ij| exactly (the same products, the sums negated)
    x87 f[3][3];
    GFX_ROLLED
    for (int i = 0; i < 3; ++i)
        GFX_ROLLED
        for (int j = i; j < 3; ++j) {
#if defined(GFX87_SHORT_C)
            if (i == j) {
                x87 d = c_norm2(x[i][0]);
                d = x_add(d, c_norm2(x[i][1]));
                d = x_add(d, c_norm2(x[i][2]));
                f[i][j] = d;
                continue;
            }
#endif
            cx87 s = c_mul(x[i][0], c_conj(x[j][0]));
            s = c_add(s, c_mul(x[i][1], c_conj(x[j][1])));
            s = c_add(s, c_mul(x[i][2], c_conj(x[j][2])));
            f[i][j] = c_abs(s);
            f[j][i] = f[i][j];
        }
    const x87 trf = x_add(x_add(f[0][0], f[1][1]), f[2][2]);
    const x87 sum = x_add(x_add(x_add(x_add(f[0][0], f[0][1]), x_add(f[0][2], f[1][0])),
                                x_add(x_add(f[1][1], f[1][2]), x_add(f[2][0], f[2][1]))), f[2][2]);
    const double rt = fabs(x_to_double(x_sub(trf, three))), rs = fabs(x_to_double(x_sub(sum, three)));
    double r = rt > rs ? rt : rs;
    if (!(r == r) || !(rt == rt) || !(rs == rs)) r = INFINITY;
    return r;
}

// One energy bin of fr.py:380-399: ham = pre * Hsm + epow * Hnp, then the residual.  `pre`, `epow` as the reference
// forms them in fp64 (1 / (2 E), E ** (d - 3)).  The sum is rescaled by an exact power of two to magnitude one.
GFX_HD inline double bin_residual(const cx87 hsm[3][3], const cx87 hnp[3][3], double pre, double epow)
{
    double big = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double a = fabs(pre * hsm[i][i].re.hi), b = fabs(epow * hnp[i][i].re.hi);
        big = a > big ? a : big;
        big = b > big ? b : big;
    }
    double p2 = 1.0;
    if (big > 0.0 && big < 1.7976931348623157e308) {
        const int e = (int)((x_bits(big) >> 52) & 0x7ff) - 1023;
        int k = -e;
        k = k > 1000 ? 1000 : (k < -1000 ? -1000 : k);
        p2 = x_from_bits((int64_t)(k + 1023) << 52);
    }
    // scaling the two fp64 prefactors by 2^k scales every entry of ham by exactly 2^k
    const x87 xp = x_from(pre * p2), xe = x_from(epow * p2);
    cx87 h[3][3];
    GFX_ROLLED
    for (int i = 0; i < 3; ++i)
        GFX_ROLLED
        for (int j = 0; j < 3; ++j) h[i][j] = c_add(c_scale(xp, hsm[i][j]), c_scale(xe, hnp[i][j]));   // fr.py:386, 394-395
    return cardano_residual(h);
}

}  // namespace gfx87
