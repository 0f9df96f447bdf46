// x87_host.cpp -- TEST INFRASTRUCTURE: the emulated x87 arithmetic of golemflavor_amd/csrc/gf_x87.hpp compiled for the
// host and compared with the CPU's own x87 unit (`long double` on x86-64) and with libm's long-double functions.
// Built by tests/test_x87_emulation.py with g++; nothing in the product links it.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../golemflavor_amd/csrc/gf_x87.hpp"

using namespace gfx87;
typedef long double ld;

static inline x87 from_ld(ld v)
{
    x87 r;
    r.hi = (double)v;
    r.lo = (double)(v - (ld)r.hi);
    return r;
}
static inline ld to_ld(x87 a) { return (ld)a.hi + (ld)a.lo; }
static inline bool same(ld a, ld b) { return a == b || (a != a && b != b); }

static uint64_t sm64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a long double with a random 64-bit significand and an exponent in [-span, span]
static ld rnd_ld(uint64_t& s, int span)
{
    const uint64_t m = sm64(s) | 0x8000000000000000ull;
    const int e = (int)(sm64(s) % (2 * span + 1)) - span;
    ld v = ldexpl((ld)m, e - 63);
    return (sm64(s) & 1) ? -v : v;
}

extern "C" {

// counts[0..4] = mismatches of add, sub, mul, div, sqrt over n random operand pairs (+ structured cases)
int x87t_arith(uint64_t seed, int n, int64_t* counts)
{
    for (int i = 0; i < 5; ++i) counts[i] = 0;
    uint64_t s = seed;
    for (int it = 0; it < n; ++it) {
        ld a, b;
        const int kind = it % 8;
        if (kind == 0) { a = rnd_ld(s, 40); b = a * (1.0L + ldexpl((ld)(sm64(s) % 4096), -60)); b = -b; }        // cancellation
        else if (kind == 1) { a = rnd_ld(s, 2); b = (ld)(1 + sm64(s) % 60); }                                    // small integer factor: ties
        else if (kind == 2) { a = rnd_ld(s, 5); b = rnd_ld(s, 5) * ldexpl(1.0L, -(int)(sm64(s) % 80)); }          // far-apart exponents
        else if (kind == 3) { a = (ld)(double)rnd_ld(s, 30); b = (ld)(double)rnd_ld(s, 30); }                     // fp64 operands
        else { a = rnd_ld(s, 60); b = rnd_ld(s, 60); }
        const x87 xa = from_ld(a), xb = from_ld(b);
        if (!same(to_ld(x_add(xa, xb)), a + b)) counts[0]++;
        if (!same(to_ld(x_sub(xa, xb)), a - b)) counts[1]++;
        if (!same(to_ld(x_mul(xa, xb)), a * b)) counts[2]++;
        if (!same(to_ld(x_div(xa, xb)), a / b)) counts[3]++;
        if (!same(to_ld(x_sqrt(x_abs(xa))), sqrtl(fabsl(a)))) counts[4]++;
    }
    return 0;
}

// A stream of + - * results folded into one 64-bit digest, and the number of them that differ from the x87 unit's: the two forms
// of round64 / x_add (round 4's and -DGFX87_ROUND64_R3) must give the same digest.  Operands: as x87t_arith's, plus the cases the
// short sum has to get right -- exponents 42 .. 64 apart with exact half-ulp ties and near-ties, sums that cancel into the binade below
// a power of two, short significands, results that are powers of two.
static inline uint64_t fold(uint64_t h, x87 r)          // of the VALUE (the pair need not be the canonical split of it)
{
    ld v = to_ld(r);
    if (v == 0.0L) v = 0.0L;                            // the sign of a zero is not part of the digest
    uint64_t m = 0, e = 0;
    memcpy(&m, &v, 8); memcpy(&e, (const char*)&v + 8, 2);
    h = (h ^ m) * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    h = (h ^ e) * 0xBF58476D1CE4E5B9ull; h ^= h >> 31;
    return h;
}
uint64_t x87t_digest(uint64_t seed, int64_t n, int64_t* wrong)
{
    uint64_t s = seed, h = 0;
    *wrong = 0;
    for (int64_t it = 0; it < n; ++it) {
        ld a, b;
        const int kind = (int)(it % 12);
        if (kind == 0) { a = rnd_ld(s, 40); b = a * (1.0L + ldexpl((ld)(sm64(s) % 4096), -60)); b = -b; }
        else if (kind == 1) { a = rnd_ld(s, 2); b = (ld)(1 + sm64(s) % 60); }
        else if (kind == 2) { a = rnd_ld(s, 5); b = rnd_ld(s, 5) * ldexpl(1.0L, -(int)(sm64(s) % 80)); }
        else if (kind == 3) { a = (ld)(double)rnd_ld(s, 30); b = (ld)(double)rnd_ld(s, 30); }
        else if (kind == 4) {                                            // b = (odd k) half-ulps of a, 42 .. 64 binades down: exact ties
            a = rnd_ld(s, 20);
            int ea; frexpl(a, &ea);
            const int d = 42 + (int)(sm64(s) % 23);
            b = ldexpl((ld)(2 * (sm64(s) % 2048) + 1), ea - 1 - d - 11);
            if (sm64(s) & 1) b = -b;
        }
        else if (kind == 5) {                                            // half an ulp of a, 2^-36 of itself off the tie (a tie missed by
            a = rnd_ld(s, 20);                                           // 2^-64 of b is beyond ANY double-double: the header's 2^-104)
            int ea; frexpl(a, &ea);
            b = ldexpl(1.0L, ea - 65);
            b += (sm64(s) & 1) ? ldexpl(b, -36) : -ldexpl(b, -36);
            if (sm64(s) & 1) b = -b;
        }
        else if (kind == 6) { a = ldexpl(1.0L, (int)(sm64(s) % 41) - 20); b = -rnd_ld(s, 30) * ldexpl(1.0L, -(int)(40 + sm64(s) % 50)); }   // under a power of two
        else if (kind == 7) { a = (ld)(sm64(s) % 4096) * ldexpl(1.0L, (int)(sm64(s) % 9) - 4); b = (ld)(sm64(s) % 4096) - 2048; }        // short, zeros
        else if (kind == 8) { a = rnd_ld(s, 3); b = ldexpl(1.0L, (int)(sm64(s) % 7) - 3) - a; }                                            // a + b = a power of two
        else { a = rnd_ld(s, 60); b = rnd_ld(s, 60); }
        const x87 xa = from_ld(a), xb = from_ld(b);
        const x87 r0 = x_add(xa, xb), r1 = x_sub(xa, xb), r2 = x_mul(xa, xb);
        h = fold(fold(fold(h, r0), r1), r2);
        *wrong += !same(to_ld(r0), a + b) + !same(to_ld(r1), a - b) + !same(to_ld(r2), a * b);
        if (b != 0.0L) {                                                     // round 4 also shortened / and sqrt (hypot: faithful in libm, so digest only)
            const x87 r3 = x_div(xa, xb), r4 = x_sqrt(x_abs(xa)), r5 = x_hypot(xa, xb);
            h = fold(fold(fold(h, r3), r4), r5);
            *wrong += !same(to_ld(r3), a / b) + !same(to_ld(r4), sqrtl(fabsl(a)));
        }
    }
    return h;
}

// functions: exact[k] = calls whose emulated result equals libm's bit for bit, off1[k] = off by one ulp, worse[k] = more;
// k = 0 asinl, 1 acosl, 2 sinl, 3 cosl, 4 hypotl
int x87t_funcs(uint64_t seed, int n, int64_t* exact, int64_t* off1, int64_t* worse)
{
    for (int i = 0; i < 5; ++i) exact[i] = off1[i] = worse[i] = 0;
    uint64_t s = seed;
    auto tally = [&](int k, ld got, ld want) {
        if (same(got, want)) { exact[k]++; return; }
        const ld u = nextafterl(want, INFINITY) - want;
        if (fabsl(got - want) <= 1.0000001L * fabsl(u)) off1[k]++;
        else worse[k]++;
    };
    for (int it = 0; it < n; ++it) {
        const ld u = ldexpl((ld)(sm64(s) >> 1), -63);                    // [0, 1)
        ld x = (it % 4 == 0) ? sqrtl((ld)(double)u) : (it % 4 == 1 ? sqrtl(sqrtl((ld)(double)u)) : 2 * u - 1);
        tally(0, to_ld(x_asin(from_ld(x))), asinl(x));
        tally(1, to_ld(x_acos(from_ld(x))), acosl(x));
        const ld t = (it % 2) ? (ld)(double)(6.283185307179586L * u) : (4 * u - 1) * 3.2L;
        x87 sn, cs;
        x_sincos(from_ld(t), sn, cs);
        tally(2, to_ld(sn), sinl(t));
        tally(3, to_ld(cs), cosl(t));
        const ld a = rnd_ld(s, 10), b = rnd_ld(s, 10);
        tally(4, to_ld(x_hypot(from_ld(a), from_ld(b))), hypotl(a, b));
    }
    return 0;
}

// the emulated functions' results as (hi, lo) pairs, for comparison with correctly rounded values computed elsewhere
// (tests/test_x87_emulation.py: mpmath, rounded to 64 bits): out = sin hi, sin lo, cos hi, cos lo, asin hi, asin lo, acos hi, acos lo
void x87t_eval(double hi, double lo, double* out)
{
    const x87 a = {hi, lo};
    x87 sn, cs;
    x_sincos(a, sn, cs);
    out[0] = sn.hi; out[1] = sn.lo; out[2] = cs.hi; out[3] = cs.lo;
    const bool in = fabs(hi) <= 1.0;
    const x87 as = in ? x_asin(a) : x_from(0.0), ac = in ? x_acos(a) : x_from(0.0);
    out[4] = as.hi; out[5] = as.lo; out[6] = ac.hi; out[7] = ac.lo;
}
double x87t_pow10_value(double x) { return cr_pow10(x); }
// the reciprocal constants of the header against the divisions they replace: number of mismatches
int x87t_consts(void)
{
    const x87 one = x_from(1.0);
    const x87 want[4] = {x_div(one, x_from(2.0)), x_div(one, x_from(3.0)), x_div(one, x_from(9.0)), x_div(one, x_from(54.0))};
    const x87 have[4] = {GFX_X87_HALF, GFX_X87_THIRD, GFX_X87_NINTH, GFX_X87_54TH};
    int bad = 0;
    for (int i = 0; i < 4; ++i) bad += !(want[i].hi == have[i].hi && want[i].lo == have[i].lo) || !same(to_ld(have[i]), 1.0L / (ld)(i == 0 ? 2 : i == 1 ? 3 : i == 2 ? 9 : 54));
    return bad;
}

// 10^x: calls (out of n, x uniform in [lo, hi]) on which cr_pow10 and libm's pow disagree
int x87t_pow10(uint64_t seed, int n, double lo, double hi)
{
    uint64_t s = seed;
    int bad = 0;
    for (int it = 0; it < n; ++it) {
        const double x = lo + (hi - lo) * (double)(sm64(s) >> 11) * 0x1p-53;
        if (cr_pow10(x) != pow(10.0, x)) ++bad;
    }
    return bad;
}

// angles_to_u (fr.py:116-162): emulated matrix -> long double pairs [18]
void x87t_angles_to_u(const double ang[4], ld* out)
{
    cx87 u[3][3];
    angles_to_u(ang, u);
    for (int i = 0; i < 9; ++i) { out[2 * i] = to_ld(u[i / 3][i % 3].re); out[2 * i + 1] = to_ld(u[i / 3][i % 3].im); }
}

// One (walker, bin): the emulated residual.  smu / npu: NULL -> emulated angles_to_u of the angles, else the matrices as
// long double (re, im) pairs (what gf_model_create computes on the host for per-model constants).
double x87t_bin_residual(const double sm_ang[4], const double np_ang[4], double m21, double m3x, double sc2, double energy,
                         int dim, const ld* smu, const ld* npu)
{
    cx87 us[3][3], un[3][3], hsm[3][3], hnp[3][3];
    if (smu) { for (int i = 0; i < 9; ++i) us[i / 3][i % 3] = c_make(from_ld(smu[2 * i]), from_ld(smu[2 * i + 1])); }
    else angles_to_u(sm_ang, us);
    if (npu) { for (int i = 0; i < 9; ++i) un[i / 3][i % 3] = c_make(from_ld(npu[2 * i]), from_ld(npu[2 * i + 1])); }
    else angles_to_u(np_ang, un);
    sandwich(us, m21, m3x, hsm);
    sandwich(un, sc2 / 100., sc2, hnp);
    return bin_residual(hsm, hnp, 1. / (2 * energy), pow(energy, (double)(dim - 3)));
}

}  // extern "C"
