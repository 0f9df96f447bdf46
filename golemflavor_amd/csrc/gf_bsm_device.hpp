// gf_bsm_device.hpp -- device functions of the BSM (texture) branch, shared by gf_bsm.hip (batch
// evaluation) and gf_sampler.hip (device-resident stretch move).  See gf_bsm.hip for the numerics notes.
#pragma once
#include "gf_device.hpp"

namespace gfdev {

constexpr int TEX_NONE = 4;
constexpr double UNI_THRESHOLD = 1e-7 * 2048.0;

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
static __device__ __attribute__((noinline)) void mixing_cols12(double s12_2, double c13_4, double s23_2, double dcp,
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

// One energy bin: eigenvalues of the trace-normalised H by the trigonometric cubic solution
// (fr.py:204-214 with a = -1), moduli by the eigenvector-eigenvalue identity.  Optionally the
// reference's eigenvector form for the unitarity status.
template <bool CHECK_UNI>
__device__ __forceinline__ void bin_moduli(const Herm3& h_in, double p[3][3], double& residual)
{
    const double s = fast_rcp((h_in.d0 + h_in.d1) + h_in.d2);
    const double d0 = h_in.d0 * s, d1 = h_in.d1 * s, d2 = h_in.d2 * s;
    const double r01 = h_in.r01 * s, i01 = h_in.i01 * s;
    const double r02 = h_in.r02 * s, i02 = h_in.i02 * s;
    const double r12 = h_in.r12 * s, i12 = h_in.i12 * s;
    const double o01 = fma(r01, r01, i01 * i01);
    const double o02 = fma(r02, r02, i02 * i02);
    const double o12 = fma(r12, r12, i12 * i12);
    // b = sum of principal 2x2 minors = (tr^2 - tr H^2)/2 (fr.py:205); c = -det (fr.py:206)
    const double b = (fma(d0, d1, fma(d0, d2, d1 * d2)) - o01) - (o02 + o12);
    // Re(h01 h12 conj(h02))
    const double tr_re = fma(r01, r12, -i01 * i12), tr_im = fma(r01, i12, i01 * r12);
    const double re3 = fma(tr_re, r02, tr_im * i02);
    const double det = fma(d0 * d1, d2, 2.0 * re3) - fma(d0, o12, fma(d1, o02, d2 * o01));
    const double Q = fma(-3.0, b, 1.0) * (1.0 / 9.0);                 // (a^2 - 3b)/9, a = -1
    const double R = (fma(9.0, b, -2.0) - 27.0 * det) * (1.0 / 54.0); // (2a^3 - 9ab + 27c)/54
    const double sq = fast_sqrt(Q);
    double x = R * fast_rcp(Q * sq);
    x = fmin(1.0, fmax(-1.0, x));
    const double phi = fast_acos(x) * (1.0 / 3.0);
    double sp, cp;
    sincos_small(phi, &sp, &cp);
    const double m2 = -2.0 * sq;
    const double HS3 = 0.8660254037844386;                            // sqrt(3)/2
    double E[3];
    E[0] = fma(m2, cp, 1.0 / 3.0);                                    // fr.py:212
    E[1] = fma(m2, fma(HS3, sp, -0.5 * cp), 1.0 / 3.0);               // cos(phi - 2pi/3), fr.py:213
    E[2] = fma(m2, fma(-HS3, sp, -0.5 * cp), 1.0 / 3.0);              // cos(phi + 2pi/3), fr.py:214

    const double os0 = o01 + o02, os1 = o01 + o12, os2 = o02 + o12;
    const double dd[3] = {d0, d1, d2};
    const double os[3] = {os0, os1, os2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double inv = fast_rcp((E[i] - E[j]) * (E[i] - E[k]));
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a][i] = fma(dd[a] - E[j], dd[a] - E[k], os[a]) * inv;
    }

    if (CHECK_UNI) {
        // fr.py:216-236 in fp64, then fr.py:489-494.  h10 = conj(h01) etc.
        double f01r = 0, f01i = 0, f02r = 0, f02i = 0, f12r = 0, f12i = 0, trf = 0;
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
            trf += (fma(b2, c2, fma(a2, c2, a2 * b2))) * invS;          // = 1 up to rounding (and NaN)
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
        const double rt = fabs(trf - 3.0);
        const double rs = fabs(fma(2.0, off, trf) - 3.0);
        double r = fmax(rt, rs);
        if (rt != rt || rs != rs) r = gf_inf();                        // NaN fails the reference's test too
        residual = fmax(residual, r);
    }
}

// flux_averaged_BSMu for one walker (fr.py:403-458).  Returns the normalised composition and the worst
// unitarity residual over the bins.
template <bool CHECK_UNI>
__device__ __forceinline__ void flux_average(const GfCommon& c, const GfBsm* __restrict__ tb, const double* ttab,
                                             const double* row, double fr[3], double& residual)
{
    // SM part, per walker: U diag(0, m21, m3x) U^+ = m21 u1 u1^+ + m3x u2 u2^+   (fr.py:383-386)
    double c1r[3], c1i[3], c2r[3], c2i[3];
    mixing_cols12(pick(row, c.idx_sm[0], c.sm_fixed[0]), pick(row, c.idx_sm[1], c.sm_fixed[1]),
                  pick(row, c.idx_sm[2], c.sm_fixed[2]), pick(row, c.idx_sm[3], c.sm_fixed[3]), c1r, c1i, c2r, c2i);
    const Herm3 S = rank2(pick(row, c.idx_mass[0], c.mass_fixed[0]), c1r, c1i,
                          pick(row, c.idx_mass[1], c.mass_fixed[1]), c2r, c2i);
    // NP part, per walker: sc1 T1 + sc2 T2, sc2 = 10^logLam, sc1 = sc2/100   (fr.py:380-393)
    const double sc2 = pow10_cold(pick(row, c.idx_scale, c.scale_fixed));
    const double sc1 = sc2 / 100.0;
    Herm3 N;
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
    const double src[3] = {c.src_fixed[0], c.src_fixed[1], c.src_fixed[2]};
    // source_flux[k] = source_ratio * E_k^gamma (fr.py:416-419) enters u_to_fr only through
    // src / sum(src) (fr.py:535): the E^gamma factor cancels, so the spectral index has no effect.
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const int nb = tb->nbins;
    for (int k = 0; k < nb; ++k) {
        const double u = tb->inv2e[k], v = tb->epow[k], w = tb->weight[k];
        Herm3 H;
        H.d0 = fma(u, S.d0, v * N.d0); H.d1 = fma(u, S.d1, v * N.d1); H.d2 = fma(u, S.d2, v * N.d2);
        H.r01 = fma(u, S.r01, v * N.r01); H.i01 = fma(u, S.i01, v * N.i01);
        H.r02 = fma(u, S.r02, v * N.r02); H.i02 = fma(u, S.i02, v * N.i02);
        H.r12 = fma(u, S.r12, v * N.r12); H.i12 = fma(u, S.i12, v * N.i12);
        double p[3][3], f[3];
        bin_moduli<CHECK_UNI>(H, p, residual);
        propagate(p, src, c.src_fixed_sum, f);                      // fr.py:451
        a0 = fma(f[0], w, a0); a1 = fma(f[1], w, a1); a2 = fma(f[2], w, a2);   // fr.py:454
    }
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
