"""Exact (60-digit mpmath) evaluation of the reference's BSM formulas on fp64-rounded inputs -- test infrastructure.

Used by tests/golden/make_golden.py for the `*_exact` fixtures and by tests/test_gpu_fuzz.py to arbitrate between the
kernel and the long-double oracle where the reference's own 80-bit closed form is noisy.  Pure mpmath + numpy:
nothing of the reference is imported here."""
import numpy as np

# The BSM branch of the reference is ill-conditioned (fr.py:204-236 forms Q, R, acos(R/sqrt(Q^3)) with
# catastrophic cancellation for hierarchical spectra), so its own float128 output carries noise well
# above 1e-10 in part of the domain.  To tell "the reference's noise" from "our error" the fixtures
# also hold the *exact* value of the reference's formulas, evaluated here with 60-digit mpmath on the
# same fp64-rounded inputs (10**logLam, E**(d-3), bin centres and widths are fp64 in the reference).
import mpmath as mp

mp.mp.dps = 60


def mp_angles_to_u(ang):
    s12_2, c13_4, s23_2, dcp = [mp.mpf(float(a)) for a in ang]
    c13_2 = mp.sqrt(c13_4)
    s12, c12 = mp.sqrt(s12_2), mp.sqrt(1 - s12_2)
    c13, s13 = mp.sqrt(c13_2), mp.sqrt(1 - c13_2)
    s23, c23 = mp.sqrt(s23_2), mp.sqrt(1 - s23_2)
    em, ep = mp.expj(-dcp), mp.expj(dcp)
    p1 = mp.matrix([[1, 0, 0], [0, c23, s23], [0, -s23, c23]])
    p2 = mp.matrix([[c13, 0, s13 * em], [0, 1, 0], [-s13 * ep, 0, c13]])
    p3 = mp.matrix([[c12, s12, 0], [-s12, c12, 0], [0, 0, 1]])
    return p1 * p2 * p3


def mp_cardano(h):
    tr = h[0, 0] + h[1, 1] + h[2, 2]
    hh = h * h
    a = -tr
    b = (tr ** 2 - (hh[0, 0] + hh[1, 1] + hh[2, 2])) / 2
    c = -mp.det(h)
    Q = (a ** 2 - 3 * b) / 9
    R = (2 * a ** 3 - 9 * a * b + 27 * c) / 54
    theta = mp.acos(R / mp.sqrt(Q ** 3))
    E = [-2 * mp.sqrt(Q) * mp.cos((theta + s) / 3) - a / 3 for s in (0, -2 * mp.pi, 2 * mp.pi)]
    m = mp.matrix(3, 3)
    for k in range(3):
        A = h[1, 2] * (h[0, 0] - E[k]) - h[1, 0] * h[0, 2]
        B = h[2, 0] * (h[1, 1] - E[k]) - h[2, 1] * h[1, 0]
        C = h[1, 0] * (h[2, 2] - E[k]) - h[1, 2] * h[2, 0]
        N = mp.sqrt(abs(A * B) ** 2 + abs(A * C) ** 2 + abs(B * C) ** 2)
        m[0, k], m[1, k], m[2, k] = mp.conj(B) * C / N, A * C / N, A * B / N
    return m


def mp_bsmu(mm_angles, log_scale, dim, energy, mass, sm_u):
    sc2 = mp.mpf(float(np.power(10., log_scale)))
    sc1 = mp.mpf(float(np.power(10., log_scale) / 100.))
    mass_m = mp.diag([0, mp.mpf(float(mass[0])), mp.mpf(float(mass[1]))])
    sm_ham = mp.mpf(float(1. / (2 * energy))) * (sm_u * mass_m * sm_u.H)
    npu = mp_angles_to_u(mm_angles)
    bsm = mp.mpf(float(np.float64(energy) ** (dim - 3))) * (npu * mp.diag([0, sc1, sc2]) * npu.H)
    return mp_cardano(sm_ham + bsm)


def mp_abs2(u):
    return np.array([[float(abs(u[i, j]) ** 2) for j in range(3)] for i in range(3)])


def mp_u_to_fr(src, u):
    p = [[abs(u[a, i]) ** 2 for i in range(3)] for a in range(3)]
    tot = sum(mp.mpf(float(s)) for s in src)
    return [sum(p[a][i] * p[b][i] * mp.mpf(float(src[a])) for a in range(3) for i in range(3)) / tot
            for b in range(3)]


def mp_flux_avg(theta, tex_angles, dim, source_ratio, binning):
    sm_u = mp_angles_to_u(theta[:4])
    mass = theta[4:6]
    centres = np.sqrt(binning[:-1] * binning[1:])
    widths = np.abs(np.diff(binning))
    acc = [mp.mpf(0)] * 3
    for e, w in zip(centres, widths):
        u = mp_bsmu(tex_angles, theta[-1], dim, e, mass, sm_u)
        r = mp_u_to_fr(source_ratio, u)
        acc = [x + y * mp.mpf(float(w)) for x, y in zip(acc, r)]
    tot = sum(acc)
    return [x / tot for x in acc]


def mp_angles_roundtrip_fr(frx):
    """fr -> fr_to_angles -> angles_to_fr, exactly (llh.py:109-112 + the Gaussian substitute)."""
    tot = sum(frx)
    f = [x / tot for x in frx]
    s = 1 - f[2]
    if s == 0:
        return [mp.mpf(0), mp.mpf(0), mp.mpf(1)]
    return [abs(f[0]), abs(s - f[0]), abs(1 - s)]


# the fixed NP mixing angles behind the named textures (fr.py:370-376, z = 1e-9)
Z = 0. + 1e-9
TEXTURE_ANGLES = {"OEU": (0.5, 1.0, Z, Z), "OET": (Z, 0.25, Z, Z), "OUT": (Z, 1.0, 0.5, Z)}


def exact_flux_avg(theta_rows, texture_name, dim, source_ratio, binning):
    """fp64 array (n, 3) of the exact flux-averaged compositions for the rows of theta (columns: 4 mixing
    parameters, 2 mass splittings, ..., logLam last)."""
    ang = TEXTURE_ANGLES[texture_name]
    return np.array([[float(x) for x in mp_flux_avg(list(map(float, row)), ang, dim, source_ratio, np.asarray(binning, dtype=float))]
                     for row in theta_rows])
