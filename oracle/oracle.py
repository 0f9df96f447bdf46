"""ctypes front-end of the CPU oracle (oracle/golem_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (golemflavor_amd/) never does.  The oracle is pinned against the golden
vectors in tests/golden/ (tests/test_oracle_golden.py).

A model is built from a duck-typed paramset (anything iterable whose items have .name,
.ranges, .prior.name, .nominal_value, .std, .tag.name), so both golemflavor_amd.param.ParamSet
and the reference's own ParamSet (in the fixture generator) can be handed in.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

MAX_DIM = 16
MAX_BINS = 64

OK, OUT_OF_PRIOR, NON_UNITARY, NAN = 0, 1, 2, 3
MODE_PRIOR_ONLY, MODE_SM_GAUSS, MODE_BSM_GAUSS = 0, 1, 2
TEXTURES = {"OEU": 1, "OET": 2, "OUT": 3, "NONE": 4}
KINDS = {"UNIFORM": 0, "GAUSSIAN": 1, "LIMITEDGAUSS": 2}

# golemflavor/fr.py:42 and :313
MASS_EIGENVALUES = (7.40e-23, 2.515e-21)
NUFIT_ANGLES = (0.307, (1 - 0.02195) ** 2, 0.565, 3.97935)


class OrcModel(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32), ("mode", C.c_int32), ("texture", C.c_int32),
        ("dimension", C.c_int32), ("nbins", C.c_int32),
        ("idx_sm", C.c_int32 * 4), ("idx_mass", C.c_int32 * 2), ("idx_src", C.c_int32 * 2),
        ("idx_scale", C.c_int32), ("idx_mm", C.c_int32 * 4), ("idx_gamma", C.c_int32), ("idx_src_x", C.c_int32),
        ("kind", C.c_int32 * MAX_DIM),
        ("lo", C.c_double * MAX_DIM), ("hi", C.c_double * MAX_DIM),
        ("loc", C.c_double * MAX_DIM), ("sigma", C.c_double * MAX_DIM),
        ("log_mass", C.c_double * MAX_DIM),
        ("sm_fixed", C.c_double * 4), ("mass_fixed", C.c_double * 2),
        ("source_ratio", C.c_double * 3), ("scale_fixed", C.c_double),
        ("mm_fixed", C.c_double * 4), ("gamma_fixed", C.c_double),
        ("bestfit_fr", C.c_double * 3), ("smearing", C.c_double), ("offset", C.c_double),
        ("flat_llh", C.c_double), ("bin_edges", C.c_double * (MAX_BINS + 1)),
    ]


_lib = None


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "golem_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "liboracle.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        mp = C.POINTER(OrcModel)
        L.orc_model_size.restype = C.c_int
        assert L.orc_model_size() == C.sizeof(OrcModel), "oracle struct layout mismatch"
        L.orc_angles_to_fr.argtypes = [dp, dp]
        L.orc_angles_to_u.argtypes = [dp, dp, dp]
        L.orc_u_to_fr.argtypes = [dp, dp, dp, dp]
        L.orc_fr_to_angles.argtypes = [dp, dp]
        L.orc_cardano.argtypes = [dp, dp, dp, dp]
        L.orc_params_to_bsmu.argtypes = [dp, C.c_double, C.c_int, C.c_double, dp, dp, C.c_double, dp, dp]
        L.orc_params_to_bsmu.restype = C.c_int
        L.orc_multi_gaussian.argtypes = [dp, dp, C.c_double, C.c_double]
        L.orc_multi_gaussian.restype = C.c_double
        L.orc_lnprior.argtypes = [mp, dp]
        L.orc_lnprior.restype = C.c_double
        L.orc_lnprob_batch.argtypes = [mp, dp, C.c_int64, dp, dp, C.POINTER(C.c_int32)]
        L.orc_lnprob_batch_mt.argtypes = [mp, dp, C.c_int64, dp, C.c_int]
        L.orc_propagate_batch.argtypes = [mp, dp, C.c_int64, dp, C.POINTER(C.c_int32)]
        L.orc_flux_averaged.argtypes = [mp, dp, dp]
        L.orc_flux_averaged.restype = C.c_int
        L.orc_log_gauss_mass.argtypes = [C.c_double, C.c_double]
        L.orc_log_gauss_mass.restype = C.c_double
        L.orc_unitarity_residual_batch.argtypes = [mp, dp, C.c_int64, dp]
        L.orc_unitarity_residual_batch_sc2.argtypes = [mp, dp, C.c_int64, dp, dp]
        up = C.POINTER(C.c_uint32)
        L.orc_philox_raw.argtypes = [up, up, up]
        L.orc_haar_draw.argtypes = [dp, C.c_uint64, C.c_int64, C.c_int64, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(x, n):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    assert a.size == n, (a.size, n)
    return a


# ---- scalar physics (fp64 views of the long-double computation) ---------------------
def angles_to_fr(src_angles):
    out = np.empty(3)
    lib().orc_angles_to_fr(_dp(_vec(src_angles, 2)), _dp(out))
    return out


def angles_to_u(angles):
    re, im = np.empty(9), np.empty(9)
    lib().orc_angles_to_u(_dp(_vec(angles, 4)), _dp(re), _dp(im))
    return (re + 1j * im).reshape(3, 3)


def u_to_fr(source_fr, u):
    u = np.asarray(u, dtype=np.complex128).reshape(9)
    re, im = np.ascontiguousarray(u.real), np.ascontiguousarray(u.imag)
    out = np.empty(3)
    lib().orc_u_to_fr(_dp(_vec(source_fr, 3)), _dp(re), _dp(im), _dp(out))
    return out


def fr_to_angles(fr):
    out = np.empty(2)
    lib().orc_fr_to_angles(_dp(_vec(fr, 3)), _dp(out))
    return out


def cardano_eqn(ham):
    h = np.asarray(ham, dtype=np.complex128).reshape(9)
    re, im = np.ascontiguousarray(h.real), np.ascontiguousarray(h.imag)
    ore, oim = np.empty(9), np.empty(9)
    lib().orc_cardano(_dp(re), _dp(im), _dp(ore), _dp(oim))
    return (ore + 1j * oim).reshape(3, 3)


def params_to_BSMu(mm_angles, log_scale, dim, energy, mass_eigenvalues=MASS_EIGENVALUES,
                   sm_angles=NUFIT_ANGLES, epsilon=1e-7):
    """Returns (u, unitary_ok)."""
    ore, oim = np.empty(9), np.empty(9)
    st = lib().orc_params_to_bsmu(_dp(_vec(mm_angles, 4)), float(log_scale), int(dim), float(energy),
                                  _dp(_vec(mass_eigenvalues, 2)), _dp(_vec(sm_angles, 4)),
                                  float(epsilon), _dp(ore), _dp(oim))
    return (ore + 1j * oim).reshape(3, 3), st == 0


def multi_gaussian(fr, fr_bf, smearing, offset=-320):
    return lib().orc_multi_gaussian(_dp(_vec(fr, 3)), _dp(_vec(fr_bf, 3)), float(smearing), float(offset))


def log_gauss_mass(a, b):
    return lib().orc_log_gauss_mass(float(a), float(b))


# ---- model -------------------------------------------------------------------------
def _name(e):
    return getattr(e, "name", str(e)).rsplit(".", 1)[-1]


def make_model(paramset, mode, *, bestfit_fr=(1 / 3, 1 / 3, 1 / 3), smearing=0.02, offset=-320.0,
               source_ratio=(1.0, 2.0, 0.0), texture="NONE", dimension=3, binning=None,
               spectral_index=-2.0, flat_llh=1.0, scale_fixed=0.0, sm_fixed=None, src_columns=None, mm_fixed=None):
    """Flatten (paramset, args) into the oracle's POD model.

    `source_ratio` is used as given (the scripts normalise it first, scripts/fr.py:118).
    `binning` = array of bin edges (scripts/fr.py:122-124).
    """
    params = list(paramset)
    if len(params) > MAX_DIM:
        raise ValueError("ndim > %d" % MAX_DIM)
    m = OrcModel()
    m.ndim = len(params)
    m.mode = {"PRIOR_ONLY": 0, "SM_GAUSS": 1, "BSM_GAUSS": 2}.get(mode, mode)
    m.texture = TEXTURES[_name(texture)]
    m.dimension = int(dimension)
    names = [p.name for p in params]
    tags = [_name(p.tag) for p in params]
    for i, p in enumerate(params):
        kind = KINDS[_name(p.prior)]
        m.kind[i] = kind
        m.lo[i], m.hi[i] = float(p.ranges[0]), float(p.ranges[1])
        m.loc[i] = float(p.nominal_value) if kind else 0.0
        m.sigma[i] = float(p.std) if kind else 1.0
        if kind == 2:      # llh.py:25-29 GaussianBoundedRV with lower/upper = ranges
            a = (m.lo[i] - m.loc[i]) / m.sigma[i]
            b = (m.hi[i] - m.loc[i]) / m.sigma[i]
            m.log_mass[i] = log_gauss_mass(a, b)
        else:              # a=-inf, b=+inf -> log1p(0) = 0
            m.log_mass[i] = 0.0

    def idx(name):
        return names.index(name) if name in names else -1

    sm_names = ["s_12_2", "c_13_4", "s_23_2", "dcp"]
    mass_names = ["m21_2", "m3x_2"]
    if m.mode == MODE_BSM_GAUSS:
        # fr.py:425-435: all six from theta, or none (defaults MASS_EIGENVALUES / NUFIT_U)
        allp = set(sm_names + mass_names).issubset(names)
        for k in range(4):
            m.idx_sm[k] = idx(sm_names[k]) if allp else -1
        for k in range(2):
            m.idx_mass[k] = idx(mass_names[k]) if allp else -1
    else:
        # notebook: from_tag(SM_ANGLES, values=True) in declaration order (ipynb:320)
        sm_idx = [i for i, t in enumerate(tags) if t == "SM_ANGLES"][:4]
        for k in range(4):
            m.idx_sm[k] = sm_idx[k] if len(sm_idx) == 4 else -1
        m.idx_mass[0] = m.idx_mass[1] = -1
    for k in range(4):
        m.sm_fixed[k] = float(sm_fixed[k]) if sm_fixed is not None else NUFIT_ANGLES[k]
    m.mass_fixed[0], m.mass_fixed[1] = MASS_EIGENVALUES

    src_idx = [i for i, t in enumerate(tags) if t == "SRCANGLES"]
    if src_columns is not None:
        src_idx = [int(x) for x in src_columns]
    m.idx_src_x = -1
    if len(src_idx) == 2:
        m.idx_src[0], m.idx_src[1] = src_idx
    else:
        m.idx_src[0] = m.idx_src[1] = -1
        if len(src_idx) == 1:          # scripts/mc_x.py:41-44 astroX
            m.idx_src_x = src_idx[0]
    for k in range(3):
        m.source_ratio[k] = float(source_ratio[k])

    sc_idx = [i for i, t in enumerate(tags) if t == "SCALE"]
    m.idx_scale = sc_idx[0] if sc_idx else -1
    m.scale_fixed = float(scale_fixed)
    mm_idx = [i for i, t in enumerate(tags) if t == "MMANGLES"]
    for k in range(4):
        m.idx_mm[k] = mm_idx[k] if len(mm_idx) == 4 else -1
        m.mm_fixed[k] = float(mm_fixed[k]) if mm_fixed is not None else 0.0
    m.idx_gamma = idx("astroDeltaGamma")
    m.gamma_fixed = float(spectral_index)

    for k in range(3):
        m.bestfit_fr[k] = float(bestfit_fr[k])
    m.smearing, m.offset, m.flat_llh = float(smearing), float(offset), float(flat_llh)
    if binning is not None:
        be = np.asarray(binning, dtype=np.float64)
        if be.size - 1 > MAX_BINS:
            raise ValueError("too many bins")
        m.nbins = be.size - 1
        for k in range(be.size):
            m.bin_edges[k] = be[k]
    else:
        m.nbins = 0
    return m


def lnprior(model, theta):
    return lib().orc_lnprior(C.byref(model), _dp(_vec(theta, model.ndim)))


def lnprob_batch(model, theta, want_fr=False, want_status=False, threads=1):
    th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).reshape(-1, model.ndim))
    n = th.shape[0]
    out = np.empty(n)
    if threads > 1 and not want_fr and not want_status:
        lib().orc_lnprob_batch_mt(C.byref(model), _dp(th), n, _dp(out), int(threads))
        return out
    fr = np.empty((n, 3)) if want_fr else None
    st = np.empty(n, dtype=np.int32) if want_status else None
    lib().orc_lnprob_batch(C.byref(model), _dp(th), n, _dp(out),
                           _dp(fr) if want_fr else None,
                           st.ctypes.data_as(C.POINTER(C.c_int32)) if want_status else None)
    res = (out,)
    if want_fr:
        res += (fr,)
    if want_status:
        res += (st,)
    return res if len(res) > 1 else out


def propagate_batch(model, theta):
    th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).reshape(-1, model.ndim))
    n = th.shape[0]
    fr = np.empty((n, 3))
    st = np.empty(n, dtype=np.int32)
    lib().orc_propagate_batch(C.byref(model), _dp(th), n, _dp(fr), st.ctypes.data_as(C.POINTER(C.c_int32)))
    return fr, st


def flux_averaged_BSMu(model, theta):
    out = np.empty(3)
    st = lib().orc_flux_averaged(C.byref(model), _dp(_vec(theta, model.ndim)), _dp(out))
    if st == NON_UNITARY:
        raise AssertionError("Matrix is not unitary!")
    return out


def philox4x32_10(counter, key):
    c = (C.c_uint32 * 4)(*counter)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    lib().orc_philox_raw(c, k, out)
    return tuple(out)


def haar_draw(source_ratio, seed, n, first=0):
    ang = np.empty((n, 4))
    fr = np.empty((n, 3))
    lib().orc_haar_draw(_dp(_vec(source_ratio, 3)), int(seed), int(first), int(n), _dp(ang), _dp(fr))
    return fr, ang


def unitarity_residual_batch(model, theta, sc2=None):
    """Worst max(|tr|XX^+|-3|, |sum|XX^+|-3|) over the energy bins, per walker (fr.py:489-494).  `sc2`: per-walker
    values to use for 10**logLam instead of libm's pow (the ones the reference's numpy produced, golden G17)."""
    th = np.ascontiguousarray(np.asarray(theta, dtype=np.float64).reshape(-1, model.ndim))
    out = np.empty(th.shape[0])
    if sc2 is None:
        lib().orc_unitarity_residual_batch(C.byref(model), _dp(th), th.shape[0], _dp(out))
    else:
        s = np.ascontiguousarray(np.asarray(sc2, dtype=np.float64).reshape(-1))
        assert s.size == th.shape[0]
        lib().orc_unitarity_residual_batch_sc2(C.byref(model), _dp(th), th.shape[0], _dp(s), _dp(out))
    return out
