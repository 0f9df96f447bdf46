"""ParamSet + args -> flat `gf_model_desc` (the POD the C ABI takes).

This is the boundary the reference draws with `functools.partial(ln_prob, ...)`
(examples/inference.ipynb:366-371, scripts/fr.py:182-187): everything the callback needs that
is not `theta`.  Compiled once per run; immutable afterwards (the reference instead mutates the
bound ParamSets on every call, llh.py:72-73).
"""
import numpy as np
from scipy import special as sc

from . import _lib
from .configs import MASS_EIGENVALUES, NUFIT_ANGLES
from .enums import ParamTag, PriorsCateg, Texture

MODES = {"PRIOR_ONLY": _lib.GF_MODE_PRIOR_ONLY, "SM_GAUSS": _lib.GF_MODE_SM_GAUSS,
         "BSM_GAUSS": _lib.GF_MODE_BSM_GAUSS}

SM_NAMES = ("s_12_2", "c_13_4", "s_23_2", "dcp")
MASS_NAMES = ("m21_2", "m3x_2")


def log_gauss_mass(a, b):
    """log of the standard-normal probability mass of [a, b].

    This is the normalisation of scipy.stats.truncnorm that the reference's GaussianBoundedRV
    (llh.py:25-29) freezes; the case split (left tail / right tail folded onto the left /
    central via log1p) follows scipy's published implementation so the constant agrees to
    rounding.  A per-run constant: evaluated here once per Gaussian column.
    """
    a, b = float(a), float(b)
    if not a < b:
        raise ValueError("empty truncation interval [%r, %r]" % (a, b))

    def left(lo, hi):                  # hi <= 0: log(Phi(hi) - Phi(lo))
        lhi, llo = sc.log_ndtr(hi), sc.log_ndtr(lo)
        return float(lhi + np.log1p(-np.exp(llo - lhi)))

    if b <= 0:
        return left(a, b)
    if a > 0:
        return left(-b, -a)
    return float(np.log1p(-sc.ndtr(a) - sc.ndtr(-b)))


def compile_model(llh_paramset, mode, *, bestfit_fr=None, smearing=None, offset=-320.0,
                  source_ratio=(1.0, 2.0, 0.0), texture=Texture.NONE, dimension=3, binning=None,
                  spectral_index=-2.0, flat_llh=1.0, scale_fixed=None, mm_fixed=None,
                  sm_fixed=None, src_columns=None, no_bsm=False):
    """Flatten a posterior definition into a `GfModelDesc`.

    llh_paramset : ParamSet whose order is the column order of theta.
    mode         : "PRIOR_ONLY" | "SM_GAUSS" | "BSM_GAUSS".
    bestfit_fr   : injected / best-fit composition of the Gaussian likelihood (llh.py:32-54).
    source_ratio : args.source_ratio, used as given when the source is not sampled
                   (the scripts normalise it first, scripts/fr.py:118).
    binning      : energy bin *edges* (args.binning after process_args, scripts/fr.py:122-124).
    sm_fixed     : the four mixing parameters used when they are not sampled (default: NuFIT, fr.py:313);
                   (0, 1, 0, 0) is the identity matrix, i.e. no oscillation (examples/tutorial.ipynb).
    src_columns  : the two columns holding the source flavor angles when they are not tagged SRCANGLES.
    mm_fixed     : the four NP mixing parameters of Texture.NONE when they are not sampled (the tuple
                   params_to_BSMu takes directly, fr.py:354-358,378).

    no_bsm       : args.no_bsm of the flux-averaged posterior (fr.py:437-438): no new-physics term at all.  The
                   reference's branch cannot run (it hands u_to_fr a 2-D source flux, an einsum subscript mismatch:
                   SURVEY App. C-2); it is defined here as what it evidently means, u_to_fr(source_ratio, sm_u) -- the
                   standard propagation with the mixing angles taken from theta under the flux average's own rule
                   (all six oscillation parameters scanned, else NuFIT: fr.py:425-435) and the source fixed.  The
                   posterior then runs through the SM kernels (no energy bins, no unitarity test: nothing is
                   diagonalised).  Parity unpinned: there is no reference output to compare with.

    A single SRCANGLES-tagged column is scripts/mc_x.py's `astroX` (mc_x.py:41-44): the source composition is
    normalize_fr((x, 1 - x, 0)) of that column (mc_x.py:186-190).
    """
    params = list(llh_paramset)
    ndim = len(params)
    if not 1 <= ndim <= _lib.GF_MAX_DIM:
        raise ValueError("ndim must be in 1..%d, got %d" % (_lib.GF_MAX_DIM, ndim))
    d = _lib.GfModelDesc()
    d.abi_version = _lib.GF_ABI_VERSION
    d.ndim = ndim
    d.mode = MODES[mode] if isinstance(mode, str) else int(mode)
    bsm_rules = d.mode == _lib.GF_MODE_BSM_GAUSS               # column rules of flux_averaged_BSMu (fr.py:420-435)
    if no_bsm:
        if not bsm_rules:
            raise ValueError("no_bsm only applies to the flux-averaged (BSM_GAUSS) posterior")
        d.mode = _lib.GF_MODE_SM_GAUSS
    d.texture = Texture(texture).value if not isinstance(texture, Texture) else texture.value
    d.dimension = int(dimension)
    names = [p.name for p in params]

    for i, p in enumerate(params):
        d.prior_kind[i] = p.prior.value
        d.lo[i], d.hi[i] = float(p.ranges[0]), float(p.ranges[1])
        if p.prior is PriorsCateg.UNIFORM:
            d.loc[i], d.sigma[i], d.log_mass[i] = 0.0, 1.0, 0.0
            continue
        if p.std is None:
            raise ValueError("param %r has a Gaussian prior but no std" % p.name)
        loc, sig = float(p.nominal_value), float(p.std)
        d.loc[i], d.sigma[i] = loc, sig
        if p.prior is PriorsCateg.LIMITEDGAUSS:      # llh.py:86-90
            d.log_mass[i] = log_gauss_mass((d.lo[i] - loc) / sig, (d.hi[i] - loc) / sig)
        else:                                        # llh.py:82-85: unbounded
            d.log_mass[i] = 0.0

    def col(name):
        return names.index(name) if name in names else -1

    if bsm_rules:
        # fr.py:422-435: mixing angles and mass splittings come from theta only if all six are scanned
        scanned = set(SM_NAMES + MASS_NAMES).issubset(names)
        sm_idx = [col(n) if scanned else -1 for n in SM_NAMES]
        mass_idx = [col(n) if scanned else -1 for n in MASS_NAMES]
    else:
        # notebook: from_tag(SM_ANGLES, values=True), declaration order (ipynb:320)
        tagged = [i for i, p in enumerate(params) if p.tag is ParamTag.SM_ANGLES][:4]
        sm_idx = tagged if len(tagged) == 4 else [-1] * 4
        mass_idx = [-1, -1]
    for k in range(4):
        d.idx_sm[k] = sm_idx[k]
        d.sm_fixed[k] = float(sm_fixed[k]) if sm_fixed is not None else NUFIT_ANGLES[k]
        d.mm_fixed[k] = float(mm_fixed[k]) if mm_fixed is not None else 0.0
    for k in range(2):
        d.idx_mass[k] = mass_idx[k]
        d.mass_fixed[k] = MASS_EIGENVALUES[k]

    src_idx = [i for i, p in enumerate(params) if p.tag is ParamTag.SRCANGLES]
    if bsm_rules:
        src_idx = []                                           # the flux average propagates args.source_ratio (fr.py:416-419)
    if src_columns is not None:
        src_idx = [int(x) for x in src_columns]
    d.idx_src_x = -1
    if len(src_idx) == 2:
        d.idx_src[0], d.idx_src[1] = src_idx
    elif not src_idx:
        d.idx_src[0] = d.idx_src[1] = -1
    elif len(src_idx) == 1:
        d.idx_src[0] = d.idx_src[1] = -1
        d.idx_src_x = src_idx[0]
    else:
        raise ValueError("expected 0, 1 (astroX) or 2 SRCANGLES params, got %d" % len(src_idx))
    for k in range(3):
        d.source_ratio[k] = float(source_ratio[k])

    scale_idx = [i for i, p in enumerate(params) if p.tag is ParamTag.SCALE]
    d.idx_scale = scale_idx[0] if scale_idx else -1
    d.scale_fixed = float(scale_fixed) if scale_fixed is not None else 0.0
    mm_idx = [i for i, p in enumerate(params) if p.tag is ParamTag.MMANGLES]
    for k in range(4):
        d.idx_mm[k] = mm_idx[k] if len(mm_idx) == 4 else -1
    d.idx_gamma = col("astroDeltaGamma")
    d.gamma_fixed = float(spectral_index)

    if d.mode != _lib.GF_MODE_PRIOR_ONLY:
        if bestfit_fr is None or smearing is None:
            raise ValueError("Gaussian-likelihood modes need bestfit_fr and smearing")
        for k in range(3):
            d.bestfit_fr[k] = float(bestfit_fr[k])
        d.smearing = float(smearing)
    d.offset = float(offset)
    d.flat_llh = float(flat_llh)

    if d.mode == _lib.GF_MODE_BSM_GAUSS:
        if binning is None:
            raise ValueError("BSM mode needs the energy bin edges")
        if d.texture == Texture.NONE.value and len(mm_idx) != 4 and mm_fixed is None:
            raise ValueError("texture NONE needs four MMANGLES params (fr.py:378) or an explicit mm_fixed")
        if d.idx_scale < 0 and scale_fixed is None:
            raise ValueError("BSM mode needs a SCALE-tagged param (logLam) or an explicit scale_fixed")
        edges = np.asarray(binning, dtype=np.float64)
        if edges.ndim != 1 or not 2 <= edges.size <= _lib.GF_MAX_BINS + 1:
            raise ValueError("binning must hold 2..%d edges" % (_lib.GF_MAX_BINS + 1))
        d.nbins = edges.size - 1
        for k in range(edges.size):
            d.bin_edges[k] = edges[k]
    return d
