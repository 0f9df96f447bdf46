"""The log-posterior callback, GPU-evaluated: drop-in for the callables the reference hands to
emcee (`partial(ln_prob, ...)`: examples/inference.ipynb:356-371, golemflavor/llh.py:121-130,
scripts/mc_unitary.py:134-143, scripts/mc_texture.py:161-170).

`LnProb(theta)` keeps the reference convention for a single walker -- `theta` of length ndim ->
Python float, `-inf` outside the box prior, AssertionError on a length mismatch or on a
non-unitary BSM mixing matrix -- and additionally accepts a whole ensemble `(n, ndim)` ->
`(n,)` array in ONE kernel launch (emcee-3's `vectorize=True` convention), which is what
`golemflavor_amd.mcmc.EnsembleSampler` uses.
"""
import numpy as np

from . import _lib
from . import fr as fr_utils
from .descriptor import compile_model
from .enums import ParamTag, Texture
from .model import Model

__all__ = ["LnProb", "CubeLnProb", "notebook_ln_prob", "tutorial_ln_prob", "bsm_ln_prob", "prior_ln_prob", "lnprior",
           "ln_prob", "triangle_llh", "multi_gaussian"]


class LnProb:
    vectorized = True          # tells golemflavor_amd.mcmc.EnsembleSampler to batch the ensemble

    def __init__(self, desc, device=0, on_nonunitary="raise", check_unitarity=True):
        if on_nonunitary not in ("raise", "-inf"):
            raise ValueError("on_nonunitary must be 'raise' or '-inf'")
        self.model = Model(desc, device=device)
        self.ndim = self.model.ndim
        self.on_nonunitary = on_nonunitary
        self.check_unitarity = bool(check_unitarity) and self.model.mode == _lib.GF_MODE_BSM_GAUSS
        self.ncalls = 0
        self.nevals = 0

    def __call__(self, theta):
        single = np.ndim(theta) == 1
        if self.check_unitarity:
            lp, st = self.model.lnprob(theta, want_status=True)
            bad = st == _lib.GF_ST_NON_UNITARY
            if bad.any():
                if self.on_nonunitary == "raise":
                    # reference: AssertionError out of test_unitarity (fr.py:493-498)
                    th = np.atleast_2d(np.asarray(theta, dtype=float))[np.argmax(bad)]
                    raise AssertionError("Matrix is not unitary!\ntheta\n{0}".format(th))
                lp = np.where(bad, -np.inf, lp)
        else:
            lp = self.model.lnprob(theta, want_status=False)
        self.ncalls += 1
        self.nevals += lp.shape[0]
        return float(lp[0]) if single else lp

    def close(self):
        self.model.close()


def notebook_ln_prob(asimov_paramset, llh_paramset, device=0):
    """The posterior of examples/inference.ipynb (cells 21 and 23): lnprior + Gaussian flavor
    likelihood of u_to_fr(angles_to_fr(source angles), angles_to_u(mixing angles)) around the
    Asimov composition.  Smearing = asimov_paramset[first BESTFIT].std (ipynb:333)."""
    bestfit = asimov_paramset.from_tag(ParamTag.BESTFIT)
    bf = fr_utils.angles_to_fr(bestfit.values)                 # ipynb:330
    desc = compile_model(llh_paramset, "SM_GAUSS", bestfit_fr=bf, smearing=bestfit[0].std)
    return LnProb(desc, device=device)


def tutorial_ln_prob(asimov_paramset, llh_paramset, device=0):
    """The posterior of examples/tutorial.ipynb (cells 33 and 35): flat box priors + multi_gaussian(
    angles_to_fr(theta), angles_to_fr(asimov angles), smearing) -- the two flavor angles ARE theta and nothing
    oscillates, which is the Gaussian-likelihood kernel with the identity for a mixing matrix."""
    bestfit = asimov_paramset.from_tag(ParamTag.BESTFIT)
    bf = fr_utils.angles_to_fr(bestfit.values)
    desc = compile_model(llh_paramset, "SM_GAUSS", bestfit_fr=bf, smearing=bestfit[0].std,
                         sm_fixed=(0.0, 1.0, 0.0, 0.0), src_columns=(0, 1))
    return LnProb(desc, device=device)


def bsm_ln_prob(args, asimov_paramset, llh_paramset, smearing=0.02, device=0, **kw):
    """golemflavor/llh.py:121-130 `ln_prob(theta, args, asimov_paramset, llh_paramset)` with the
    GolemFit likelihood replaced by the Gaussian substitute the README sanctions
    (README.md:70-74): multi_gaussian(measured, angles_to_fr(asimov BESTFIT angles), smearing).
    `args` needs: source_ratio, dimension, texture, binning (edges); `args.no_bsm` (fr.py:437-438) is honoured:
    the posterior is then lnprior + the Gaussian around u_to_fr(source_ratio, sm_u) (descriptor.compile_model)."""
    bf = fr_utils.angles_to_fr(asimov_paramset.from_tag(ParamTag.BESTFIT, values=True))
    desc = compile_model(llh_paramset, "BSM_GAUSS", bestfit_fr=bf, smearing=smearing,
                         source_ratio=args.source_ratio, texture=args.texture,
                         dimension=args.dimension, binning=args.binning, no_bsm=bool(getattr(args, "no_bsm", False)))
    return LnProb(desc, device=device, **kw)


def prior_ln_prob(llh_paramset, device=0, flat_llh=1.0):
    """scripts/mc_unitary.py:134-143 / mc_texture.py:161-170: lnprior + flat likelihood (1.0)."""
    return LnProb(compile_model(llh_paramset, "PRIOR_ONLY", flat_llh=flat_llh), device=device)


def lnprior(theta, paramset, device=0):
    """golemflavor/llh.py:65-91 for one theta or a batch, GPU-evaluated."""
    f = prior_ln_prob(paramset, device=device, flat_llh=0.0)
    try:
        return f(theta)
    finally:
        f.close()


# ---- the reference's own function names and signatures -------------------------------------------------
# `partial(ln_prob, args=args, asimov_paramset=..., llh_paramset=...)` (scripts/fr.py:182-187) keeps working:
# the bound state is compiled into a device model the first time it is seen and looked up afterwards.
_BOUND = {}
_BOUND_MAX = 64


def _fingerprint(args, asimov_paramset, llh_paramset, smearing):
    """Everything of the bound state the posterior depends on (param *values* are overwritten by theta in the
    reference, llh.py:72-73, so they are not part of it -- except for columns the model keeps fixed)."""
    ps = tuple((p.name, float(p.ranges[0]), float(p.ranges[1]), p.prior.value, float(p.nominal_value),
                None if p.std is None else float(p.std), p.tag.value if hasattr(p.tag, "value") else str(p.tag))
               for p in llh_paramset)
    bf = tuple(float(x) for x in asimov_paramset.from_tag(ParamTag.BESTFIT, values=True))
    tex = args.texture.value if hasattr(args.texture, "value") else int(args.texture)
    return (ps, bf, tuple(float(x) for x in np.ravel(args.source_ratio)), int(args.dimension), tex,
            tuple(float(x) for x in np.ravel(args.binning)), float(smearing), bool(getattr(args, "no_bsm", False)))


def _bound(args, asimov_paramset, llh_paramset):
    smearing = float(getattr(args, "smearing", 0.02))
    key = _fingerprint(args, asimov_paramset, llh_paramset, smearing)
    f = _BOUND.get(key)
    if f is None:
        if len(_BOUND) >= _BOUND_MAX:                      # oldest first
            _BOUND.pop(next(iter(_BOUND))).close()
        f = _BOUND[key] = bsm_ln_prob(args, asimov_paramset, llh_paramset, smearing=smearing,
                                      device=int(getattr(args, "device", 0)))
    return f


def multi_gaussian(fr, fr_bf, smearing, offset=-320):
    """golemflavor/llh.py:32-54: log(multivariate_normal.pdf(fr, mean=fr_bf, cov=smearing^2 I)) + offset, with
    the reference's underflow to -inf (the pdf is formed in fp64 before the log).  Host closed form for single
    compositions; ensembles go through the model's kernel."""
    d = (np.asarray(fr, dtype=np.float64) - np.asarray(fr_bf, dtype=np.float64)) * np.sqrt(1.0 / (smearing * smearing))
    logpdf = -0.5 * (3 * np.log(2 * np.pi) + 3 * np.log(smearing * smearing) + np.sum(d * d, axis=-1))
    with np.errstate(divide="ignore", under="ignore"):
        return np.log(np.exp(logpdf)) + offset


def triangle_llh(theta, args, asimov_paramset, llh_paramset):
    """golemflavor/llh.py:94-119 with the Gaussian substitute for gf.get_llh (README.md:70-74): the composition
    comes from the device (flux_averaged_BSMu of theta), the Gaussian is applied to it."""
    if np.shape(theta)[-1] != len(llh_paramset):
        raise AssertionError('Length of MCMC scan is not the same as the input '
                             'params\ntheta={0}\nparamset]{1}'.format(theta, llh_paramset))
    f = _bound(args, asimov_paramset, llh_paramset)
    frs, st = f.model.propagate(theta)
    if np.any(st == _lib.GF_ST_NON_UNITARY):
        raise AssertionError("Matrix is not unitary!")
    bf = fr_utils.angles_to_fr(asimov_paramset.from_tag(ParamTag.BESTFIT, values=True))
    out = multi_gaussian(frs, bf, float(getattr(args, "smearing", 0.02)))
    return float(out[0]) if np.ndim(theta) == 1 else out


def ln_prob(theta, args, asimov_paramset, llh_paramset):
    """golemflavor/llh.py:121-130, same name and signature; theta (ndim,) -> float or (n, ndim) -> (n,)."""
    return _bound(args, asimov_paramset, llh_paramset)(theta)


class CubeLnProb:
    """MultiNest-style callback adapter (golemflavor/mn.py:26-45 `lnProb(cube, ndim, n_params, ...)`).

    `mn_paramset` is the scanned subset of `llh_paramset`; the unit cube is mapped onto its ranges,
    theta_i = (hi_i - lo_i) * cube_i + lo_i (mn.py:35-36), the other columns keep their current
    `.value`, and the batch goes to the GPU in one call: a nested sampler can hand over all its
    live-point proposals at once (`cube` of shape (n, ndim)) or one point (shape (ndim,)), as
    MultiNest does.  When `ln_prob` is a device posterior (`LnProb`) whose box for the scanned columns is the
    scanned ranges -- the reference's case: mn_paramset holds the same Param objects -- the map itself runs on
    the device (`gf_lnprob_cube_batch`) and only the cube crosses PCIe; otherwise it is one numpy expression.
    """

    def __init__(self, ln_prob, mn_paramset, llh_paramset):
        self.ln_prob = ln_prob
        names = list(llh_paramset.names)
        self.cols = np.array([names.index(n) for n in mn_paramset.names], dtype=np.intp)
        rng = np.array(mn_paramset.ranges, dtype=np.float64)
        self.lo, self.span = rng[:, 0], rng[:, 1] - rng[:, 0]
        self.base = np.array(llh_paramset.values, dtype=np.float64)
        self.ndim = len(self.cols)
        model = getattr(ln_prob, "model", None)
        box = np.array(llh_paramset.ranges, dtype=np.float64)[self.cols]
        self.on_device = (isinstance(ln_prob, LnProb) and model is not None and model.ndim == len(names)
                          and np.array_equal(box, rng))

    def __call__(self, cube, ndim=None, n_params=None):
        if ndim is not None and ndim != self.ndim:
            raise AssertionError("Length of MultiNest scan paramset is not the same as the input params")
        u = np.asarray([cube[i] for i in range(self.ndim)], dtype=np.float64) if np.ndim(cube) <= 1 and not \
            isinstance(cube, np.ndarray) else np.asarray(cube, dtype=np.float64)
        single = u.ndim == 1
        u = np.atleast_2d(u)[:, :self.ndim]
        if self.on_device:
            f = self.ln_prob
            lp, st = f.model.lnprob_cube(u, self.cols, self.base)
            bad = st == _lib.GF_ST_NON_UNITARY
            if f.check_unitarity and bad.any():
                if f.on_nonunitary == "raise":
                    raise AssertionError("Matrix is not unitary!")
                lp = np.where(bad, -np.inf, lp)
            f.ncalls += 1
            f.nevals += lp.shape[0]
            return float(lp[0]) if single else lp
        theta = np.tile(self.base, (u.shape[0], 1))
        theta[:, self.cols] = self.span * u + self.lo
        out = self.ln_prob(theta)
        return float(out[0]) if single else out
