"""Host-side scalar flavor utilities used for *per-run setup* (best-fit composition, Asimov
angles, known-answer checks).  They mirror the names and argument conventions of
golemflavor/fr.py but are not an evaluation path: every per-walker computation goes through
the HIP kernels (golemflavor_amd.model.Model).  Extended precision (numpy longdouble, as the
reference uses, fr.py:22-30) costs nothing here because each is called a handful of times.
"""
import numpy as np

from .configs import MASS_EIGENVALUES, NUFIT_ANGLES, SCALE_BOUNDARIES  # noqa: F401

_LD = np.longdouble
_CLD = np.clongdouble


def normalize_fr(fr):
    """Scale a flavor combination so that it sums to one (fr.py:240-259)."""
    fr = np.asarray(fr)
    return fr / float(np.sum(fr))


def angles_to_fr(src_angles):
    """(sin^4 phi, cos 2psi) -> (f_e, f_mu, f_tau) as Python floats (fr.py:82-113).

    Uses sin^2(acos(c)/2) = (1 - c)/2, which is what the reference's acos/sin chain evaluates.
    """
    sphi4, c2psi = (_LD(v) for v in src_angles)
    sphi2 = np.sqrt(sphi4)
    spsi2 = (1 - c2psi) / 2
    return (float(abs(sphi2 * (1 - spsi2))), float(abs(sphi2 * spsi2)), float(abs(1 - sphi2)))


def fr_to_angles(ratios):
    """(f_e, f_mu, f_tau) -> (sin^4 phi, cos 2psi) (fr.py:289-310); (0, 0) when f_tau = 1."""
    f = normalize_fr(np.asarray(ratios, dtype=_LD))
    sphi2 = 1 - f[2]
    if sphi2 == 0:
        return (0., 0.)
    return (sphi2 ** 2, 2 * (f[0] / sphi2) - 1)      # cos(2 acos sqrt(x)) = 2x - 1


def angles_to_u(bsm_angles):
    """(s12^2, c13^4, s23^2, delta) -> 3x3 complex mixing matrix, PDG convention (fr.py:116-162)."""
    s12_2, c13_4, s23_2, dcp = (_LD(v) for v in bsm_angles)
    c13_2 = np.sqrt(c13_4)
    s12, c12 = np.sqrt(s12_2), np.sqrt(1 - s12_2)
    c13, s13 = np.sqrt(c13_2), np.sqrt(1 - c13_2)
    s23, c23 = np.sqrt(s23_2), np.sqrt(1 - s23_2)
    ep = _CLD(np.cos(dcp) + 1j * np.sin(dcp))
    em = np.conj(ep)
    return np.array([
        [c12 * c13, s12 * c13, s13 * em],
        [-s12 * c23 - c12 * s23 * s13 * ep, c12 * c23 - s12 * s23 * s13 * ep, s23 * c13],
        [s12 * s23 - c12 * c23 * s13 * ep, -c12 * s23 - s12 * c23 * s13 * ep, c23 * c13],
    ], dtype=_CLD)


NUFIT_U = angles_to_u(NUFIT_ANGLES)
"""NuFIT mixing matrix (fr.py:313)."""


def u_to_fr(source_fr, matrix):
    """Decoherent propagation: out_b = sum_a sum_i |U_ai|^2 |U_bi|^2 src_a / sum(src) (fr.py:502-536)."""
    p = np.abs(np.asarray(matrix, dtype=_CLD)) ** 2
    src = np.asarray(source_fr, dtype=_LD)
    return (p @ p.T) @ src / np.sum(src)


def flux_averaged_BSMu(theta, args, spectral_index, llh_paramset):
    """golemflavor/fr.py:403-458, same name and signature, evaluated on the device: the flux-averaged measured
    composition for theta (ndim,) -> (3,) or (n, ndim) -> (n, 3).  `spectral_index` cancels in u_to_fr
    (fr.py:535 divides by the summed source flux) and is accepted for signature compatibility.
    Raises AssertionError('Matrix is not unitary!') where the reference's test_unitarity would (fr.py:493-498).
    `args.no_bsm` (fr.py:437-438): u_to_fr(source_ratio, sm_u), see descriptor.compile_model."""
    from . import _lib
    from .descriptor import compile_model
    from .model import Model
    desc = compile_model(llh_paramset, "BSM_GAUSS", bestfit_fr=(1 / 3,) * 3, smearing=0.02, source_ratio=args.source_ratio,
                         texture=args.texture, dimension=args.dimension, binning=args.binning,
                         no_bsm=bool(getattr(args, "no_bsm", False)))
    with Model(desc, device=int(getattr(args, "device", 0))) as m:
        frs, st = m.propagate(theta)
    if np.any(st == _lib.GF_ST_NON_UNITARY):
        raise AssertionError("Matrix is not unitary!")
    return frs[0] if np.ndim(theta) == 1 else frs
