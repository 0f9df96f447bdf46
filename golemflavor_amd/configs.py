"""The paramsets of the reference's example posteriors, built with this package's data model.

Each builder cites the reference code whose ParamSet it reproduces (same names, values,
ranges, seeds, stds, priors, tags and declaration order -- the order is the column order of
`theta`).  bench.py and the parity tests use these as the BASELINE.json configurations:

  C1/C2  notebook_paramsets()   examples/inference.ipynb cells 9 and 17 (6-dim, Gaussian llh)
  C3     unitary_paramset()     scripts/mc_unitary.py:28-41             (4-dim, flat priors)
  C4     texture_paramset(d)    scripts/mc_texture.py:28-76             (7-dim)
  --     mcx_paramset()         scripts/mc_x.py:28-46                   (5-dim: 4 mixing + astroX)
  C5     fr_paramsets(d, inj)   scripts/fr.py:30-104                    (12-dim)
"""
import numpy as np

from .enums import ParamTag, PriorsCateg
from .param import Param, ParamSet

# golemflavor/fr.py:42, :45-52, :313
MASS_EIGENVALUES = (7.40e-23, 2.515e-21)
SCALE_BOUNDARIES = {3: (-32, -20), 4: (-40, -24), 5: (-48, -27), 6: (-56, -30), 7: (-64, -33), 8: (-72, -36)}
NUFIT_ANGLES = (0.307, (1 - 0.02195) ** 2, 0.565, 3.97935)
DEFAULT_BINNING = (6e4, 1e7, 20)       # fr.py:283-286


def default_bin_edges(binning=DEFAULT_BINNING):
    """scripts/fr.py:122-124: logspace(log10 lo, log10 hi, nbins + 1)."""
    lo, hi, nb = binning
    return np.logspace(np.log10(lo), np.log10(hi), int(nb) + 1)


def _mixing_params(priors, eps_dcp):
    tag = ParamTag.SM_ANGLES
    lg = PriorsCateg.LIMITEDGAUSS if priors else None
    e = 1e-9 if eps_dcp else 0.0
    return [
        Param(name='s_12_2', value=0.307, seed=[0.26, 0.35], ranges=[0., 1.], std=0.013,
              tex=r's_{12}^2', prior=lg, tag=tag),
        Param(name='c_13_4', value=(1 - (0.02206)) ** 2, seed=[0.950, 0.961], ranges=[0., 1.],
              std=0.00147, tex=r'c_{13}^4', prior=lg, tag=tag),
        Param(name='s_23_2', value=0.538, seed=[0.31, 0.75], ranges=[0., 1.], std=0.069,
              tex=r's_{23}^2', prior=lg, tag=tag),
        Param(name='dcp', value=4.08404, seed=[0 + e, 2 * np.pi - e], ranges=[0., 2 * np.pi],
              std=2.0, tex=r'\delta_{CP}', tag=tag),
    ]


def _mass_params():
    tag, g = ParamTag.SM_ANGLES, PriorsCateg.GAUSSIAN
    return [
        Param(name='m21_2', value=7.40E-23, seed=[7.2E-23, 7.6E-23], ranges=[6.80E-23, 8.02E-23],
              std=2.1E-24, tex=r'\Delta m_{21}^2{\rm GeV}^{-2}', prior=g, tag=tag),
        Param(name='m3x_2', value=2.494E-21, seed=[2.46E-21, 2.53E-21], ranges=[2.399E-21, 2.593E-21],
              std=3.3E-23, tex=r'\Delta m_{3x}^2{\rm GeV}^{-2}', prior=g, tag=tag),
    ]


def _gf_nuisance():
    tag, lg = ParamTag.NUISANCE, PriorsCateg.LIMITEDGAUSS
    return [
        Param(name='convNorm', value=1., seed=[0.5, 2.], ranges=[0.1, 10.], std=0.4, prior=lg, tag=tag),
        Param(name='promptNorm', value=0., seed=[0., 6.], ranges=[0., 20.], std=2.4, prior=lg, tag=tag),
        Param(name='muonNorm', value=1., seed=[0.1, 2.], ranges=[0., 10.], std=0.1, tag=tag),
        Param(name='astroNorm', value=6.9, seed=[0., 5.], ranges=[0., 20.], std=1.5, tag=tag),
        Param(name='astroDeltaGamma', value=2.5, seed=[2.4, 3.], ranges=[-5., 5.], std=0.1, tag=tag),
    ]


def _scale_param(dimension):
    b = SCALE_BOUNDARIES[dimension]
    return Param(name='logLam', value=float(np.mean(b)), ranges=b, std=3, tag=ParamTag.SCALE)


def tutorial_paramsets(asimov_angles, smearing=0.02):
    """(asimov_paramset, llh_paramset) of examples/tutorial.ipynb cell 30: the two measured flavor angles are
    sampled directly, flat in their boxes."""
    tag = ParamTag.BESTFIT
    asimov = ParamSet([
        Param(name='measured_angle1', value=asimov_angles[0], ranges=[0., 1.], std=smearing, tag=tag,
              tex=r'\sin^4\phi_\oplus'),
        Param(name='measured_angle2', value=asimov_angles[1], ranges=[-1., 1.], std=smearing, tag=tag,
              tex=r'\cos(2\psi_\oplus)'),
    ])
    llh = ParamSet([
        Param(name='measured_angle1', value=0, ranges=[0., 1.], tag=tag, tex=r'\sin^4\phi_\oplus'),
        Param(name='measured_angle2', value=0, ranges=[-1., 1.], tag=tag, tex=r'\cos(2\psi_\oplus)'),
    ])
    return asimov, llh


def notebook_paramsets(asimov_angles, smearing=0.02):
    """(asimov_paramset, llh_paramset) of examples/inference.ipynb cells 9 and 17.

    `asimov_angles` = fr_to_angles(u_to_fr((1,0,0), NUFIT_U)) in the notebook.
    """
    tag = ParamTag.BESTFIT
    asimov = ParamSet([
        Param(name='measured_angle1', value=asimov_angles[0], ranges=[0., 1.], std=smearing, tag=tag,
              tex=r'\sin^4\phi_\oplus'),
        Param(name='measured_angle2', value=asimov_angles[1], ranges=[-1., 1.], std=smearing, tag=tag,
              tex=r'\cos(2\psi_\oplus)'),
    ])
    tag = ParamTag.SRCANGLES
    src = [
        Param(name='source_angle1', value=0, ranges=[0., 1.], tag=tag, tex=r'\sin^4\phi_S'),
        Param(name='source_angle2', value=0, ranges=[-1., 1.], tag=tag, tex=r'\cos(2\psi_S)'),
    ]
    return asimov, ParamSet(_mixing_params(priors=True, eps_dcp=False) + src)


def unitary_paramset():
    """scripts/mc_unitary.py:28-41: four mixing parameters, all UNIFORM."""
    return ParamSet(_mixing_params(priors=False, eps_dcp=True))


def mcx_paramset():
    """scripts/mc_x.py:28-46: four mixing parameters (priors as scripts/fr.py) + the source parameter astroX,
    source composition (x, 1 - x, 0)."""
    return ParamSet(_mixing_params(True, True) + [
        Param(name='astroX', value=0.5, seed=[0., 1.], ranges=[0., 1.], std=0.1, tex=r'x', tag=ParamTag.SRCANGLES)])


def texture_paramset(dimension):
    """scripts/mc_texture.py:28-76: 4 mixing + 2 mass splittings + logLam."""
    return ParamSet(_mixing_params(True, True) + _mass_params() + [_scale_param(dimension)])


def fr_paramsets(dimension, bestfit_angles):
    """(asimov_paramset, llh_paramset) of scripts/fr.py:62-104 (12-dim)."""
    nuis = _gf_nuisance()
    llh_ps = ParamSet(_mixing_params(True, True) + _mass_params() + nuis + [_scale_param(dimension)])
    tag = ParamTag.BESTFIT
    asimov = ParamSet(nuis + [
        Param(name='astroFlavorAngle1', value=bestfit_angles[0], ranges=[0., 1.], std=0.2, tag=tag),
        Param(name='astroFlavorAngle2', value=bestfit_angles[1], ranges=[-1., 1.], std=0.2, tag=tag),
    ])
    return asimov, llh_ps
