#!/usr/bin/env python3
"""Spread of the fp64 estimate of the reference's unitarity residual (what the evaluation kernel computes) against the
80-bit value (the oracle), per (dimension, texture) pair and binned by the SM weight `a` of the top energy bin: sizes the
bands of tier 2 (gf_capi.hip).  GPU; GF_UNI_DUMP makes the kernel hand out its estimate in fr[0], GF_UNI_NO_WEIGHT_GATE
switches tier 1 off so that every bin is estimated."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from oracle import oracle
from common import BIN_EDGES, uniform_theta
os.environ["GF_DIAGNOSTICS"] = "1"; os.environ["GF_UNI_DUMP"] = "1"; os.environ["GF_UNI_NO_WEIGHT_GATE"] = "1"
cent = np.sqrt(BIN_EDGES[:-1] * BIN_EDGES[1:])
rows = []
for dim in (4, 5, 6, 7, 8):
    for tex in (Texture.OEU, Texture.OET, Texture.OUT):
        ps = Cf.texture_paramset(dim)
        lo, hi = Cf.SCALE_BOUNDARIES[dim]
        rng = np.random.default_rng(dim * 10 + tex.value + 100)
        n = 12000
        th = uniform_theta(ps, n, rng, seeds=True)
        th[:, 6] = rng.uniform(lo, hi, n)
        kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(1/3, 2/3, 0.), bestfit_fr=(1/3,)*3, smearing=0.02)
        om = oracle.make_model(ps, "BSM_GAUSS", texture=tex.name, **kw)
        r80 = oracle.unitarity_residual_batch(om, th)
        with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
            fr, st = m.propagate(th)
        est = fr[:, 0]
        sm = (th[:, 4] + th[:, 5]) / (2 * cent[-1]); npt = 1.01 * 10.0 ** th[:, 6] * cent[-1] ** (dim - 3)
        amin = sm / (sm + npt)
        for reg, sel in (('a>=1e-10', amin >= 1e-10), ('1e-12..1e-10', (amin >= 1e-12) & (amin < 1e-10)), ('1e-14..1e-12', (amin >= 1e-14) & (amin < 1e-12)), ('1e-16..1e-14', (amin >= 1e-16) & (amin < 1e-14)), ('<1e-16', amin < 1e-16)):
            ok = sel & np.isfinite(est) & (est > 0) & (r80 > 1e-12)
            if ok.sum() == 0: continue
            lr = np.log10(est[ok] / r80[ok])
            # dangerous cases: est says fail (>=1e-7*10^2.5) but r80<1e-7 ; est says ok (<1e-7*10^-3.5) but r80>=1e-7
            d1 = ok & (est >= 1e-7 * 10**2.5) & (r80 < 1e-7); d2 = ok & (est < 1e-7 * 10**-3.5) & (r80 >= 1e-7)
            print(dim, tex.name, reg, 'n', ok.sum(), 'log ratio min %.2f max %.2f' % (lr.min(), lr.max()), 'wrong-fail', d1.sum(), 'wrong-ok', d2.sum(), 'frac r80>=1e-7: %.3f' % np.mean(r80[ok] >= 1e-7), flush=True)
