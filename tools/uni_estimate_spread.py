#!/usr/bin/env python3
"""Spread of the fp64 estimate of the reference's unitarity residual (what the evaluation kernel computes) against the
80-bit value (the oracle), over the transition zone of every (dimension, texture) pair: sizes the arbitration band of
gf_capi.hip.  GPU; GF_UNI_DUMP makes the kernel hand out its estimate in fr[0]."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from oracle import oracle
from common import BIN_EDGES, uniform_theta
os.environ["GF_UNI_DUMP"] = "1"
allr = []
for dim in (3, 4, 5, 6, 7, 8):
    for tex in (Texture.OEU, Texture.OET, Texture.OUT):
        ps = Cf.texture_paramset(dim)
        lo, hi = Cf.SCALE_BOUNDARIES[dim]
        rng = np.random.default_rng(dim * 10 + tex.value)
        n = 4000
        th = uniform_theta(ps, n, rng, seeds=True)
        th[:, 6] = rng.uniform(lo, hi, n)
        kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(1/3, 2/3, 0.), bestfit_fr=(1/3,)*3, smearing=0.02)
        om = oracle.make_model(ps, "BSM_GAUSS", texture=tex.name, **kw)
        r80 = oracle.unitarity_residual_batch(om, th)
        with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
            fr, st = m.propagate(th)
        est = fr[:, 0]
        sel = (r80 > 1e-10) & (r80 < 1e-4) & np.isfinite(est) & (est > 0)
        if sel.sum() == 0:
            continue
        lr = np.log10(est[sel] / r80[sel])
        allr.append(lr)
        print(dim, tex.name, 'n', sel.sum(), 'log10(est/r80): min %.2f  1%% %.2f  median %.2f  99%% %.2f  max %.2f' % (lr.min(), np.quantile(lr, 0.01), np.median(lr), np.quantile(lr, 0.99), lr.max()), flush=True)
lr = np.concatenate(allr)
print('ALL n', len(lr), 'quantiles 1e-4 %.2f 1e-3 %.2f 1e-2 %.2f | 0.99 %.2f 0.999 %.2f 0.9999 %.2f  min %.2f max %.2f' % tuple(list(np.quantile(lr, [1e-4, 1e-3, 1e-2, 0.99, 0.999, 0.9999])) + [lr.min(), lr.max()]))
