"""How the device's unitarity verdict compares with the reference's STORED verdicts inside the band where nothing is promised
(half a decade around 1e-7; two decades on rows whose stored 10**logLam is not libm's).  G17 (the transition sweep) and the BSM
goldens G8-G14.  GPU.  usage: python tools/verdict_band_census.py"""
import math
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from common import BIN_EDGES, TEX_BY_VALUE, stored_verdict_zone
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
from oracle import oracle

z = {}
for f in ("golden.npz", "golden_r2.npz", "golden_r3.npz"):
    with np.load(os.path.join("tests", "golden", f), allow_pickle=False) as d:
        z.update({k: d[k] for k in d.files})
rows, ref_st = z["g17_rows"], z["g17_status"]
tot = dict(rows=0, must=0, must_agree=0, band=0, band_agree=0, other_pow=0)
for key in np.unique(rows[:, :2], axis=0):
    sel = np.all(rows[:, :2] == key, axis=1)
    dim, tex = int(key[0]), TEX_BY_VALUE[int(key[1])]
    ps = Cf.texture_paramset(dim)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=z["g17_source"], bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    th = np.ascontiguousarray(rows[sel][:, 2:])
    om = oracle.make_model(ps, "BSM_GAUSS", texture=tex.name, **kw)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        st = m.propagate(th)[1]
    must, res, same = stored_verdict_zone(oracle, om, th, z["g17_sc2"][sel])
    agree = (st == _lib.GF_ST_NON_UNITARY) == (ref_st[sel] == 2)
    tot["rows"] += len(th); tot["must"] += int(must.sum()); tot["must_agree"] += int(agree[must].sum())
    tot["band"] += int((~must).sum()); tot["band_agree"] += int(agree[~must].sum()); tot["other_pow"] += int((~same).sum())
print("G17 (792 walkers swept through the transition, 18 (dimension, texture) pairs):")
print("  rows %(rows)d; outside the band %(must)d, device = stored verdict on %(must_agree)d; inside the band %(band)d, device = stored verdict "
      "on %(band_agree)d; rows whose stored 10**logLam is not libm's: %(other_pow)d" % tot)
