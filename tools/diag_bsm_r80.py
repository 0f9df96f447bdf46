import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from common import BIN_EDGES, uniform_theta
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
for dim, tex in [(3, Texture.OET), (6, Texture.OUT), (6, Texture.OEU), (4, Texture.OET), (5, Texture.OEU), (7, Texture.OUT), (8, Texture.OET)]:
    ps = Cf.texture_paramset(dim); lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(1000 + dim + tex.value)
    th = uniform_theta(ps, 6000, rng, seeds=True); th[:, 6] = rng.uniform(lo, hi, len(th))
    src = np.array([1., 2., 0.]) / 3
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=src, bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    om = O.make_model(ps, "BSM_GAUSS", texture=tex.name, **kw)
    ref, ref_fr, ref_st = O.lnprob_batch(om, th, want_fr=True, want_status=True)
    r80 = O.unitarity_residual_batch(om, th)
    with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
        lp, fr, st = m.lnprob(th, want_fr=True)
    good = (ref_st == 0) & (st == 0)
    err = np.abs(fr - ref_fr).max(axis=1)
    big = good & (err > 1e-11)
    ratio = err[big] / np.maximum(r80[big], 1e-300)
    print(dim, tex.name, "good", good.sum(), "rows err>1e-11:", big.sum(),
          "max err/r80 %.1f" % (ratio.max() if big.any() else 0), "min r80 among them %.1e" % (r80[big].min() if big.any() else 0),
          "max err with r80<1e-13: %.1e" % err[good & (r80 < 1e-13)].max(), "n clean", (good & (r80 < 1e-13)).sum())
