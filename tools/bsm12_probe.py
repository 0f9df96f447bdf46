"""Why does the 12-column flux-averaged kernel issue at 0.78-0.83 of the fp64 rate when the 7-column one reaches 0.86?  The 12-column
instance with the mixing angles SAMPLED (the C5 posterior) against the same 12 columns with one oscillation parameter renamed, so that
the rule of fr.py:425-435 takes NuFIT mixing (the per-walker complex U_SM and its invariants drop out, the tile stays 96 B wide)."""
import os, sys, time, copy
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import configs as Cf, fr as fr_utils
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from golemflavor_amd.param import ParamSet
N = 1 << 22
_, ps = Cf.fr_paramsets(6, fr_utils.fr_to_angles((1, 1, 1)))
ps_fixed = ParamSet([copy.deepcopy(p) for p in ps])
ps_fixed[3].name = "dcp_renamed"          # not all six oscillation parameters scanned any more -> NuFIT mixing, default masses
rng = np.random.default_rng(1)
box = np.array(ps.seeds, dtype=float)
th = rng.uniform(box[:, 0], box[:, 1], size=(N, 12)); th[:, 11] = rng.uniform(-50, -42, N)
kw = dict(texture=Texture.OET, dimension=6, binning=Cf.default_bin_edges(), source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
for rep in range(2):
    for label, p in (("12 columns, mixing sampled", ps), ("12 columns, NuFIT mixing  ", ps_fixed), ("7 columns                 ", None)):
        if p is None:
            p = Cf.texture_paramset(6); t7 = np.ascontiguousarray(np.concatenate([th[:, :6], th[:, 11:12]], axis=1)); data = t7
        else:
            data = th
        with Model(compile_model(p, "BSM_GAUSS", **kw)) as m:
            d = m.alloc(data.nbytes).upload(data); o = m.alloc(8 * N)
            for _ in range(20): m.lnprob_device(d.ptr, N, o.ptr, None, None)
            m.sync(); e0, e1 = m.event(), m.event(); e0.record()
            for _ in range(30): m.lnprob_device(d.ptr, N, o.ptr, None, None)
            e1.record(); m.sync(); ms = e0.elapsed_ms(e1) / 30
            print("%s  %.1f us per 4.19 M walkers = %.3e evals/s" % (label, 1e3 * ms, N / ms * 1e3), flush=True)
