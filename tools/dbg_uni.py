import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from oracle import oracle
seed=8
rng = np.random.default_rng(7000 + seed)
dim = int(rng.integers(3, 9))
tex = [Texture.OEU, Texture.OET, Texture.OUT][int(rng.integers(0, 3))]
src = rng.dirichlet((1, 1, 1)) if rng.random() < 0.5 else np.eye(3)[int(rng.integers(0, 3))]
nbins = int(rng.choice([1, 2, 5, 20, 33, 64]))
lo_e, hi_e = 10 ** rng.uniform(4, 5), 10 ** rng.uniform(6, 7.5)
edges = np.logspace(np.log10(lo_e), np.log10(hi_e), nbins + 1)
twelve = rng.random() < 0.5
ps = Cf.fr_paramsets(dim, (0.4444, 0.0))[1] if twelve else Cf.texture_paramset(dim)
bf = tuple(rng.dirichlet((3, 3, 3)))
kw = dict(texture=tex, dimension=dim, binning=edges, source_ratio=src, bestfit_fr=bf, smearing=float(rng.choice([0.02, 0.2])))
n = int(rng.choice([64, 700, 3000, 9000]))
box = np.array(ps.seeds, dtype=float)
th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
lo, hi = Cf.SCALE_BOUNDARIES[dim]
th[:, -1] = rng.uniform(lo, lo + rng.uniform(0.3, 1.0) * (hi - lo), n)
print('dim', dim, tex, 'nbins', nbins, 'n', n, 'ndim', len(ps), 'edges', edges)
om = oracle.make_model(ps, "BSM_GAUSS", **dict(kw, texture=tex.name))
r80 = oracle.unitarity_residual_batch(om, th)
ref, ref_st = oracle.lnprob_batch(om, th, want_status=True)
for dec_env in (None, "12"):
    if dec_env: os.environ["GF_UNI_BAND_DECADES"] = dec_env
    with Model(compile_model(ps, "BSM_GAUSS", **kw)) as m:
        lp, st = m.lnprob(th)
    mis = np.flatnonzero((st == 2) != (ref_st == 2))
    print('band', dec_env, 'mismatches', [(int(i), '%.4g' % r80[i], int(st[i]), int(ref_st[i]), th[i, -1]) for i in mis[:8]])
os.environ.pop("GF_UNI_BAND_DECADES")
os.environ["GF_UNI_DUMP"] = "1"
with Model(compile_model(ps, "BSM_GAUSS", **kw)) as m:
    fr, st = m.propagate(th)
for i in mis[:8]:
    print(i, 'est', fr[i, 0], 'r80', r80[i])
