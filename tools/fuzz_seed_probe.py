"""One seed of tests/test_gpu_fuzz.py::test_random_bsm_configurations, looked at closely: which walkers' verdicts differ from the oracle's,
their 80-bit residual, the oracle's per-bin residuals.  python tools/fuzz_seed_probe.py SEED"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from oracle import oracle

seed = int(sys.argv[1])
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
om = oracle.make_model(ps, "BSM_GAUSS", **dict(kw, texture=tex.name))
n = int(rng.choice([64, 700, 3000, 9000]))
box = np.array(ps.seeds, dtype=float)
th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
lo, hi = Cf.SCALE_BOUNDARIES[dim]
th[:, -1] = rng.uniform(lo, lo + rng.uniform(0.3, 1.0) * (hi - lo), n)
wild = rng.random(n) < 0.01
th[wild, rng.integers(0, len(ps), wild.sum())] = rng.choice([np.nan, np.inf, -np.inf], wild.sum())
ref, ref_fr, ref_st = oracle.lnprob_batch(om, th, want_fr=True, want_status=True)
with Model(compile_model(ps, "BSM_GAUSS", **kw)) as m:
    lp, fr, st = m.lnprob(th, want_fr=True)
r80 = oracle.unitarity_residual_batch(om, th)
inbox = (st != _lib.GF_ST_OUT_OF_PRIOR) & (ref_st != 1)
clear = ((r80 < 10 ** -7.25) | (r80 > 10 ** -6.75)) & inbox
flagged, ref_flagged = st == _lib.GF_ST_NON_UNITARY, ref_st == 2
bad = np.nonzero(clear & (flagged != ref_flagged))[0]
print("library", os.environ.get("GOLEMHIP_LIB", "(shipped)"), "seed", seed, "dim", dim, tex, "nbins", nbins, "twelve", twelve, "n", n, "mismatches", len(bad), "flagged", int(flagged.sum()), "ref", int(ref_flagged.sum()))
for i in bad[:5]:
    print(" walker", i, "device flagged", bool(flagged[i]), "oracle flagged", bool(ref_flagged[i]), "r80 %.4e" % r80[i], "theta", np.array2string(th[i], precision=17))
np.save("gpurun_out/fuzz_seed_%d_theta.npy" % seed, th[bad[:5]])
