"""One walker through the unitarity tiers: status alone, status inside its batch, the fp64 estimate (GF_UNI_DUMP, child
process) and the emulated-x87 residual of every bin."""
import os, sys, subprocess
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from common import BIN_EDGES, uniform_theta
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
dim, tex = 6, Texture.OEU
ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1]
lo, hi = Cf.SCALE_BOUNDARIES[dim]
rng = np.random.default_rng(1000 + dim)
n = 60000
th = uniform_theta(ps, n, rng, seeds=True)
th[:, -1] = rng.uniform(lo, hi, n)
i = 9233
kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
    if os.environ.get("GF_UNI_DUMP"):
        fr, st = m.propagate(th[i:i + 1])
        print("child: estimate (max over bins, real scale) %.4e  overrides %s" % (fr[0, 0], _lib.diagnostic_overrides()))
        sys.exit(0)
    from test_gpu_unitarity_r3 import _residuals
    print("alone:", m.lnprob(th[i:i + 1])[1], " in a batch of 64:", m.lnprob(th[i - 10:i + 54])[1][10], " in the batch of 60000:", m.lnprob(th)[1][i])
    print("repeated x8:", m.lnprob(np.repeat(th[i:i + 1], 8, axis=0))[1])
    d_th = m.alloc(th.nbytes).upload(th)
    for which in (0, 1):
        r = _residuals(m, d_th, n, np.full(20, i), np.arange(20), which)
        print("residuals (%s):" % ("serial", "three-lane")[which], " ".join("%.3e" % v for v in r))
env = dict(os.environ, GF_DIAGNOSTICS="1", GF_UNI_DUMP="1", GF_UNI_NO_WEIGHT_GATE="1")
print(subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True).stdout)
