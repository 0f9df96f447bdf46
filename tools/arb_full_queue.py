"""The arbitration kernel on a FULL queue, for profiling: the bench's failing-region workload (texture OEU, d = 6, 12 columns,
logLam over its whole range: 20 % of walkers non-unitary, ~80 000 walkers with ~280 000 undecided bins per 1 M-walker launch)
evaluated with status ten times.  Run under rocprofv3 (profiles/run_profile_arbitration.sh)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from common import BIN_EDGES, uniform_theta
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

n = 1 << 20
ps = Cf.fr_paramsets(6, (0.4, 0.0))[1]
lo, hi = Cf.SCALE_BOUNDARIES[6]
rng = np.random.default_rng(1)
th = uniform_theta(ps, n, rng, seeds=True)
th[:, -1] = rng.uniform(lo, hi, n)
kw = dict(dimension=6, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
with Model(compile_model(ps, "BSM_GAUSS", texture=Texture.OEU, **kw)) as m:
    d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8 * n); d_st = m.alloc(4 * n)
    for rep in range(12):
        m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr)
    m.sync()
    st = d_st.download((n,), dtype=np.int32)
print("non-unitary fraction %.4f" % np.mean(st == 2))
