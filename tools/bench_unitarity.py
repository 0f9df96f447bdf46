import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
from common import BIN_EDGES, uniform_theta
n = 1 << 20
for dim, tex, twelve in ((6, Texture.OEU, True), (3, Texture.OUT, False), (6, Texture.OUT, False)):
    ps = Cf.fr_paramsets(dim, (0.4, 0.0))[1] if twelve else Cf.texture_paramset(dim)
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    rng = np.random.default_rng(3)
    th = uniform_theta(ps, n, rng, seeds=True)
    th[:, -1] = rng.uniform(lo, hi, n)
    kw = dict(dimension=dim, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1/3,)*3, smearing=0.02)
    for dec_env in ("0", "2", "3", "4"):
        os.environ["GF_UNI_BAND_DECADES"] = dec_env
        os.environ["GF_DIAGNOSTICS"] = "1"      # result-changing overrides are honoured only with this set
        with Model(compile_model(ps, "BSM_GAUSS", texture=tex, **kw)) as m:
            d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8*n); d_st = m.alloc(4*n)
            for rep in range(2):
                m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr); m.sync()
            t0 = time.perf_counter()
            for rep in range(3):
                m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr)
            m.sync(); dt = (time.perf_counter() - t0) / 3
            st = d_st.download((n,), dtype=np.int32)
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None); m.sync()
            t0 = time.perf_counter()
            for rep in range(3):
                m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None)
            m.sync(); dt0 = (time.perf_counter() - t0) / 3
        print(dim, tex.name, 'ndim', len(ps), 'band', dec_env, 'with status %.3f ms (%.2e evals/s)' % (1e3*dt, n/dt), 'no status %.3f ms' % (1e3*dt0), 'nonunitary frac %.4f' % np.mean(st == 2), flush=True)
