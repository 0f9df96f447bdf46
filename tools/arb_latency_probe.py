import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from common import BIN_EDGES, uniform_theta
from golemflavor_amd import _lib, configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
L = _lib.lib()
L.gf_internal_uni_residuals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
ps = Cf.fr_paramsets(6, (0.4, 0.0))[1]
rng = np.random.default_rng(0)
n = 4096
th = uniform_theta(ps, n, rng, seeds=True); th[:, -1] = rng.uniform(-40, -32, n)
kw = dict(dimension=6, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
with Model(compile_model(ps, "BSM_GAUSS", texture=Texture.OEU, **kw)) as m:
    d_th = m.alloc(th.nbytes).upload(th)
    for npairs in (1, 42, 1000, 5000, 20000, 21504, 43008, 200000):
        w = np.arange(npairs, dtype=np.int64) % n
        b = (np.arange(npairs) % 20).astype(np.int32)
        d_w, d_b, d_o = m.alloc(w.nbytes).upload(w), m.alloc(b.nbytes).upload(b), m.alloc(8 * npairs)
        for which in (1,):
            L.gf_internal_uni_residuals(m._h, d_th.ptr, 0, n, d_w.ptr, d_b.ptr, npairs, which, d_o.ptr)
            t0 = time.perf_counter()
            for _ in range(5):
                L.gf_internal_uni_residuals(m._h, d_th.ptr, 0, n, d_w.ptr, d_b.ptr, npairs, which, d_o.ptr)
            dt = (time.perf_counter() - t0) / 5
            print("pairs %7d (setup + 1 bin each, 3 lanes per pair, grid 512 x 128): %.1f us per call" % (npairs, 1e6 * dt))

    # where one (walker, bin)'s chain goes: shader-clock cycles of the walker's terms (angles_to_u of the sampled mixing angles,
    # the two sandwiches, 10^x) and of one bin (cardano_eqn + test_unitarity), a single pair alone on the GPU and a full grid
    for npairs in (1, 21504):
        w = np.arange(npairs, dtype=np.int64) % n
        b = (np.arange(npairs) % 20).astype(np.int32)
        d_w, d_b, d_o = m.alloc(w.nbytes).upload(w), m.alloc(b.nbytes).upload(b), m.alloc(8 * npairs)
        for which, label in ((101, "NINE lanes: walker terms"), (102, "NINE lanes: one bin"), (110, "NINE lanes: terms + bin, wall ns"), (2, "walker terms"), (3, "one bin"), (4, " bin: scaling + H"), (5, " bin: tr, tr H^2, det"), (6, " bin: Q, R, sqrt, /, arccos"),
                             (7, " bin: sqrt Q, cos, eigenvalue"), (8, " bin: eigenvector"), (9, " bin: |X X^+|, sums"), (10, "clock64 counts per wall-clock ns"), (11, "terms + bin, wall-clock ns")):
            L.gf_internal_uni_residuals(m._h, d_th.ptr, 0, n, d_w.ptr, d_b.ptr, npairs, which, d_o.ptr)
            L.gf_internal_uni_residuals(m._h, d_th.ptr, 0, n, d_w.ptr, d_b.ptr, npairs, which, d_o.ptr)
            cyc = d_o.download((npairs,))
            print("pairs %6d  %-30s: median %.0f cycles (min %.0f, max %.0f) = %.1f us at 2.4 GHz" % (npairs, label, np.median(cyc), cyc.min(), cyc.max(), np.median(cyc) / 2400.0))
