#!/usr/bin/env python3
"""One model, 4 194 304 walkers: no-status and with-status timings alternated (does the order of measurement matter?)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

n = 4 * 1024 * 1024
for name, ps in (("C4 7-dim", Cf.texture_paramset(6)), ("C5 12-dim", Cf.fr_paramsets(6, (0.4444, 0.0))[1])):
    rng = np.random.default_rng(1)
    box = np.array(ps.seeds, dtype=float)
    th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
    lo, hi = Cf.SCALE_BOUNDARIES[6]
    th[:, -1] = rng.uniform(lo, hi - 6, n)
    desc = compile_model(ps, "BSM_GAUSS", texture=Texture.OET, dimension=6, binning=Cf.default_bin_edges(),
                         source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(desc) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        d_out, d_st = m.alloc(8 * n), m.alloc(4 * n)
        for rnd in range(3):
            for st in (None, d_st.ptr):
                for _ in range(3):
                    m.lnprob_device(d_th.ptr, n, d_out.ptr, None, st)
                e0, e1 = m.event(), m.event()
                m.sync(); e0.record()
                for _ in range(20):
                    m.lnprob_device(d_th.ptr, n, d_out.ptr, None, st)
                e1.record(); m.sync()
                ms = e0.elapsed_ms(e1) / 20
                stv = d_st.download((n,), dtype=np.int32) if st else None
                print(name, "round", rnd, "status" if st else "plain ", "%.4f ms %.3fe9/s" % (ms, n / ms / 1e6),
                      "" if stv is None else "status counts %s" % dict(zip(*np.unique(stv, return_counts=True))), flush=True)
