#!/usr/bin/env python3
"""lnprob kernel rates for the posteriors of the other BASELINE configs (device-resident theta, 16.8 M walkers
per launch): C3 4-dim flat priors, C4 7-dim priors (the chain mc_texture.py samples), the 12-dim priors of C5,
the 6-dim notebook posterior with the fr / status outputs, the 2-dim tutorial posterior."""
import json, time
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf, fr as fr_utils      # noqa: E402
from golemflavor_amd.descriptor import compile_model           # noqa: E402
from golemflavor_amd.model import Model                        # noqa: E402

N = 4096 * 4096
ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
bf = fr_utils.angles_to_fr(ang)
_, nb = Cf.notebook_paramsets(ang)
_, tut = Cf.tutorial_paramsets(ang)
_, fr12 = Cf.fr_paramsets(6, (0.4444, 0.0))
cases = [
    ("C2 6-dim SM_GAUSS", nb, dict(mode="SM_GAUSS", bestfit_fr=bf, smearing=0.02), False, False),
    ("C2 6-dim SM_GAUSS + fr", nb, dict(mode="SM_GAUSS", bestfit_fr=bf, smearing=0.02), True, False),
    ("C2 6-dim SM_GAUSS + fr + status", nb, dict(mode="SM_GAUSS", bestfit_fr=bf, smearing=0.02), True, True),
    ("tutorial 2-dim SM_GAUSS (identity mixing)", tut, dict(mode="SM_GAUSS", bestfit_fr=bf, smearing=0.02, sm_fixed=(0, 1, 0, 0),
                                                           src_columns=(0, 1)), False, False),
    ("C3 4-dim PRIOR_ONLY", Cf.unitary_paramset(), dict(mode="PRIOR_ONLY", source_ratio=(1, 2, 0)), False, False),
    ("C4 7-dim PRIOR_ONLY", Cf.texture_paramset(6), dict(mode="PRIOR_ONLY"), False, False),
    ("C5 12-dim PRIOR_ONLY", fr12, dict(mode="PRIOR_ONLY"), False, False),
]
rng = np.random.default_rng(1)
for name, ps, kw, want_fr, want_st in cases:
    kw = dict(kw)
    mode = kw.pop("mode")
    box = np.array(ps.seeds, dtype=float)
    nd = len(ps)
    blk = rng.uniform(box[:, 0], box[:, 1], size=(1 << 20, nd))
    th = np.tile(blk, (N >> 20, 1))
    with Model(compile_model(ps, mode, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        d_out = m.alloc(8 * N)
        d_fr = m.alloc(24 * N) if want_fr else None
        d_st = m.alloc(4 * N) if want_st else None
        args = (d_th.ptr, N, d_out.ptr, d_fr.ptr if want_fr else None, d_st.ptr if want_st else None)
        for _ in range(5):
            m.lnprob_device(*args)
        t_warm = time.perf_counter()                                # from idle the chip needs tens of ms under load to settle its clock
        while time.perf_counter() - t_warm < 0.08:
            for _ in range(8):
                m.lnprob_device(*args)
            m.sync()
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(50):
            m.lnprob_device(*args)
        e1.record(); m.sync()
        ms = e0.elapsed_ms(e1) / 50
        b = 8 * nd + 8 + (24 if want_fr else 0) + (4 if want_st else 0)
        print(json.dumps({"case": name, "ndim": nd, "bytes_per_eval": b, "kernel_ms": round(ms, 4), "evals_per_s": N / ms * 1e3,
                          "GBps_algorithmic": N * b / ms / 1e6, "frac_hbm_peak": round(N * b / ms / 1e6 / 8000, 3)}), flush=True)

# the same notebook posterior from an SoA buffer ([ndim][n]): k_lnprob_sm_soa (per-lane coalesced column loads, no LDS tile)
name, ps, kw = cases[0][0], cases[0][1], dict(cases[0][2])
mode = kw.pop("mode")
box = np.array(ps.seeds, dtype=float)
blk = rng.uniform(box[:, 0], box[:, 1], size=(1 << 20, 6))
th = np.ascontiguousarray(np.tile(blk, (N >> 20, 1)).T)
with Model(compile_model(ps, mode, **kw)) as m:
    d_th = m.alloc(th.nbytes).upload(th)
    d_out = m.alloc(8 * N)
    for _ in range(3):
        m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None, layout=1)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.08:
        for _ in range(8):
            m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None, layout=1)
        m.sync()
    e0, e1 = m.event(), m.event()
    m.sync(); e0.record()
    for _ in range(20):
        m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None, layout=1)
    e1.record(); m.sync()
    ms = e0.elapsed_ms(e1) / 20
    print(json.dumps({"case": name + " (SoA input, register kernel)", "ndim": 6, "bytes_per_eval": 56, "kernel_ms": round(ms, 4),
                      "evals_per_s": N / ms * 1e3, "GBps_algorithmic": N * 56 / ms / 1e6, "frac_hbm_peak": round(N * 56 / ms / 1e6 / 8000, 3)}))
