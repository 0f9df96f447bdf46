#!/usr/bin/env python3
"""Kernel-rate probes for BASELINE config C3 (Haar draws) and chain post-processing (propagate)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model

ps = Cf.unitary_paramset()
src = np.array([1., 2., 0.]) / 3
with Model(compile_model(ps, "PRIOR_ONLY", source_ratio=src)) as m:
    for n in (10_000_000, 100_000_000):
        d_fr = m.alloc(24 * n)
        d_ang = m.alloc(32 * n)
        for with_ang in (False, True):
            for _ in range(3):
                m.haar_draw_device(26, 0, n, d_ang.ptr if with_ang else None, d_fr.ptr)
            e0, e1 = m.event(), m.event()
            m.sync(); e0.record()
            for _ in range(20):
                m.haar_draw_device(26, 0, n, d_ang.ptr if with_ang else None, d_fr.ptr)
            e1.record(); m.sync()
            ms = e0.elapsed_ms(e1) / 20
            b = 24 + (32 if with_ang else 0)
            print(json.dumps({"kernel": "haar", "n": n, "angles": with_ang, "ms": ms, "draws_per_s": n / ms * 1e3, "GBps": n * b / ms / 1e6,
                              "frac_hbm_peak": n * b / ms / 1e6 / 8000}))
        # propagate on the stored angles (4-dim theta -> fr): 32 B in, 24 B out
        for _ in range(3):
            m.propagate_device(d_ang.ptr, n, d_fr.ptr)
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(20):
            m.propagate_device(d_ang.ptr, n, d_fr.ptr)
        e1.record(); m.sync()
        ms = e0.elapsed_ms(e1) / 20
        print(json.dumps({"kernel": "propagate_sm<4>", "n": n, "ms": ms, "evals_per_s": n / ms * 1e3, "GBps": n * 56 / ms / 1e6,
                          "frac_hbm_peak": n * 56 / ms / 1e6 / 8000}))
        d_fr.free(); d_ang.free()
