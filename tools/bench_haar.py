#!/usr/bin/env python3
"""C3: k_haar alone (1e7 draws per launch, composition only / with the angle blob), for timing and PMC passes."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = 10_000_000
with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=np.array([1., 2., 0.]) / 3)) as m:
    d_fr, d_ang = m.alloc(24 * n), m.alloc(32 * n)
    for with_ang in (False, True):
        for _ in range(10):
            m.haar_draw_device(26, 0, n, d_ang.ptr if with_ang else None, d_fr.ptr)
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(reps):
            m.haar_draw_device(26, 0, n, d_ang.ptr if with_ang else None, d_fr.ptr)
        e1.record(); m.sync()
        ms = e0.elapsed_ms(e1) / reps
        b = 24 + (32 if with_ang else 0)
        print(json.dumps({"kernel": "k_haar", "n": n, "angles": with_ang, "us": 1e3 * ms, "draws_per_s": n / ms * 1e3,
                          "frac_hbm_peak": n * b / ms / 1e6 / 8000}), flush=True)
