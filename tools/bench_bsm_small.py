#!/usr/bin/env python3
"""Latency of a small BSM batch (a host-driven emcee half-ensemble): kernel time on the device and the full
host-buffer call, 12-column posterior with status.  GF_BSM_LPW=1 forces one lane per walker (A/B)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model
_, ps = Cf.fr_paramsets(6, (0.4444, 0.0))
rng = np.random.default_rng(1)
box = np.array(ps.seeds, dtype=float)
desc = compile_model(ps, "BSM_GAUSS", texture=Texture.OET, dimension=6, binning=Cf.default_bin_edges(), source_ratio=(0., 1., 0.),
                     bestfit_fr=(1 / 3,) * 3, smearing=0.02)
with Model(desc) as m:
    for n in (30, 128, 256, 1024, 4096, 32768):
        th = rng.uniform(box[:, 0], box[:, 1], size=(n, 12)); th[:, 11] = rng.uniform(-56, -40, n)
        d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8 * n); d_st = m.alloc(4 * n)
        for _ in range(5):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr)
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(200):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr)
        e1.record(); m.sync()
        k_us = 1e3 * e0.elapsed_ms(e1) / 200
        for _ in range(5):
            m.lnprob(th)
        t0 = time.perf_counter()
        for _ in range(200):
            m.lnprob(th)
        h_us = 1e6 * (time.perf_counter() - t0) / 200
        print(json.dumps({"lpw": os.environ.get("GF_BSM_LPW", "auto"), "n": n, "kernel_us": round(k_us, 2), "host_call_us": round(h_us, 2)}), flush=True)
