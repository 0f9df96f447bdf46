#!/usr/bin/env python3
"""the notebook posterior in bulk, n = 4096 k walkers resident on the device, k in {1, 16, 256, 4096}:
kernel-only rate against the HBM roofline (small launches are launch-latency-bound, the last one is the bench line)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import notebook_descriptor, synth_theta
from golemflavor_amd.model import Model
ps, bf, desc = notebook_descriptor()
with Model(desc) as m:
    for k in (1, 16, 256, 4096):
        n = 4096 * k
        th = synth_theta(ps, n, 26)
        d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8 * n)
        for _ in range(20):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None)
        reps = 2000 if k <= 256 else 300
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(reps):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, None)
        e1.record(); m.sync()
        ms = e0.elapsed_ms(e1) / reps
        print(json.dumps({"k": k, "n": n, "us_per_launch": round(1e3 * ms, 2), "evals_per_s": n / ms * 1e3,
                          "GBps_algorithmic": round(n * 56 / ms / 1e6, 1), "frac_hbm_peak": round(n * 56 / ms / 1e6 / 8000, 4)}), flush=True)
        d_th.free(); d_out.free()
