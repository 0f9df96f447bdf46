#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point gf_lnprob_batch (H2D + kernel + D2H, pinned staging)."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import notebook_descriptor, synth_theta
from golemflavor_amd.model import Model
ps, bf, desc = notebook_descriptor()
with Model(desc) as m:
    for n in (50, 100, 512, 2048, 4096, 65536, 1 << 20, 1 << 21, 1 << 22, 1 << 24):
        th = synth_theta(ps, n, 1)
        for _ in range(3):
            m.lnprob(th, want_status=False)
        reps = 200 if n <= 65536 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            m.lnprob(th, want_status=False)
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"zerocopy": "off" if os.environ.get("GF_NO_ZEROCOPY") else "auto", "n": n, "us_per_call": dt * 1e6, "evals_per_s": n / dt, "GBps_over_pcie": n * 56 / dt / 1e9}))
