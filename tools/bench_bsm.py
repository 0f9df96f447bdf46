#!/usr/bin/env python3
"""Kernel-rate probe for the BSM path (BASELINE configs C4 7-dim / C5 12-dim), device-resident theta."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model

WARM_MS = float(os.environ.get("GF_BENCH_WARM_MS", "60"))


def run(name, ps, n, steps=20, status=True, dim=6, tex=Texture.OET):
    rng = np.random.default_rng(1)
    box = np.array(ps.seeds, dtype=float)
    th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    th[:, -1] = rng.uniform(lo, hi - 6, n)
    desc = compile_model(ps, "BSM_GAUSS", texture=tex, dimension=dim, binning=Cf.default_bin_edges(),
                         source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(desc) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        d_out = m.alloc(8 * n)
        d_st = m.alloc(4 * n) if status else None
        for _ in range(3):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr if status else None)
        # from idle the chip needs tens of milliseconds under load to settle its clock: warm for WARM_MS before timing
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < WARM_MS:
            for _ in range(4):
                m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr if status else None)
            m.sync()
        e0, e1 = m.event(), m.event()
        m.sync()
        e0.record()
        for _ in range(steps):
            m.lnprob_device(d_th.ptr, n, d_out.ptr, None, d_st.ptr if status else None)
        e1.record()
        m.sync()
        ms = e0.elapsed_ms(e1) / steps
    nb = 20
    print(json.dumps({"case": name, "n": n, "status": status, "kernel_ms": ms, "evals_per_s": n / ms * 1e3,
                      "bin_diag_per_s": n * nb / ms * 1e3, "GBps_algorithmic": n * (8 * len(ps) + 8 + (4 if status else 0)) / ms / 1e6}))


if __name__ == "__main__":
    ps7 = Cf.texture_paramset(6)
    ps12 = Cf.fr_paramsets(6, (0.4444, 0.0))[1]
    for n in (16384, 131072, 4 * 1024 * 1024):
        for st in (True, False):
            run("C4 7-dim", ps7, n, status=st)
    for st in (True, False):
        run("C5 12-dim", ps12, 131072, status=st)
        run("C5 12-dim", ps12, 4 * 1024 * 1024, status=st)
