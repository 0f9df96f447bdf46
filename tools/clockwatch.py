#!/usr/bin/env python3
"""Sample the GPU's clocks and power (rocm-smi) while a kernel loop runs: does the fp64 physics pull the clocks
down compared with the prior-only stream of the same tiles?   python tools/clockwatch.py sm_gauss|prior_only"""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf, fr as fr_utils      # noqa: E402
from golemflavor_amd.descriptor import compile_model           # noqa: E402
from golemflavor_amd.model import Model                        # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sm_gauss"
N = 4096 * 4096
ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
bf = fr_utils.angles_to_fr(ang)
_, nb = Cf.notebook_paramsets(ang)
rng = np.random.default_rng(1)
box = np.array(nb.seeds, dtype=float)
th = np.tile(rng.uniform(box[:, 0], box[:, 1], size=(1 << 20, 6)), (N >> 20, 1))
kw = dict(bestfit_fr=bf, smearing=0.02) if which == "sm_gauss" else {}
BYTES = 56
if which == "bsm":                     # the 12-dim BSM posterior (C5), 4 M walkers per launch, no status
    from golemflavor_amd.enums import Texture
    _, nb = Cf.fr_paramsets(6, (0.4444, 0.0))
    N = 1 << 22
    BYTES = 104
    box = np.array(nb.seeds, dtype=float)
    th = np.tile(rng.uniform(box[:, 0], box[:, 1], size=(1 << 20, 12)), (N >> 20, 1))
    th[:, 11] = rng.uniform(-56, -36, N)
    kw = dict(bestfit_fr=(1 / 3,) * 3, smearing=0.02, texture=Texture.OET, dimension=6, binning=Cf.default_bin_edges(),
              source_ratio=(0., 1., 0.))
samples = []
stop = False


def watch():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            card = d[sorted(d)[0]]
            samples.append({k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "power"))})
        except Exception as exc:       # noqa: BLE001
            samples.append({"error": str(exc)[:200]})
        time.sleep(0.5)


MODE = {"sm_gauss": "SM_GAUSS", "prior_only": "PRIOR_ONLY", "bsm": "BSM_GAUSS"}[which]
with Model(compile_model(nb, MODE, **kw)) as m:
    d_th = m.alloc(th.nbytes).upload(th)
    d_out = m.alloc(8 * N)
    t = threading.Thread(target=watch, daemon=True)
    t.start()
    t_end = time.time() + 8.0
    e0, e1 = m.event(), m.event()
    n = 0
    m.sync(); e0.record()
    while time.time() < t_end:
        reps = 200 if which != "bsm" else 40
        for _ in range(reps):
            m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None)
        m.sync()
        n += reps
    e1.record(); m.sync()
    stop = True
    ms = e0.elapsed_ms(e1) / n
print(json.dumps({"kernel": which, "kernel_ms": round(ms, 4), "evals_per_s": N / ms * 1e3, "GBps": round(N * BYTES / ms / 1e6), "samples": samples[2:8]}, indent=0))
