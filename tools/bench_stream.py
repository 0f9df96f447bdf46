#!/usr/bin/env python3
"""The bench kernel next to the same stream with almost no arithmetic (6-dim PRIOR_ONLY: same 48 B in / 8 B out per
walker, same tiles): how much of the streaming ceiling the physics costs.  Honours GOLEMHIP_LIB (A/B of variants)."""
import json
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
rng = np.random.default_rng(1)
box = np.array(nb.seeds, dtype=float)
th = np.tile(rng.uniform(box[:, 0], box[:, 1], size=(1 << 20, 6)), (N >> 20, 1))
res = {}
for name, mode, kw in (("sm_gauss", "SM_GAUSS", dict(bestfit_fr=bf, smearing=0.02)), ("prior_only", "PRIOR_ONLY", {})):
    with Model(compile_model(nb, mode, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        d_out = m.alloc(8 * N)
        for _ in range(10):
            m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None)
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(100):
            m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None)
        e1.record(); m.sync()
        res[name] = e0.elapsed_ms(e1) / 100
print(json.dumps({"lib": os.path.basename(os.environ.get("GOLEMHIP_LIB", "base")), "sm_gauss_ms": round(res["sm_gauss"], 4),
                  "prior_only_ms": round(res["prior_only"], 4), "sm_GBps": round(N * 56 / res["sm_gauss"] / 1e6),
                  "prior_GBps": round(N * 56 / res["prior_only"] / 1e6)}))
