"""Host-side cost of standing up one grid point (descriptor, gf_model_create, close): the part of a scan
that is not sampling.  python tools/bench_setup.py  (GPU box)"""
import argparse
import cProfile
import json
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from golemflavor_amd import configs as Cf, fr as fr_utils, llh as llh_utils      # noqa: E402
from golemflavor_amd import scan                                                   # noqa: E402
from golemflavor_amd.descriptor import compile_model                               # noqa: E402
from golemflavor_amd.model import Model                                            # noqa: E402
from golemflavor_amd.enums import Texture                                          # noqa: E402

n = 64
inj = fr_utils.fr_to_angles((1, 1, 1))
t0 = time.perf_counter()
sets = [Cf.fr_paramsets(6, inj) for _ in range(n)]
t_ps = (time.perf_counter() - t0) / n
asimov, ps = sets[0]
t0 = time.perf_counter()
descs = [compile_model(ps, "BSM_GAUSS", texture=Texture.OET, dimension=6, binning=Cf.default_bin_edges(),
                       source_ratio=(1, 0, 0), bestfit_fr=(1 / 3,) * 3, smearing=0.02) for _ in range(n)]
t_desc = (time.perf_counter() - t0) / n
Model(descs[0]).close()
t0 = time.perf_counter()
models = [Model(d) for d in descs]
t_create = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for m in models:
    m.close()
t_close = (time.perf_counter() - t0) / n
pts = scan.sens_grid()[:n]
t0 = time.perf_counter()
jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
t_point = (time.perf_counter() - t0) / n
print(json.dumps({"per_point_ms": {"paramsets": t_ps * 1e3, "compile_model": t_desc * 1e3, "gf_model_create": t_create * 1e3,
                                   "close": t_close * 1e3, "_SensPoint": t_point * 1e3}}))
pr = cProfile.Profile()
pr.enable()
jobs2 = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
