"""Rates of the wide SM instances (12 columns with the flavor likelihood; 12 columns from an SoA buffer; 7 columns generic):
the ones that spilled at four waves per SIMD.  16.8 M walkers per launch, device resident."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf, fr as fr_utils
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
BIN_EDGES = Cf.default_bin_edges()
N = 4096 * 4096
_, fr12 = Cf.fr_paramsets(6, (0.4444, 0.0))
rng = np.random.default_rng(1)
box = np.array(fr12.seeds, dtype=float)
blk = rng.uniform(box[:, 0], box[:, 1], size=(1 << 20, 12))
for name, mode, kw, layout in (("12 columns SM_GAUSS (no_bsm), AoS", "BSM_GAUSS", dict(no_bsm=True, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02, dimension=6), 0),
                               ("12 columns SM_GAUSS (no_bsm), SoA", "BSM_GAUSS", dict(no_bsm=True, binning=BIN_EDGES, source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02, dimension=6), 1),
                               ("12 columns PRIOR_ONLY, AoS", "PRIOR_ONLY", {}, 0),
                               ("12 columns PRIOR_ONLY, SoA", "PRIOR_ONLY", {}, 1)):
    th = np.tile(blk, (N >> 20, 1))
    if layout == 1:
        th = np.ascontiguousarray(th.T)
    with Model(compile_model(fr12, mode, **kw)) as m:
        d_th = m.alloc(th.nbytes).upload(th); d_out = m.alloc(8 * N)
        for _ in range(5):
            m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None, layout=layout)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.08:
            for _ in range(8):
                m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None, layout=layout)
            m.sync()
        e0, e1 = m.event(), m.event()
        m.sync(); e0.record()
        for _ in range(50):
            m.lnprob_device(d_th.ptr, N, d_out.ptr, None, None, layout=layout)
        e1.record(); m.sync()
        ms = e0.elapsed_ms(e1) / 50
        print(json.dumps({"lib": os.path.basename(os.environ.get("GOLEMHIP_LIB", "libgolemhip.so")), "case": name, "kernel_ms": round(ms, 4), "evals_per_s": N / ms * 1e3,
                          "frac_hbm_peak": round(N * 104 / ms / 1e6 / 8000, 3)}), flush=True)
