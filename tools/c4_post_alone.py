"""C4 at the reference's length: the post-processing alone (rows assembled on the device, nothing read back) against the link
(a plain 9.4 GB device-to-host DMA into registered memory, GPU otherwise idle / while the post-processing runs again)."""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench  # noqa: E402
from golemflavor_amd import mcmc as mcmc_utils, scan  # noqa: E402

pts, nw, make, evals = bench.scan_setup("C4", 0)
jobs = [make(p, g) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(nw, jobs[0].ndim, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 200, storechain=False)
s.reset()
s.run_mcmc(None, 1000)
m = jobs[0].f.model
nbytes = 64 * 1000 * nw * 9 * 8
d_rows = m.alloc(nbytes)
models = [j.post_model for j in jobs]
for rep in range(3):
    t0 = time.perf_counter()
    s.postprocess_rows_to_device(d_rows.ptr, models=models)
    m.sync()
    print(json.dumps({"C4 post-processing alone (rows stay on the device) s": round(time.perf_counter() - t0, 4)}), flush=True)
arena = scan.ResultArena(nbytes)
out = arena.take((64, 1000 * nw, 9))
for rep in range(3):
    t0 = time.perf_counter()
    d_rows.download(out.shape, out=out)
    dt = time.perf_counter() - t0
    print(json.dumps({"9.4 GB device -> registered host, GPU otherwise idle: GB/s": round(nbytes / dt / 1e9, 1)}), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    s.postprocess_rows(models=models, out=out)
    dt = time.perf_counter() - t0
    print(json.dumps({"post-processing + read-back into the arena s": round(dt, 4), "GB/s": round(nbytes / dt / 1e9, 1)}), flush=True)
