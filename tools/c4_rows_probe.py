import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import scan, mcmc as mcmc_utils, configs as Cf
from golemflavor_amd.enums import Texture
from golemflavor_amd.model import Model, empty_for_download
from golemflavor_amd.descriptor import compile_model
pts = scan.texture_grid(6)
jobs = [scan._TexturePoint(p, g, dimension=6, texture=Texture.OET, nwalkers=2048, device=0) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(2048, 6, [j.f for j in jobs], seed=25, stream_ids=list(range(64)))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False); s.reset(); s.run_mcmc(None, 200)
models = [j.post_model for j in jobs]
rows = None
for rep in range(4):
    t0 = time.perf_counter(); rows = None; dfree = time.perf_counter() - t0      # the previous result is freed OUTSIDE the timed call
    t0 = time.perf_counter(); rows = s.postprocess_rows(models=models); dt = time.perf_counter() - t0
    print("(freeing the previous rows: %.1f ms)" % (1e3 * dfree), end=" ")
    print("postprocess_rows -> host: %.1f ms for %.2f GB = %.1f GB/s" % (1e3 * dt, rows.nbytes / 1e9, rows.nbytes / dt / 1e9))
m0 = jobs[0].f.model
d_rows = m0.alloc(rows.nbytes)
for rep in range(2):
    t0 = time.perf_counter(); s.postprocess_rows_to_device(d_rows.ptr, models=models); dt = time.perf_counter() - t0
    print("postprocess_rows -> device: %.1f ms" % (1e3 * dt))
out = None
for rep in range(3):
    out = None
    t0 = time.perf_counter(); out = d_rows.download(rows.shape); dt = time.perf_counter() - t0
    print("plain download of the same bytes (fresh array, pages mapped chunk by chunk): %.1f ms = %.1f GB/s" % (1e3 * dt, rows.nbytes / dt / 1e9))
t0 = time.perf_counter(); x = np.empty(rows.shape); dt = time.perf_counter() - t0
print("np.empty: %.2f ms" % (1e3 * dt))
