"""Does a device-to-host copy slow down while the GPU runs the C5 sampler (and the sampler while the copy runs)?  The copy: 4.7 GB through
gf_memcpy_d2h on its own model / stream, from this thread; the sampler: 256 x 512 walkers, grid kernels, no chain stored, enqueued
asynchronously first.   python tools/d2h_under_load.py"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GF_SAMPLER_CHAIN"] = "0"
import bench
from golemflavor_amd import _lib, configs as Cf, mcmc as M
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
L = _lib.lib()
m = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=0)
n = 4_718_592_000
d = m.alloc(n)
pts, nw, make, evals = bench.scan_setup("C5", 0)
jobs = [make(p, g) for g, p in enumerate(pts)]
s = M.DeviceEnsembleSampler(nw, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 200, storechain=False)

def copy():
    a = np.empty(n // 8)
    t0 = time.perf_counter()
    _lib.check(L.gf_memcpy_d2h(m._h, a.ctypes.data_as(C.c_void_p), d.ptr, a.nbytes), "d2h")
    return time.perf_counter() - t0

for rep in range(3):
    t = copy()
    print(json.dumps({"what": "copy alone", "GBps": round(n / t / 1e9, 1)}), flush=True)
    t0 = time.perf_counter(); s.run_async(None, 400, storechain=False); s.wait(); ts = time.perf_counter() - t0
    print(json.dumps({"what": "400 steps alone", "seconds": round(ts, 4)}), flush=True)
    t0 = time.perf_counter()
    import threading
    box = {}
    th = threading.Thread(target=lambda: box.update(t=copy()))
    th.start()
    s.run_async(None, 400, storechain=False); s.wait(); ts = time.perf_counter() - t0
    th.join()
    print(json.dumps({"what": "both at once", "copy_GBps": round(n / box["t"] / 1e9, 1), "400_steps_seconds": round(ts, 4)}), flush=True)
