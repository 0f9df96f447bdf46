"""Host-side phases of the C5 scan's sampler: where the milliseconds beside the kernels go."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import scan, mcmc as mcmc_utils
pts = scan.sens_grid()
for rep in range(3):
    t = [time.perf_counter()]
    jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]; t.append(time.perf_counter())
    s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs)))); t.append(time.perf_counter())
    s.on_nonunitary = "-inf"
    p0 = np.stack([j.p0 for j in jobs]); t.append(time.perf_counter())
    s.run_mcmc(p0, 100, storechain=False); t.append(time.perf_counter())
    s.reset(); t.append(time.perf_counter())
    c = s.run_mcmc_to_host(None, 200); t.append(time.perf_counter())
    s.close(); [j.close() for j in jobs]; t.append(time.perf_counter())
    names = ["make 256 points", "create sampler", "stack p0", "burn-in 100 (set_state + run + sync + state)", "reset", "run_to_host 200", "close"]
    print("rep %d: " % rep + "; ".join("%s %.1f ms" % (n, 1e3 * (b - a)) for n, a, b in zip(names, t, t[1:])), flush=True)
    del c

# the stored run, enqueue and wait apart
jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False)
for rep in range(3):
    s.reset()
    t0 = time.perf_counter(); s.run_async(None, 200); t1 = time.perf_counter(); s.wait(); t2 = time.perf_counter()
    print("stored run of 200 steps: enqueue %.1f ms, wait %.1f ms (400 half-steps: %.1f us each)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e6 * (t2 - t0) / 400))
t0 = time.perf_counter(); s.run_async(None, 200, storechain=False); t1 = time.perf_counter(); s.wait(); t2 = time.perf_counter()
print("unstored run of 200 steps: enqueue %.1f ms, wait %.1f ms (%.1f us per half-step)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e6 * (t2 - t0) / 400))
