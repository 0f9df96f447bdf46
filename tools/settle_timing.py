"""Where a settle launch's time goes (diagnostics build: tools/build_variants.sh "settle_timing:-DGF_SETTLE_TIMING", then
GOLEMHIP_LIB=variants/settle_timing.so python tools/settle_timing.py): the C5 scan's sampler, 100 + 200 steps."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import _lib, scan, mcmc as mcmc_utils
L = _lib.lib()
pts = scan.sens_grid()
jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False)
out = (C.c_uint64 * 14)()
L.gf_internal_settle_timing(out, 1)
t0 = time.perf_counter(); s.run_mcmc(None, 200, storechain=False); dt = time.perf_counter() - t0
L.gf_internal_settle_timing(out, 0)
n = max(int(out[5]), 1)
print("%d settle launches with work in 400 half-steps (%.1f us per half-step overall); parked walkers %.0f per launch" % (out[5], 1e6 * dt / 400, out[10] / n))
for i, label in ((6, "latest end of a walker's terms"), (7, "latest end of a bin"), (8, "latest completed walker"), (9, "latest block exit")):
    print("   %-32s %.1f us after the earliest kernel entry (mean over the launches)" % (label, out[i] / n / 100.0))
print("   shader-clock counts: terms %.0f, terms + bin %.0f per launch -> %.2f / %.2f counts per wall-clock ns" % (out[12] / n, out[13] / n, out[12] / n / (out[6] / n * 10.0), out[13] / n / (out[7] / n * 10.0)))
