import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from golemflavor_amd import mcmc as mcmc_utils
from golemflavor_amd.model import Model
ps, bf, desc = bench.notebook_descriptor()
rng = np.random.default_rng(26)
box = np.array(ps.seeds, dtype=np.float64)
with Model(desc) as model:
    for nw in (100,):
        for nch, steps in ((256, 4000), (512, 2000), (1024, 2000), (2048, 1000), (4096, 600)):
            p1 = rng.uniform(box[:, 0], box[:, 1], size=(nch, nw, 6))
            smp = mcmc_utils.DeviceEnsembleSampler(nw, 6, model, nchains=nch, seed=26)
            smp.run_mcmc(p1 if nch > 1 else p1[0], 100, storechain=False)
            t0 = time.perf_counter(); smp.run_async(None, steps, storechain=False); smp.wait(); dt = time.perf_counter() - t0
            print(os.environ.get("GF_SAMPLER_NO_PRODUCERS", "producers"), "walkers", nw, "chains", nch, "us/step %.2f" % (1e6 * dt / steps), "evals/s %.3e" % (nw * nch * steps / dt), flush=True)
            smp.close()
