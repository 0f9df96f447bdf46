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
    for nw in (100, 512):
        for nch, steps in ((1, 20000), (16, 20000), (256, 4000), (4096, 600)):
            if nw == 512 and nch == 4096: continue
            p1 = rng.uniform(box[:, 0], box[:, 1], size=(nch, nw, 6))
            smp = mcmc_utils.DeviceEnsembleSampler(nw, 6, model, nchains=nch, seed=26)
            smp.run_mcmc(p1 if nch > 1 else p1[0], 100, storechain=False)
            t0 = time.perf_counter(); smp.run_mcmc(None, steps, storechain=False); dt = time.perf_counter() - t0
            print(os.environ.get("GF_SAMPLER_NO_PRODUCERS", "producers"), "walkers", nw, "chains", nch, "us/step %.2f" % (1e6 * dt / steps), "evals/s %.3e" % (nw * nch * steps / dt), flush=True)
            smp.close()
