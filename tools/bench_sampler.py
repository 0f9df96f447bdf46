#!/usr/bin/env python3
"""emcee-driven throughput: host-driven sampler (one launch + PCIe round trip per half-ensemble) vs the
device-resident sampler, notebook posterior (BASELINE configs C1/C2)."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf, fr as fr_utils, llh as llh_utils, mcmc as mcmc_utils


def main():
    ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
    asimov, ps = Cf.notebook_paramsets(ang)
    f = llh_utils.notebook_ln_prob(asimov, ps)
    cases = ((100, 1, 20000), (100, 256, 5000), (512, 1, 10000), (512, 256, 2000), (512, 2048, 500),
             (4096, 1, 2000), (4096, 16, 1000), (4096, 64, 400), (4096, 256, 200), (2048, 64, 500))
    for nwalkers, nchains, nsteps in cases:
        np.random.seed(26)
        p0 = np.stack([mcmc_utils.flat_seed(ps, nwalkers) for _ in range(nchains)])
        s = mcmc_utils.DeviceEnsembleSampler(nwalkers, 6, f, nchains=nchains, seed=1)
        s.run_mcmc(p0, 50, storechain=False)
        t0 = time.perf_counter()
        s.run_mcmc(None, nsteps, storechain=False)
        dt = time.perf_counter() - t0
        acc = float(np.mean(s.acceptance_fraction))
        print(json.dumps({"sampler": "device", "persist": os.environ.get("GF_SAMPLER_PERSIST", "auto"), "nwalkers": nwalkers, "nchains": nchains, "steps": nsteps,
                          "us_per_step": 1e6 * dt / nsteps, "evals_per_s": nwalkers * nchains * nsteps / dt, "acceptance": acc}))
        s.close()
    for nwalkers, nsteps in ((100, 500), (4096, 300)):
        np.random.seed(26)
        p0 = mcmc_utils.flat_seed(ps, nwalkers)
        h = mcmc_utils.EnsembleSampler(nwalkers, 6, f, seed=1)
        pos = h.run_mcmc(p0, 20)[0]
        t0 = time.perf_counter()
        h.run_mcmc(pos, nsteps)
        dt = time.perf_counter() - t0
        print(json.dumps({"sampler": "host-driven", "nwalkers": nwalkers, "nchains": 1, "steps": nsteps,
                          "us_per_step": 1e6 * dt / nsteps, "evals_per_s": nwalkers * nsteps / dt}))


if __name__ == "__main__":
    main()
