#!/usr/bin/env python3
"""Soak: long device-resident chains (both launch shapes) and a long bench loop; checks invariants, prints rates."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from golemflavor_amd import configs as Cf, fr as fr_utils, llh as llh_utils, mcmc as mcmc_utils

ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
asimov, ps = Cf.notebook_paramsets(ang)
f = llh_utils.notebook_ln_prob(asimov, ps)
box = np.array(ps.ranges, dtype=float)
np.random.seed(26)
for shape, nwalkers, nsteps in (("1", 100, 1_000_000), ("0", 100, 200_000), ("0", 4096, 100_000)):
    os.environ["GF_SAMPLER_PERSIST"] = shape
    p0 = mcmc_utils.flat_seed(ps, nwalkers)
    s = mcmc_utils.DeviceEnsembleSampler(nwalkers, 6, f, seed=3)
    s.run_mcmc(p0, 1000, storechain=False)
    s.reset()
    t0 = time.perf_counter()
    s.run_mcmc(None, nsteps, thin=max(1, nsteps // 1000))
    dt = time.perf_counter() - t0
    ch, lp = s.chain, s.lnprobability
    ok = bool(np.all(ch >= box[:, 0]) and np.all(ch <= box[:, 1]) and np.all(np.isfinite(lp)))
    again = f.model.lnprob(ch.reshape(-1, 6), want_status=False).reshape(lp.shape)
    print(json.dumps({"shape": "persistent" if shape == "1" else "grid", "nwalkers": nwalkers, "steps": nsteps, "seconds": round(dt, 3),
                      "us_per_step": round(1e6 * dt / nsteps, 3), "acceptance": round(float(s.acceptance_fraction.mean()), 4),
                      "in_box_and_finite": ok, "stored_lnprob_bitwise_reproducible": bool(np.array_equal(again, lp)),
                      "posterior_mean": [round(float(x), 4) for x in ch.reshape(-1, 6).mean(axis=0)]}), flush=True)
    s.close()
f.close()
