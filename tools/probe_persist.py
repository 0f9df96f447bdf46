#!/usr/bin/env python3
"""Where does the wall time of a many-chain persistent-sampler run go?  (bench.py's emcee_driven_c1_scaling rows)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from golemflavor_amd import mcmc as mcmc_utils
from golemflavor_amd.model import Model

ps, bf, desc = bench.notebook_descriptor()
model = Model(desc, device=0)
rngp = np.random.default_rng(26)
box = np.array(ps.seeds, dtype=np.float64)
for nch, steps in ((256, 4000), (4096, 600), (4096, 600), (4096, 2400)):
    p1 = rngp.uniform(box[:, 0], box[:, 1], size=(nch, 100, 6))
    smp = mcmc_utils.DeviceEnsembleSampler(100, 6, model, nchains=nch, seed=26)
    smp.run_mcmc(p1, 100, storechain=False)
    L, lib, h = smp._L, smp._lib, smp._h
    e0, e1 = model.event(), model.event()
    t0 = time.perf_counter()
    e0.record()
    lib.check(L.gf_sampler_run(h, steps, 1, 0), "run")
    e1.record()
    t1 = time.perf_counter()
    lib.check(L.gf_sampler_sync(h), "sync")
    t2 = time.perf_counter()
    smp._check_flags()
    t3 = time.perf_counter()
    pos, lnp = smp.state
    t4 = time.perf_counter()
    print("chains %d steps %d: launch %.2f ms, sync %.2f ms, flags %.2f ms, state %.2f ms; events %.2f ms -> %.2f us/step (wall incl. state %.2f)"
          % (nch, steps, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), e0.elapsed_ms(e1),
             1e3 * e0.elapsed_ms(e1) / steps, 1e6 * (t4 - t0) / steps), flush=True)
    smp.close()
model.close()

# bench.py's own function, twice
model = Model(desc, device=0)
for rep in range(2):
    out = bench.extra_emcee(model, ps, 4096)
    print([(r["chains"], round(r["us_per_step"], 2)) for r in out["emcee_driven_c1_scaling"]["rows"]], flush=True)
model.close()
