#!/usr/bin/env python3
"""Is mcmc.mcmc() deterministic run to run?  (tests/test_gpu_sampler.py::test_mcmc_driver_device_resident failed once in ~20 suites)"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golemflavor_amd import llh as llh_utils, mcmc as mcmc_utils
from common import notebook_sets
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"), allow_pickle=False))
asimov, ps = notebook_sets(g)
f = llh_utils.notebook_ln_prob(asimov, ps)
which = sys.argv[1] if len(sys.argv) > 1 else "device"
ref = None
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    np.random.seed(26)
    p0 = mcmc_utils.flat_seed(ps, nwalkers=100)
    with contextlib.redirect_stdout(io.StringIO()):
        s = mcmc_utils.mcmc(p0=p0, ln_prob=f, ndim=6, nwalkers=100, burnin=400, nsteps=1500, device_resident=(which == "device"))
    sig = (s.mean(axis=0), s.std(axis=0))
    if ref is None:
        ref = s.copy(); print(which, "rep 0 std", np.round(sig[1], 4), flush=True)
    elif not np.array_equal(ref, s):
        print(which, "rep", rep, "DIFFERS: std", np.round(sig[1], 4), "first differing row", int(np.argmax(np.any(ref != s, axis=1))), flush=True)
print(which, "done")
