"""Posterior widths of the notebook posterior (examples/inference.ipynb) per column: a long device chain as the reference, then the
spread of the std estimate a SHORT run gives (100 walkers, 400 burn-in + 1500 steps: tests/test_gpu_sampler.py
test_mcmc_driver_device_resident) over many seeds, device-resident and host-driven.  GPU.
Explains gpurun_out/flaky_failed.log of round 2 (device std 0.062 vs host std 0.110 on one column)."""
import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import configs as Cf, fr as fr_utils, llh as llh_utils, mcmc as mcmc_utils

ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
asimov, ps = Cf.notebook_paramsets(ang)
f = llh_utils.notebook_ln_prob(asimov, ps)
names = [p.name for p in ps]
rng = np.random.default_rng(0)
box = np.array(ps.seeds, dtype=float)
# reference: 4096 walkers x 20000 steps after 5000 of burn-in, thinned by 10
s = mcmc_utils.DeviceEnsembleSampler(4096, 6, f, seed=1)
s.run_mcmc(rng.uniform(box[:, 0], box[:, 1], size=(4096, 6)), 5000, storechain=False)
s.reset()
s.run_mcmc(None, 20000, thin=10)
ref = s.flatchain
s.close()
print("columns", names)
print("reference mean", np.round(ref.mean(axis=0), 4).tolist())
print("reference std ", np.round(ref.std(axis=0), 4).tolist())
for d in range(6):
    h, e = np.histogram(ref[:, d], bins=12)
    print("  %-14s histogram %s over [%.3f, %.3f]" % (names[d], (h / h.sum()).round(3).tolist(), e[0], e[-1]))
# short runs, device-resident: 256 seeds at once (256 chains of 100 walkers)
nch = 256
p0 = rng.uniform(box[:, 0], box[:, 1], size=(nch, 100, 6))
s = mcmc_utils.DeviceEnsembleSampler(100, 6, f, nchains=nch, seed=7)
s.run_mcmc(p0, 400, storechain=False)
s.reset()
s.run_mcmc(None, 1500)
ch = s.chain                                   # (nch, 100, 1500, 6)
s.close()
sd = ch.reshape(nch, -1, 6).std(axis=1)
mu = ch.reshape(nch, -1, 6).mean(axis=1)
print("short runs (device, %d seeds): std of the run's std / reference std, per column:" % nch)
for d in range(6):
    r = sd[:, d] / ref[:, d].std()
    print("  %-14s ratio: min %.2f  p5 %.2f  median %.2f  p95 %.2f  max %.2f   |mean - ref| / ref std: p95 %.2f max %.2f"
          % (names[d], r.min(), np.percentile(r, 5), np.median(r), np.percentile(r, 95), r.max(),
             np.percentile(np.abs(mu[:, d] - ref[:, d].mean()) / ref[:, d].std(), 95), (np.abs(mu[:, d] - ref[:, d].mean()) / ref[:, d].std()).max()))
# the burn-in: how long until a 100-walker ensemble has the reference width? (std over walkers at step t, median over seeds)
s = mcmc_utils.DeviceEnsembleSampler(100, 6, f, nchains=nch, seed=9)
s.run_mcmc(p0, 3000)
ch = s.chain
s.close()
for t in (0, 100, 200, 400, 800, 1500, 2999):
    w = ch[:, :, t, :].std(axis=1) / ref.std(axis=0)
    print("  step %4d: across-walker std / reference std, median over seeds: %s" % (t, np.round(np.median(w, axis=0), 2).tolist()))
# host-driven sampler, a few seeds
for seed in (26, 1, 2, 3):
    np.random.seed(seed)
    hp0 = mcmc_utils.flat_seed(ps, nwalkers=100)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        host = mcmc_utils.mcmc(p0=hp0, ln_prob=f, ndim=6, nwalkers=100, burnin=400, nsteps=1500, device_resident=False)
    print("host-driven seed %d: std / reference std %s" % (seed, np.round(host.std(axis=0) / ref.std(axis=0), 2).tolist()))
f.close()
