"""Follow-up of tools/flaky_widths.py: the rare short runs whose width is far off -- what do they contain?  GPU."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from golemflavor_amd import configs as Cf, fr as fr_utils, llh as llh_utils, mcmc as mcmc_utils
ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
asimov, ps = Cf.notebook_paramsets(ang)
f = llh_utils.notebook_ln_prob(asimov, ps)
ref_std = np.array([0.0122, 0.0015, 0.0469, 1.3258, 0.0627, 0.0522])
rng = np.random.default_rng(0)
box = np.array(ps.seeds, dtype=float)
nch = 2048
p0 = rng.uniform(box[:, 0], box[:, 1], size=(nch, 100, 6))
lp0 = f(p0.reshape(-1, 6)).reshape(nch, 100)
print("start positions with lnprob = -inf (Gaussian underflow wall or outside the box): %.3f of all walkers" % np.mean(np.isneginf(lp0)))
s = mcmc_utils.DeviceEnsembleSampler(100, 6, f, nchains=nch, seed=7)
s.run_mcmc(p0, 400, storechain=False)
pos400, lnp400 = s.state
s.reset()
s.run_mcmc(None, 1500)
ch = s.chain                                   # (nch, 100, 1500, 6)
lnp_end = s.state[1]
s.close()
sd = ch.reshape(nch, -1, 6).std(axis=1) / ref_std
off = np.any(sd > 1.2, axis=1)
print("runs with a column's std more than 20 %% above the reference: %d of %d (%.2f %%)" % (off.sum(), nch, 100 * off.mean()))
stuck400 = np.isneginf(lnp400)                 # walkers still at -inf after the burn-in
print("walkers still at lnprob = -inf after 400 burn-in steps: %d in %d runs; of the %d off runs, %d have such a walker"
      % (stuck400.sum(), np.any(stuck400, axis=1).sum(), off.sum(), np.any(stuck400, axis=1)[off].sum()))
print("runs WITHOUT such a walker: max std ratio over columns and runs %.3f; runs WITH: min %.3f" % (sd[~np.any(stuck400, axis=1)].max(), sd[np.any(stuck400, axis=1)].max(axis=1).min() if np.any(stuck400) else float('nan')))
never = np.all(ch[:, :, 0, :] == ch[:, :, -1, :], axis=2)     # a walker that did not move in 1500 steps
print("walkers that never moved during the 1500 stored steps: %d (in %d runs)" % (never.sum(), np.any(never, axis=1).sum()))
w = np.argwhere(stuck400)[:5]
for c, k in w:
    print("  run %d walker %d: start %s lnprob0 %s; after burn-in %s lnprob %s; at the end lnprob %s"
          % (c, k, np.round(p0[c, k], 3).tolist(), lp0[c, k], np.round(pos400[c, k], 3).tolist(), lnp400[c, k], lnp_end[c, k]))
# how long does it take: fraction of walkers at -inf vs step
s = mcmc_utils.DeviceEnsembleSampler(100, 6, f, nchains=nch, seed=11)
s.run_mcmc(p0, 3000)
lp = s.lnprobability                           # (nch, 100, 3000)
s.close()
for t in (0, 50, 100, 200, 400, 800, 1500, 2999):
    print("  step %4d: walkers at -inf %.5f, runs with one %.4f" % (t, np.mean(np.isneginf(lp[:, :, t])), np.mean(np.any(np.isneginf(lp[:, :, t]), axis=1))))
f.close()

# ---- the walkers that never move: where are they, and what do their proposals look like? ----------------------------
f = llh_utils.notebook_ln_prob(asimov, ps)
s = mcmc_utils.DeviceEnsembleSampler(100, 6, f, nchains=nch, seed=7)
s.run_mcmc(p0, 400, storechain=False)
s.reset()
s.run_mcmc(None, 1500)
ch = s.chain
lp = s.lnprobability
acc = s.acceptance_fraction
s.close()
never = np.argwhere(np.all(ch[:, :, 0, :] == ch[:, :, -1, :], axis=2))
for c, k in never[:8]:
    others = np.delete(np.arange(100), k)
    print("run %d walker %d: position %s lnprob %.3f | ensemble: median lnprob %.3f, mean position %s | this walker's start %s"
          % (c, k, np.round(ch[c, k, 0], 4).tolist(), lp[c, k, 0], np.median(lp[c, others, -1]), np.round(ch[c, others, -1].mean(axis=0), 4).tolist(),
             np.round(p0[c, k], 4).tolist()))
    # proposals toward the bulk from there: lnprob along the segment to the ensemble mean
    seg = ch[c, k, 0][None, :] + np.linspace(0, 1, 11)[:, None] * (ch[c, others, -1].mean(axis=0) - ch[c, k, 0])[None, :]
    print("    lnprob along the segment to the ensemble mean:", np.round(f(seg), 2).tolist())
f.close()
