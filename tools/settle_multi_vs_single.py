"""Is the settle step slower when every chain has its own model (per-group constant pointers: vector loads) than when all
chains share one (wave-uniform pointers: scalar loads)?  256 chains x 512 walkers of ONE grid point's posterior (texture OEU,
seeded into the band so that proposals get parked), once as nchains = 256 of one model, once as a list of 256 models."""
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import configs as Cf, fr as fr_utils, llh as llh_utils, mcmc as mcmc_utils
from golemflavor_amd.enums import Texture
asimov, ps = Cf.fr_paramsets(6, fr_utils.fr_to_angles((1, 1, 1)))
args = argparse.Namespace(source_ratio=np.array([0.3, 0.7, 0.0]), dimension=6, texture=Texture.OEU, binning=Cf.default_bin_edges())
nch, nw = 256, 512
rng = np.random.default_rng(3)
box = np.array(ps.seeds, dtype=float)
p0 = rng.uniform(box[:, 0], box[:, 1], size=(nch, nw, 12))
p0[:, :, 11] = rng.uniform(-38.5, -36.5, size=(nch, nw))
for mode in ("single", "multi", "single", "multi"):
    fs = [llh_utils.bsm_ln_prob(args, asimov, ps, smearing=0.02, on_nonunitary="-inf") for _ in range(nch if mode == "multi" else 1)]
    s = mcmc_utils.DeviceEnsembleSampler(nw, 12, fs if mode == "multi" else fs[0], nchains=nch, seed=5) if mode == "single" else mcmc_utils.DeviceEnsembleSampler(nw, 12, fs, seed=5)
    s.on_nonunitary = "-inf"
    s.run_mcmc(p0, 64, storechain=False)
    t0 = time.perf_counter(); s.run_mcmc(None, 160, storechain=False); dt = time.perf_counter() - t0
    print("%-6s: %.1f us per half-step, %d non-unitary proposals" % (mode, 1e6 * dt / 320, s.nonunitary_proposals), flush=True)
    s.close()
    for f in fs:
        f.close()
