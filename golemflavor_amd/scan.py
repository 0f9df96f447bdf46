"""Grid scans of independent chains: BASELINE configs C4 (scripts/mc_texture.py) and C5 (the 12-dim
posterior of scripts/fr.py / sens.py's scale scan), sharded over one process per GPU.

The reference runs one HTCondor job per grid point (submitter/mc_texture_dag.py:57-71,
submitter/sens_dag.py:75-95).  Here grid point g runs on rank g mod N (`dist.shard`) with the
device-resident sampler; chains are gathered at the end (RCCL all-gather through `gf_comm_*`, or
gloo / nothing for one rank).  No collective on the data path.

    python -m golemflavor_amd.scan --config C4 [--nwalkers 2048 --burnin 100 --nsteps 200]
    python -m torch.distributed.run --nproc-per-node 8 ... -m golemflavor_amd.scan --config C5
"""
import argparse
import json
import os
import time

import numpy as np

from . import configs as Cf
from . import dist as gdist
from . import fr as fr_utils
from . import llh as llh_utils
from . import mcmc as mcmc_utils
from .descriptor import compile_model
from .enums import Texture
from .model import Model


def texture_grid(dimension=6, n_scales=8, n_sources=8):
    """C4: 8 logLam (linspace over SCALE_BOUNDARIES[d]) x 8 sources (x, 1-x, 0)."""
    lo, hi = Cf.SCALE_BOUNDARIES[dimension]
    return [(float(s), (float(x), float(1 - x), 0.0))
            for s in np.linspace(lo, hi, n_scales) for x in np.linspace(0, 1, n_sources)]


class _TexturePoint:
    """One grid point of scripts/mc_texture.py: flat-likelihood chain over the mixing/mass priors
    (mc_texture.py:148-170), then flux_averaged_BSMu of every sample (mc_texture.py:216-221) at this
    point's scale and source ratio.  Result: (nwalkers*nsteps, 3 + 6): fr columns then the sample."""

    def __init__(self, point, g, *, dimension, texture, nwalkers, device, seed=25):
        self.scale, self.source = point
        self.dimension, self.texture, self.device = dimension, texture, device
        self.ps6 = Cf.ParamSet(list(Cf.texture_paramset(dimension))[:6])       # scale fixed per grid point
        self.prior = llh_utils.prior_ln_prob(self.ps6, device=device)
        rng = np.random.default_rng(seed + g)
        box = np.array(self.ps6.seeds, dtype=float)
        self.p0 = rng.uniform(box[:, 0], box[:, 1], size=(nwalkers, 6))        # mcmc.flat_seed, seeded per point
        self.sampler = mcmc_utils.DeviceEnsembleSampler(nwalkers, 6, self.prior, seed=seed + g)

    def collect(self):
        samples = self.sampler.flatchain                                       # (nwalkers*nsteps, 6)
        self.sampler.close()
        self.prior.close()
        desc = compile_model(self.ps6, "BSM_GAUSS", texture=self.texture, dimension=self.dimension,
                             binning=Cf.default_bin_edges(), source_ratio=self.source, scale_fixed=self.scale,
                             bestfit_fr=(1 / 3,) * 3, smearing=0.02)
        with Model(desc, device=self.device) as m:
            frs, st = m.propagate(samples)
        frs[st != 0] = np.nan                                                  # the reference would have raised there
        return np.column_stack([frs, samples])


def sens_grid(n_scales=8, n_sources=8):
    """C5: 2 dims {3, 6} x 2 textures x 8 sources x 8 scales = 256 points."""
    pts = []
    for dim in (3, 6):
        lo, hi = Cf.SCALE_BOUNDARIES[dim]
        for tex in (Texture.OET, Texture.OUT):
            for x in np.linspace(0, 1, n_sources):
                for sc in np.linspace(lo, hi - 0.25 * (hi - lo), n_scales):
                    pts.append((dim, tex, (float(x), float(1 - x), 0.0), float(sc)))
    return pts


class _SensPoint:
    """One grid point of the 12-dim posterior (scripts/fr.py:62-104 paramset, llh.py:121-130 with the
    Gaussian substitute): logLam seeded around the point's scale; result: the flat chain (.., 12)."""

    def __init__(self, point, g, *, nwalkers, device, seed=25, smearing=0.02):
        dim, tex, source, scale = point
        inj = fr_utils.fr_to_angles((1, 1, 1))
        asimov, ps = Cf.fr_paramsets(dim, inj)
        args = argparse.Namespace(source_ratio=np.array(source), dimension=dim, texture=tex, binning=Cf.default_bin_edges())
        self.f = llh_utils.bsm_ln_prob(args, asimov, ps, smearing=smearing, device=device, on_nonunitary="-inf")
        rng = np.random.default_rng(seed + g)
        box = np.array(ps.seeds, dtype=float)
        self.p0 = rng.uniform(box[:, 0], box[:, 1], size=(nwalkers, 12))
        lo, hi = Cf.SCALE_BOUNDARIES[dim]
        self.p0[:, 11] = np.clip(rng.normal(scale, 0.5, nwalkers), lo, hi)
        self.sampler = mcmc_utils.DeviceEnsembleSampler(nwalkers, 12, self.f, seed=seed + g)

    def collect(self):
        out = self.sampler.flatchain
        self.sampler.close()
        self.f.close()
        return out


def run_points(points, indices, make, burnin, nsteps):
    """All of this rank's grid points advance together: every sampler enqueues its launches on its own
    stream (run_async) before anybody waits, so small ensembles overlap on the GPU."""
    jobs = {g: make(points[g], g) for g in indices}
    for j in jobs.values():
        j.sampler.run_async(j.p0, burnin, storechain=False)
    for j in jobs.values():
        j.sampler.wait()
        j.sampler.reset()
    for j in jobs.values():
        j.sampler.run_async(None, nsteps)
    for j in jobs.values():
        j.sampler.wait()
    return {g: j.collect() for g, j in jobs.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", choices=["C4", "C5"], default="C4")
    ap.add_argument("--nwalkers", type=int, default=None)
    ap.add_argument("--burnin", type=int, default=100)
    ap.add_argument("--nsteps", type=int, default=200)
    ap.add_argument("--points", type=int, default=None, help="only the first N grid points (smoke runs)")
    ap.add_argument("--dimension", type=int, default=6)
    ap.add_argument("--texture", default="OET")
    ap.add_argument("--outfile", default=None, help="np.save the gathered chains here (rank 0)")
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = int(os.environ.get("GF_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    backend = gdist.LocalBackend()
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
        backend = gdist.GlooBackend()

    t0 = time.perf_counter()
    if a.config == "C4":
        pts = texture_grid(a.dimension)[:a.points]
        nw = a.nwalkers or 2048
        make = lambda p, g: _TexturePoint(p, g, dimension=a.dimension, texture=Texture[a.texture], nwalkers=nw,  # noqa: E731
                                          device=device)
        evals_per_point = nw * (a.burnin + a.nsteps) + nw * a.nsteps
    else:
        pts = sens_grid()[:a.points]
        nw = a.nwalkers or 512
        make = lambda p, g: _SensPoint(p, g, nwalkers=nw, device=device)  # noqa: E731
        evals_per_point = nw * (a.burnin + a.nsteps)
    mine = gdist.shard(len(pts), backend.rank, backend.world)
    local = run_points(pts, mine, make, a.burnin, a.nsteps)
    chains = gdist.gather_chains(local, len(pts), backend)
    dt = time.perf_counter() - t0
    if rank == 0:
        arr = np.stack(chains)
        if a.outfile:
            mcmc_utils.save_chains(arr, a.outfile)
        print(json.dumps({"config": a.config, "grid_points": len(pts), "ranks": world, "walkers": nw, "burnin": a.burnin,
                          "nsteps": a.nsteps, "chains_shape": list(arr.shape), "seconds": dt,
                          "evals_per_s": len(pts) * evals_per_point / dt,
                          "finite_fraction": float(np.mean(np.isfinite(arr)))}))
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
