"""Grid scans of independent chains: BASELINE configs C4 (scripts/mc_texture.py) and C5 (the 12-dim
posterior of scripts/fr.py / sens.py's scale scan), sharded over one process per GPU.

The reference runs one HTCondor job per grid point (submitter/mc_texture_dag.py:57-71,
submitter/sens_dag.py:75-95).  Here grid point g runs on rank (g + g div N) mod N (`dist.owner`), and all grid points
of a rank are stacked into ONE device-resident sampler -- one ensemble per posterior, one launch per
half-step for all of them.  No collective on the data path.  The results are DELIVERED the way the reference's
jobs deliver theirs -- N writers, no funnel (golemflavor/mcmc.py:108-126): on one node every rank reads its own chains back
over its own PCIe link into one host segment that rank 0 maps too (`SharedHostGather`, dist.HostSegment), or, with
--datadir, writes its own files.  Where the ranks do not share a node the chain blocks go GPU to GPU onto rank 0
(`DeviceGather`: `gf_comm_gather` over RCCL / xGMI, or hipIpc) and cross PCIe once there.  The control
plane (rendezvous, the RCCL id, barriers) is `dist.SocketBackend` -- no PyTorch in the process; any launcher that sets
RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT works, torch.distributed.run included.

    python -m golemflavor_amd.scan --config C4 [--nwalkers 2048 --burnin 100 --nsteps 200]
    python -m torch.distributed.run --nproc-per-node 8 ... -m golemflavor_amd.scan --config C5
"""
import argparse
import json
import os
import time

import numpy as np

from . import _lib
from . import configs as Cf
from . import dist as gdist
from . import fr as fr_utils
from . import llh as llh_utils
from . import mcmc as mcmc_utils
from .descriptor import compile_model
from .enums import Texture
from .model import Model, Prefaulted, empty_hugepages


def _patched(template, **fields):
    """A copy of a compiled descriptor with some of its plain fields replaced.  The grid points of a scan differ from one
    another in the source ratio, the texture and the fixed scale only -- fields `compile_model` copies through without
    deriving anything from them -- so a scan compiles one descriptor per (paramset, dimension) and patches it per point
    (50 us per point saved; `tests/test_host_logic.py` holds the patched copy to the freshly compiled one byte for byte)."""
    d = type(template).from_buffer_copy(template)
    for name, value in fields.items():
        if name == "source_ratio":
            for k in range(3):
                d.source_ratio[k] = float(value[k])
        elif name == "texture":
            d.texture = value.value if isinstance(value, Texture) else int(value)
            if d.texture == Texture.NONE.value:               # compile_model checks the NP-angle columns for that one
                raise ValueError("Texture.NONE is compiled, not patched")
        elif name == "scale_fixed":
            d.scale_fixed = float(value)
        else:
            raise KeyError(name)
    return d


_BIN_EDGES = Cf.default_bin_edges()


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
        self.ps6, box, prior_desc, post_desc = self.descriptors(point, dimension, texture)
        self.f = llh_utils.LnProb(prior_desc, device=device)   # llh.prior_ln_prob(ps6): mc_texture.py:161-170
        rng = np.random.default_rng(seed + g)
        self.p0 = rng.uniform(box[:, 0], box[:, 1], size=(nwalkers, 6))        # mcmc.flat_seed, seeded per point
        self.ndim, self.nwalkers, self.seed = 6, nwalkers, seed + g
        self.post_model = Model(post_desc, device=device)      # the chain is propagated with this one, on the device

    _templates = {}

    @classmethod
    def descriptors(cls, point, dimension, texture):
        """(paramset of the chain, its seed box, descriptor of the prior chain, descriptor of the point's flux average)"""
        scale, source = point
        tex = texture if isinstance(texture, Texture) else Texture(texture)
        key = (dimension, tex)
        if key not in cls._templates:                          # the same for every grid point: compiled once
            ps6 = Cf.ParamSet(list(Cf.texture_paramset(dimension))[:6])        # scale fixed per grid point
            cls._templates[key] = (ps6, np.array(ps6.seeds, dtype=float), compile_model(ps6, "PRIOR_ONLY", flat_llh=1.0),
                                   compile_model(ps6, "BSM_GAUSS", texture=tex, dimension=dimension, binning=_BIN_EDGES,
                                                 source_ratio=(1.0, 0.0, 0.0), scale_fixed=0.0, bestfit_fr=(1 / 3,) * 3,
                                                 smearing=0.02))
        ps6, box, prior_desc, post = cls._templates[key]
        return ps6, box, prior_desc, _patched(post, source_ratio=source, scale_fixed=scale)

    @staticmethod
    def assemble(samples, frs, status):
        frs = np.array(frs)
        frs[status != 0] = np.nan                                              # the reference would have raised there
        out = np.empty((samples.shape[0], 9))
        out[:, :3] = frs
        out[:, 3:] = samples
        return out

    def close(self):
        self.f.close()
        self.post_model.close()

    def collect(self, samples, frs=None, status=None):
        """samples: this point's flat chain (nwalkers*nsteps, 6); frs/status: its device post-processing"""
        if frs is None:
            frs, status = self.post_model.propagate(samples)
        out = self.assemble(samples, frs, status)
        self.close()
        return out


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
        ps, box, desc = self.descriptor(point, smearing)
        self.f = llh_utils.LnProb(desc, device=device, on_nonunitary="-inf")
        rng = np.random.default_rng(seed + g)
        self.p0 = rng.uniform(box[:, 0], box[:, 1], size=(nwalkers, 12))
        lo, hi = Cf.SCALE_BOUNDARIES[dim]
        self.p0[:, 11] = np.clip(rng.normal(scale, 0.5, nwalkers), lo, hi)
        self.ndim, self.nwalkers, self.seed = 12, nwalkers, seed + g

    post_model = None
    _paramsets = {}

    @classmethod
    def descriptor(cls, point, smearing=0.02):
        """(paramset, its seed box, descriptor) of the point: llh.bsm_ln_prob(args, asimov, ps, smearing)'s, the template
        compiled once per dimension"""
        dim, tex, source, scale = point
        key = (dim, smearing)
        if key not in cls._paramsets:                         # the same for every grid point of a dimension; read-only here
            asimov, ps = Cf.fr_paramsets(dim, fr_utils.fr_to_angles((1, 1, 1)))
            bf = fr_utils.angles_to_fr(asimov.from_tag(llh_utils.ParamTag.BESTFIT, values=True))
            template = compile_model(ps, "BSM_GAUSS", bestfit_fr=bf, smearing=smearing, source_ratio=(1.0, 0.0, 0.0),
                                     texture=Texture.OET, dimension=dim, binning=_BIN_EDGES)
            cls._paramsets[key] = (asimov, ps, np.array(ps.seeds, dtype=float), template)
        asimov, ps, box, template = cls._paramsets[key]
        return ps, box, _patched(template, source_ratio=source, texture=tex)

    @staticmethod
    def assemble(samples, frs, status):
        return samples                                       # the chain is the result: no copy

    def close(self):
        self.f.close()

    def collect(self, samples, frs=None, status=None):
        self.close()
        return samples


class ResultArena:
    """Host memory for MANY results: one block on 2 MiB pages, its pages mapped, registered with the HIP runtime (`gf_host_register`,
    ABI 5) -- the DMA engines then write a scan's chain or rows straight into it at the speed of the PCIe link (57 GB/s on the
    MI355X boxes) instead of through the pinned staging ring and the host's copy threads (28-47 GB/s into fresh pages, the rate
    that bounded both scans at the reference's chain length; profiles/r04/host_register.txt).  Setting it up costs what mapping its
    pages costs (~25 GB/s: 0.5 s for 12.6 GB), so it pays for a process that produces more than one result -- a job that scans a
    grid batch by batch and writes every batch out (scan.py's --datadir loop), bench.py -- not for a single read-back.

    The results handed out ALIAS the arena: the next scan overwrites the previous one's arrays (save or reduce them first).
    `set_result_arena(arena)` makes `result_array` -- every one-rank scan's destination -- use it; a request that does not fit gets
    fresh memory as before.  `close()` unregisters."""

    def __init__(self, nbytes, register=True):
        import time as _time
        from . import _lib
        t0 = _time.perf_counter()
        self.array = empty_hugepages((max(int(nbytes), 8) // 8,))
        self.registered, self.register_error = False, None
        if register:
            try:
                _lib.check(_lib.lib().gf_host_register(self.array.ctypes.data, self.array.nbytes), "gf_host_register")
                self.registered = True
            except Exception as exc:       # noqa: BLE001  (RLIMIT_MEMLOCK, no GPU ...): an unregistered arena still saves the page faults
                self.register_error = "%s: %s" % (type(exc).__name__, exc)
        if not self.registered:
            self.array[:: 4096 // 8] = 0.0                 # map the pages at least
        self.seconds = _time.perf_counter() - t0

    @property
    def nbytes(self):
        return 0 if self.array is None else int(self.array.nbytes)

    def take(self, shape):
        n = int(np.prod(shape, dtype=np.int64))
        if self.array is None or n > self.array.size:
            return None
        return self.array[:n].reshape(shape)

    def close(self):
        if self.registered and self.array is not None:
            try:
                from . import _lib
                _lib.lib().gf_host_unregister(self.array.ctypes.data)
            except Exception:              # noqa: BLE001
                pass
        self.registered, self.array = False, None


_ARENA = {"arena": None}


def set_result_arena(arena):
    """The arena `result_array` serves from (None: fresh memory per result, the default).  Returns the previous one."""
    old, _ARENA["arena"] = _ARENA["arena"], arena
    return old


def result_array(shape):
    """Host memory for a scan's result (GBs that a device-to-host copy is about to fill): the process's `ResultArena` if one is set
    and the result fits (registered memory: the copy is a DMA at link speed) -- GF_SCAN_ARENA=1 sets one up on first use, sized by
    that use --; else fresh memory backed by 2 MiB pages where the kernel grants them (model.empty_hugepages; GF_SCAN_NO_HUGEPAGES=1:
    plain np.empty, the A/B of profiles/r04/readback.jsonl)."""
    n = int(np.prod(shape, dtype=np.int64))
    arena = _ARENA["arena"]
    if arena is None and os.environ.get("GF_SCAN_ARENA"):
        arena = _ARENA["arena"] = ResultArena(n * 8)
    if arena is not None:
        a = arena.take(shape)
        if a is not None:
            return a
    return np.empty(shape) if os.environ.get("GF_SCAN_NO_HUGEPAGES") else empty_hugepages(shape)


PHASES = {}          # wall-clock seconds of the last run_points call, by phase (reported by main)
GATHER_STATS = {}     # the last DeviceGather.run: bytes and seconds by phase
LAST_NONUNITARY = {}  # the last stacked run_points call: proposals the reference would have raised on, and how they were settled


def run_points(points, indices, make, burnin, nsteps, stacked=True, seed=25, gather=None):
    """All of this rank's grid points advance together.  stacked (default): one sampler, one ensemble per
    grid point's posterior, every half-step of every chain in one launch; chain g draws from random stream g (its
    GLOBAL grid index), so that a grid point's chain does not depend on the number of ranks.  Otherwise one sampler per
    point, each on its own stream (`run_async`), so that the small launches overlap on the GPU.

    gather: None -> {grid index: collected array} of this rank's points (host);
            a `DeviceGather` -> every grid point's array, in grid order, on rank 0 (None elsewhere): the chain blocks
            go from the sampler's device buffer through RCCL and cross PCIe once."""
    t0 = time.perf_counter()
    jobs = {g: make(points[g], g) for g in indices}
    if not jobs:
        return {}
    order = list(jobs)
    first = jobs[order[0]]
    PHASES.clear()
    PHASES["setup"] = time.perf_counter() - t0
    if stacked:
        t0 = time.perf_counter()
        sampler = mcmc_utils.DeviceEnsembleSampler(first.nwalkers, first.ndim, [jobs[g].f for g in order], seed=seed,
                                                   stream_ids=order)
        sampler.on_nonunitary = "-inf"
        PHASES["sampling: sampler created"] = time.perf_counter() - t0
        # burn-in: enqueued, and only then -- while the GPU works through it and this thread has nothing to launch or allocate --
        # the result array is allocated and its pages are mapped (the size is known; model.Prefaulted).  Many threads taking page
        # faults hold up this thread's launches and allocations (they share the address space's lock), so the mapping must not
        # run beside them: it is waited for before the stored run starts (gather.destination)
        sampler.run_async(np.stack([jobs[g].p0 for g in order]), burnin, storechain=False)
        PHASES["sampling: burn-in enqueued"] = time.perf_counter() - t0
        if gather is not None:
            gather.prepare(first, len(order), len(points), nsteps)
        PHASES["sampling: destination prepared"] = time.perf_counter() - t0
        sampler.wait()
        sampler.reset()
        PHASES["sampling: burn-in done"] = time.perf_counter() - t0
        streamed = None
        if gather is not None and gather.streams_chain(first):
            # one rank, the chain is the result: its blocks of steps cross PCIe while the later ones are sampled
            streamed = sampler.run_mcmc_to_host(None, nsteps, out=gather.destination((len(order), nsteps, first.nwalkers, first.ndim)))
            gather.finish_destination()
            gather.readback_tail_s = sampler.readback_tail_s
        else:
            sampler.run_mcmc(None, nsteps)
        PHASES["sampling"] = time.perf_counter() - t0
        # proposals of the stored run the reference would have died on (fr.py:493-498): rejected on the device, counted here
        LAST_NONUNITARY.clear()
        LAST_NONUNITARY.update({"nonunitary_proposals_rejected": int(sampler.nonunitary_proposals),
                                "settled": "on the device, before the accept step (k_stretch_chain / k_stretch_settle)"})
        census = sampler.undecided_census()
        if census is not None:
            LAST_NONUNITARY.update(census)
        shape = sampler.launch_shape()
        if shape is not None:
            LAST_NONUNITARY["launch_shape"] = shape
        if streamed is not None:
            times = sampler.run_to_host_times()
            if times is not None:
                LAST_NONUNITARY["host_thread_times"] = times
        t0 = time.perf_counter()
        if gather is not None:
            out = gather.run(sampler, jobs, order, len(points), streamed=streamed)
            sampler.close()
            for j in jobs.values():
                j.close()
            PHASES["gather"] = time.perf_counter() - t0
            return out
        flat = sampler.flat_steps()                               # (npoints, nsteps*nwalkers, ndim), device order
        frs = sts = None
        if first.post_model is not None:                          # mc_texture.py:216-221 on the device
            post = sampler.postprocess(want_fr=True, want_status=True, step_major=True,
                                       models=[jobs[g].post_model for g in order])
            frs, sts = post["fr"].reshape(len(order), -1, 3), post["status"].reshape(len(order), -1)
        if len(order) == 1:
            flat = flat[None]
        sampler.close()
        PHASES["fetch"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = {g: jobs[g].collect(flat[i], None if frs is None else frs[i], None if sts is None else sts[i])
               for i, g in enumerate(order)}
        PHASES["collect"] = time.perf_counter() - t0
        return out
    samplers = {g: mcmc_utils.DeviceEnsembleSampler(j.nwalkers, j.ndim, j.f, seed=j.seed) for g, j in jobs.items()}
    for g, sm in samplers.items():
        sm.on_nonunitary = "-inf"
        sm.run_async(jobs[g].p0, burnin, storechain=False)
    for sm in samplers.values():
        sm.wait()
        sm.reset()
    for sm in samplers.values():
        sm.run_async(None, nsteps)
    out = {}
    for g, sm in samplers.items():
        sm.wait()
        flat = sm.flatchain
        sm.close()
        out[g] = jobs[g].collect(flat)
    return out


class DeviceGather:
    """The chain gather of a stacked scan on the device: every rank packs its chains (and, for mc_texture, their
    post-processed compositions) into one device block of `slots` grid points; RCCL gathers the blocks over xGMI onto
    RANK 0 ONLY (`rccl`: dist.RcclBackend.gather_device -- one group of point-to-point transfers into the root; None with one
    rank), and rank 0 downloads the result once, into prefaulted memory.  Only the root ever holds `world x` the block: the
    reference's equivalent is N jobs saving N files to one place (golemflavor/mcmc.py:108-126).

    `stats` (after `run`): block bytes per rank, gathered bytes, seconds of the pack / xGMI / device-to-host phases."""

    def __init__(self, rccl, rank, world, model_for_buffers, root=0, control=None):
        self.rccl, self.rank, self.world, self.m, self.root = rccl, rank, world, model_for_buffers, root
        self.control = control         # where an (agreed) hipIpc failure falls back to: blocks to the root over the control plane
        self.stats = {}
        self._dest = None

    def prepare(self, first_job, n_local, n_points, nsteps):
        """Hook for allocating the result array ahead of the run and mapping its pages in the background (model.Prefaulted,
        GF_SCAN_PREFAULT=1).  OFF by default: measured on the MI355X boxes with the pages REALLY mapped
        (madvise(MADV_POPULATE_WRITE); an earlier version's atomic OR with zero had been compiled into a load, which maps
        nothing) every variant lost -- beside the run (page-faulting threads hold up this thread's launches and allocations),
        and also confined to the burn-in and waited for (C5 100 + 200: 0.166-0.179 s against 0.122-0.126 s; at the reference's
        length 0.67-0.69 s against 0.38-0.52 s): the copy threads of the read-back map the pages they write as they go, on
        the cores that use them, and that is the fastest arrangement found (profiles/r03/readback.txt)."""
        if not os.environ.get("GF_SCAN_PREFAULT"):
            return
        per = nsteps * first_job.nwalkers
        width = first_job.ndim if first_job.post_model is None else 3 + first_job.ndim
        if self.streams_chain(first_job):
            shape = (n_local, nsteps, first_job.nwalkers, first_job.ndim)
        elif self.rccl is not None and self.rank == self.root:
            shape = (self.world, gdist.slots_per_rank(n_points, self.world), per, width)
        else:
            shape = (n_local, per, width)
        self._dest = Prefaulted(shape)

    def destination(self, shape):
        """Where the result goes: the prepared array, its pages mapped (GF_SCAN_PREFAULT; waits for the mapping threads), if it has
        this shape, else fresh memory on huge pages (`result_array`)."""
        d, self._dest = self._dest, None
        if d is not None:
            a = d.get()
            if a.shape == tuple(shape):
                return a
        return result_array(tuple(shape))

    def finish_destination(self):
        pass

    def _exchange(self, d_send, nbytes, shape, dtype):
        if self.rccl is None:
            t0 = time.perf_counter()
            out = d_send.download(shape[1:], dtype=dtype, out=self.destination(shape[1:]))[None] if self.rank == self.root else None
            self.finish_destination()
            self.stats.update(xgmi_s=0.0, d2h_s=time.perf_counter() - t0, gather_bytes=0, d2h_bytes=int(nbytes) if self.rank == self.root else 0)
            return out
        is_root = self.rank == self.root
        d_recv = self.m.alloc(nbytes * self.world) if is_root else None          # world x block on the root only
        t0 = time.perf_counter()
        try:
            self.rccl.gather_device(d_send.ptr, d_recv.ptr if is_root else None, nbytes, self.root)
        except _lib.GolemHipError as exc:
            # a hipIpc failure is AGREED: the root's report has reached every rank (dist.IpcBackend.gather_device), so all of
            # them land here together and the blocks can still go to the root through the host (an RCCL error is local: re-raise)
            if getattr(self.rccl, "kind", "") != "hipIpc" or self.control is None or exc.code != _lib.GF_ERR_COMM:
                raise
            if is_root:
                d_recv.free()
            mine = d_send.download(shape[1:], dtype=dtype)
            t1 = time.perf_counter()
            npseudo = self.world * shape[1]                 # every slot of every rank as a point of its own
            pts = {g: mine[b] for b, g in enumerate(gdist.shard(npseudo, self.rank, self.world))}
            parts = gdist.gather_chains_to_root(pts, npseudo, self.control, self.root)
            self.stats.update(xgmi_s=0.0, d2h_s=t1 - t0, host_gather_s=time.perf_counter() - t1, gather_bytes=0,
                              d2h_bytes=int(nbytes), ipc_error=str(exc))
            if not is_root:
                return None
            out = np.empty(shape, dtype=dtype)
            for g, part in enumerate(parts):
                out[gdist.owner(g, self.world), g // self.world] = part
            return out
        t1 = time.perf_counter()
        out = d_recv.download(shape, dtype=dtype, out=self.destination(shape)) if is_root else None
        self.finish_destination()
        t2 = time.perf_counter()
        if is_root:
            d_recv.free()
        # bytes that crossed xGMI into the root -- every block but its own -- counted ONCE, on the root: bench.reduce_phases SUMS
        # the `*_bytes` keys over the ranks (round 3 counted them on every rank and the line's xGMI rate read `world` times too high)
        self.stats.update(xgmi_s=t1 - t0, d2h_s=t2 - t1, gather_bytes=int(nbytes) * (self.world - 1) if is_root else 0,
                          d2h_bytes=int(nbytes) * self.world if is_root else 0)
        return out

    def streams_chain(self, first_job):
        """True where nothing has to be exchanged or post-processed: one rank and the chain itself is the result (C5)."""
        if os.environ.get("GF_SCAN_NO_STREAMED_CHAIN"):           # A/B: sample, then read back
            return False
        return self.rccl is None and self.world == 1 and first_job.post_model is None

    def run(self, sampler, jobs, order, n_points, streamed=None):
        if streamed is not None:                                  # (nchains, nstored, nwalkers, ndim), read back during the run
            rows = streamed.reshape(streamed.shape[0], -1, streamed.shape[-1])
            self.stats = GATHER_STATS
            self.stats.clear()
            self.stats.update({"ranks": 1, "slots_per_rank": n_points, "pack_s": 0.0, "xgmi_s": 0.0,
                               "d2h_s": float(getattr(self, "readback_tail_s", 0.0)), "gather_bytes": 0, "block_bytes": int(rows.nbytes),
                               "note": "the chain was read back while it was sampled (one rank): d2h_s is what was left after the run's last step"})
            return [rows[order.index(g)] for g in range(n_points)] if list(order) != list(range(n_points)) else list(rows)
        slots = gdist.slots_per_rank(n_points, self.world)
        first = jobs[order[0]]
        per = sampler.nstored * first.nwalkers                    # samples per grid point
        width = first.ndim if first.post_model is None else 3 + first.ndim
        self.stats = GATHER_STATS
        self.stats.clear()
        self.stats.update({"ranks": self.world, "slots_per_rank": slots})
        if self.rccl is None and self.world == 1 and first.post_model is not None:
            # one rank: no exchange -- rows to the host group by group while the later chains are still post-processed
            t0 = time.perf_counter()
            rows = sampler.postprocess_rows(models=[jobs[g].post_model for g in order],
                                            out=self.destination((len(order), per, width)))
            self.finish_destination()
            self.stats.update(pack_s=0.0, xgmi_s=0.0, d2h_s=time.perf_counter() - t0, gather_bytes=0,
                              block_bytes=int(rows.nbytes), note="post-processing and read-back overlap (one rank)")
            return [rows[order.index(g)] for g in range(n_points)] if list(order) != list(range(n_points)) else list(rows)
        blk = slots * per * width * 8
        t0 = time.perf_counter()
        d_rows = self.m.alloc(blk)                                # ranks with fewer points leave the tail unused
        if first.post_model is None:
            sampler.chain_to_device(d_rows.ptr)
        else:                                                     # mc_texture.py:216-223 on the device: (fr, sample) rows
            sampler.postprocess_rows_to_device(d_rows.ptr, models=[jobs[g].post_model for g in order])
        self.stats.update(pack_s=time.perf_counter() - t0, block_bytes=int(blk))
        rows = self._exchange(d_rows, blk, (self.world, slots, per, width), np.float64)
        d_rows.free()
        if self.rank != self.root:
            return None
        return [rows[gdist.owner(g, self.world), g // self.world] for g in range(n_points)]


class SharedHostGather:
    """Delivery of a multi-rank scan over N PCIe links: every rank reads its OWN chains back from its GPU into its part of one
    host segment all ranks of the node map (`dist.HostSegment`); rank 0 has the whole grid without a byte crossing xGMI, a
    socket or somebody else's link.  This is the reference's arrangement -- N jobs, each writing its own chain to one place
    (golemflavor/mcmc.py:108-126, submitter/mc_texture_dag.py:57-71, submitter/sens_dag.py:75-95) -- and the only one in
    which eight GPUs shorten a scan whose cost is its read-back (c4_scan_ref / c5_scan_ref at the reference's chain length
    are 9.4 / 12.6 GB; gathered onto GPU 0 they would all leave through GPU 0's link).

    Every rank runs exactly the ONE-rank code path into its region: C5's chain crosses PCIe while it is sampled
    (`gf_sampler_run_to_host`), C4's rows are post-processed and read back group by group (`gf_sampler_postprocess_rows`) -- so
    pack, exchange and download are not three phases but one pipeline per rank, and there is no exchange.

    Layout of the segment: [world][slots][samples per point][width] -- rank r's points in its own contiguous block, grid point g
    at [owner(g), g div world] (the layout `DeviceGather` produces on rank 0).  If the segment cannot be set up (agreed on by
    all ranks inside `HostSegment`), every rank reads back into private memory and the blocks go to rank 0 over the control
    plane (`dist.gather_chains_to_root`); `stats["delivery"]` says which."""

    kind = "shared host segment"

    def __init__(self, control, rank, world, root=0, arena=None):
        """arena: a `dist.HostSegment` that outlives this scan (every rank passes its own handle of the same segment, or every
        rank None).  The pages of a shared segment are 4 KiB shmem pages that somebody has to allocate and zero -- 17 GB/s at
        best on the MI355X boxes (profiles/r04/shm_pages.txt), less than ONE PCIe link delivers -- so a job that runs more than one
        scan, or has anything to do before its scan, creates the segment once, ahead of time (bench.py: beside the RCCL
        bootstrap), and every scan whose result fits uses it; the result of an earlier scan is overwritten by the next."""
        self.control, self.rank, self.world, self.root = control, int(rank), int(world), int(root)
        self.rccl = None
        self.stats = {}
        self.arena = arena
        self.seg, self.region, self._all = None, None, None
        self.readback_tail_s = 0.0

    def streams_chain(self, first_job):
        return first_job.post_model is None and not os.environ.get("GF_SCAN_NO_STREAMED_CHAIN")

    def prepare(self, first_job, n_local, n_points, nsteps):
        """Collective (every rank calls it, once the burn-in is enqueued: the size is known and the host has nothing else to do):
        creates / maps the segment.  Pages are mapped by the copy threads as they fill them."""
        per = nsteps * first_job.nwalkers
        width = first_job.ndim if first_job.post_model is None else 3 + first_job.ndim
        slots = gdist.slots_per_rank(n_points, self.world)
        self._geom = (slots, per, width, n_local, n_points)
        t0 = time.perf_counter()
        need = self.world * slots * per * width * 8
        if self.arena is not None and self.arena.error is None and self.arena.nbytes >= need:
            self.seg = self.arena                  # (sizes are a function of the grid: every rank takes this branch or none does)
        else:
            self.seg = gdist.HostSegment(self.control, need, root=self.root)
        if self.seg.error is None:
            self._all = self.seg.array((self.world, slots, per, width))
            self.region = self._all[self.rank]
        else:
            self.region = np.empty((slots, per, width))
        self.segment_s = time.perf_counter() - t0

    def destination(self, shape):
        """This rank's region, shaped as the callee wants it: (n_local, nsteps, nwalkers, ndim) for the streamed chain,
        (n_local, per, width) for post-processed rows.  Collective the first time: the read-back starts once the segment's pages
        exist (rank 0's background posix_fallocate: a copy into pages that are still being allocated takes a fault per 4 KiB AND
        fights the allocation for the file's locks -- 1.2 s instead of 0.25 s for C4's 9.4 GB on two ranks)."""
        if not getattr(self, "_allocated", False):
            self._allocated = True
            if self._all is not None:
                t0 = time.perf_counter()
                self.seg.wait_allocated()
                self.control.barrier()
                self.allocation_wait_s = time.perf_counter() - t0
        slots, per, width, n_local, _ = self._geom
        assert int(np.prod(shape)) == n_local * per * width, (shape, self._geom)
        return self.region[:n_local].reshape(shape)

    def finish_destination(self):
        pass

    def run(self, sampler, jobs, order, n_points, streamed=None):
        slots, per, width, n_local, _ = self._geom
        first = jobs[order[0]]
        t0 = time.perf_counter()
        note = None
        if streamed is not None:
            d2h_s = float(self.readback_tail_s)
            note = "the chain was read back while it was sampled: d2h_s is what was left after the run's last step"
        elif first.post_model is not None:
            sampler.postprocess_rows(models=[jobs[g].post_model for g in order], out=self.destination((n_local, per, width)))
            d2h_s = time.perf_counter() - t0
            note = "post-processing and read-back overlap, group of chains by group of chains"
        else:
            sampler.chain_to_host(self.destination((n_local, sampler.nstored, first.nwalkers, first.ndim)))
            d2h_s = time.perf_counter() - t0
        t1 = time.perf_counter()
        self.stats = GATHER_STATS
        self.stats.clear()
        nbytes = n_local * per * width * 8
        self.stats.update({"ranks": self.world, "slots_per_rank": slots, "pack_s": 0.0, "xgmi_s": 0.0, "gather_bytes": 0,
                           "d2h_s": d2h_s, "d2h_bytes": int(nbytes), "block_bytes": int(nbytes), "segment_s": self.segment_s,
                           "segment_allocation_wait_s": float(getattr(self, "allocation_wait_s", 0.0)),
                           "delivery": ("shared host segment (%s%s): every rank reads its own chains back over its own PCIe link"
                                        % (self.seg.kind, ", created ahead of the scan" if self.seg is self.arena else "")) if self._all is not None else
                                       "private read-back, blocks to rank 0 over the control plane (no shared segment: %s)" % self.seg.error})
        if note:
            self.stats["note"] = note
        if self._all is None:
            # no segment: the host-side stand-in for the gather
            local = {g: self.region[i] for i, g in enumerate(order)}
            out = gdist.gather_chains_to_root(local, n_points, self.control, self.root)
            self.stats["host_gather_s"] = time.perf_counter() - t1
            return out
        self.control.barrier()                     # every rank's part is in place
        self.stats["wait_for_peers_s"] = time.perf_counter() - t1
        if self.rank != self.root:
            return None
        return [self._all[gdist.owner(g, self.world), g // self.world] for g in range(n_points)]

    def release(self):
        """Drop this object's views of the segment and unmap it (the arrays `run` handed out on rank 0 keep the pages for as long
        as they live)."""
        self._all = self.region = None
        if self.seg is not None and self.seg is not self.arena:
            self.seg.close()
        self.seg = None


def segment_bytes(n_points, world, nwalkers, nsteps, width):
    """Bytes of the host segment a `SharedHostGather` needs for a grid of n_points on `world` ranks."""
    return world * gdist.slots_per_rank(n_points, world) * nsteps * nwalkers * width * 8


def finite_fraction(arrays, stride_target=4096):
    """Fraction of finite entries, on a strided sample of each array (the check of a 10 GB result must not cost a pass over it)."""
    vals = [np.isfinite(c[:: max(1, len(c) // stride_target)]).mean() for c in arrays if len(c)]
    return float(np.mean(vals)) if vals else float("nan")


def point_filename(config, point, a):
    """File stem of one grid point's chain: the reference's `mc_texture` + gen_identifier (scripts/mc_texture.py:184,
    misc.py:44-51: `_DIM{d}_sfr_{source}[_{texture}]`), plus the scale, which the reference's identifier lacks
    because its jobs do not scan it."""
    if config == "C4":
        scale, source = point
        ns = argparse.Namespace(dimension=a.dimension, source_ratio=source, texture=Texture[a.texture])
        return "mc_texture%s_logLam%+.3f" % (mcmc_utils.chain_identifier(ns), scale)
    dim, tex, source, scale = point
    ns = argparse.Namespace(dimension=dim, source_ratio=source, texture=tex)
    return "fr%s_logLam%+.3f" % (mcmc_utils.chain_identifier(ns), scale)


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
    ap.add_argument("--no-stack", action="store_true", help="one sampler per grid point on its own stream (A/B)")
    ap.add_argument("--datadir", default=None,
                    help="write one .npy per grid point, named as the reference's jobs name theirs "
                         "(scripts/mc_texture.py:184, misc.py:44-51), each rank its own points")
    a = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = int(os.environ.get("GF_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    control = gdist.SocketBackend(rank, world) if world > 1 else gdist.LocalBackend()

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
    mine = gdist.shard(len(pts), rank, world)
    stacked = not a.no_stack
    can_deliver = stacked and len(pts) >= world       # every rank has a point: the gather objects' collectives line up
    # How the chains reach their destination, in order of preference:
    #   --datadir         every rank writes its own points' files, as the reference's jobs do: no gather at all
    #   one node          every rank reads its chains back over its own PCIe link into one shared host segment (SharedHostGather)
    #   several nodes     device gather to rank 0 over RCCL / xGMI (or hipIpc), one download (DeviceGather; GF_SCAN_RCCL forces it)
    #   no communicator   the blocks go to rank 0 through the host control plane
    want_gather = (not a.datadir) or bool(a.outfile)
    force_device = bool(os.environ.get("GF_SCAN_RCCL")) or bool(os.environ.get("GF_SCAN_DEVICE_GATHER"))
    shared = False
    arena = None
    if world > 1 and want_gather and can_deliver and not force_device:
        shared = gdist.same_node(control)
        if shared:
            # created FIRST: rank 0 allocates its pages in the background (17 GB/s at best for 4 KiB shmem pages) while every rank
            # compiles its models, burns in and samples
            width = 9 if a.config == "C4" else 12
            arena = gdist.HostSegment(control, segment_bytes(len(pts), world, nw, a.nsteps, width))
    rccl, rccl_err, stuck = None, None, False
    if want_gather and not shared and (world > 1 or os.environ.get("GF_SCAN_RCCL")):
        rccl, rccl_err, stuck = gdist.open_device_gather(rank, world, device, control, timeout=float(os.environ.get("GF_RCCL_TIMEOUT", "60")))
    device_gather = want_gather and can_deliver and not shared and (rccl is not None or world == 1)
    gather_name, chains, local = "local", None, None
    if a.datadir and not want_gather:
        local = run_points(pts, mine, make, a.burnin, a.nsteps, stacked=stacked)
        gather_name = "none: every rank saved its own files (--datadir)"
    elif shared:
        g = SharedHostGather(control, rank, world, arena=arena)
        chains = run_points(pts, mine, make, a.burnin, a.nsteps, stacked=True, gather=g)
        gather_name = g.stats.get("delivery", g.kind)
        g.release()
    elif device_gather:
        stage = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=device)
        chains = run_points(pts, mine, make, a.burnin, a.nsteps, stacked=True,
                            gather=DeviceGather(rccl, rank, world, stage, control=control))
        stage.close()
        gather_name = ("%s device gather to rank 0" % rccl.kind) if rccl is not None else "device -> host"
    else:
        local = run_points(pts, mine, make, a.burnin, a.nsteps, stacked=stacked)
        t1 = time.perf_counter()
        chains = gdist.gather_chains_to_root(local, len(pts), control)
        PHASES["gather"] = time.perf_counter() - t1
        gather_name = "socket control plane, blocks to rank 0" if world > 1 else "local"
    if a.datadir:
        # the reference's jobs each save their own chain to the shared filesystem; so does every rank here (where the chains
        # were also gathered for --outfile, rank 0 holds them all and writes the files)
        todo = local if local is not None else ({g: chains[g] for g in range(len(pts))} if rank == 0 else {})
        for g, arr in todo.items():
            mcmc_utils.save_chains(arr, os.path.join(a.datadir, point_filename(a.config, pts[g], a)))
    if rccl is not None:
        rccl.close()
    dt = time.perf_counter() - t0
    # what the line says about the result: from the gathered chains on rank 0, or -- nothing gathered -- from every rank's own
    # points through a reduction of two numbers per rank
    if chains is None and local is not None and a.datadir and not want_gather:
        vals = [(finite_fraction([local[g]]), local[g].shape) for g in mine]
        fsum = np.array([sum(v for v, _ in vals), float(len(vals))])
        tot = control.allgather(fsum).sum(axis=0) if world > 1 else fsum
        finite = float(tot[0] / max(tot[1], 1.0))
        shape0 = list(vals[0][1]) if vals else []
        shapes = control.allgather_bytes(json.dumps(shape0).encode()) if world > 1 else [json.dumps(shape0).encode()]
        shape0 = next((json.loads(x.decode()) for x in shapes if json.loads(x.decode())), [])
        chains_shape = [len(pts)] + shape0
    elif rank == 0:
        finite = finite_fraction(chains)
        chains_shape = [len(chains)] + list(chains[0].shape)
    if rank == 0:
        if a.outfile:
            mcmc_utils.save_chains(np.stack(chains), a.outfile)
        print(json.dumps({"config": a.config, "grid_points": len(pts), "ranks": world, "walkers": nw, "burnin": a.burnin,
                          "nsteps": a.nsteps, "stacked": stacked, "gather": gather_name,
                          "rccl_error": rccl_err, "librccl": gdist.rccl_library_info(),
                          "diagnostic_overrides": _lib.diagnostic_overrides(),
                          "gather_stats": GATHER_STATS, "nonunitary": LAST_NONUNITARY,
                          "chains_shape": chains_shape, "seconds": dt,
                          "phases": {k: round(v, 4) for k, v in PHASES.items()},
                          "evals_per_s": len(pts) * evals_per_point / dt,
                          "finite_fraction": finite}), flush=True)
    control.barrier()
    control.close()
    if rccl_err is not None:
        # the run is complete (host gather) but the RCCL path is broken: say so with the exit status too
        import sys
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3) if stuck else sys.exit(3)


if __name__ == "__main__":
    main()
