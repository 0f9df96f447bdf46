#!/usr/bin/env python3
"""bench.py -- walker-lnprob evaluations per second (BASELINE.json metric).

Workload (config.workload): BASELINE config 2 -- the 6-dim Gaussian-likelihood posterior of
examples/inference.ipynb, 4096-walker ensembles, fp64 -- with `--ensembles` independent ensembles
stacked into ONE launch (default 4096 -> 16.8M walkers / launch), theta resident in HBM when the
timed region starts.  A "step" is one pass of the hot path (one kernel launch) over that batch.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Ensembles (independent chains) shard across ranks with no data-path
collective (weak scaling: every rank evaluates `--ensembles` ensembles).  RCCL (through the
library's own gf_comm_* C ABI) broadcasts the packed model descriptor from rank 0 before the
timed region and all-gathers one chain block per rank after it; host-side control (barrier,
max-over-ranks) goes through torch.distributed/gloo.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant (only) kernel: algorithmic bytes
(56 B per evaluation: 6 x 8 B read + 8 B written) over the average launch duration measured with
HIP events on the stream the kernel runs on.  `cpu_baseline` is the CPU oracle (oracle/, a
long-double restatement of the reference, kind "port") timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from golemflavor_amd import _lib  # noqa: E402
from golemflavor_amd import configs as Cf  # noqa: E402
from golemflavor_amd import fr as fr_utils  # noqa: E402
from golemflavor_amd.descriptor import compile_model  # noqa: E402
from golemflavor_amd.model import Model  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_EVAL = 6 * 8 + 8     # 8*ndim read + 8 written, no fr / status blob


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--walkers", type=int, default=4096, help="walkers per ensemble (BASELINE config 2)")
    ap.add_argument("--ensembles", type=int, default=4096, help="independent ensembles stacked per launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="skip the emcee-driven extra (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="evaluations in the CPU-baseline sample (0 = auto)")
    return ap.parse_args()


def notebook_descriptor():
    ang = fr_utils.fr_to_angles(fr_utils.u_to_fr((1, 0, 0), fr_utils.NUFIT_U))
    asimov, ps = Cf.notebook_paramsets(ang)
    bf = fr_utils.angles_to_fr(asimov.values)
    return ps, bf, compile_model(ps, "SM_GAUSS", bestfit_fr=bf, smearing=0.02)


def synth_theta(ps, n, seed):
    """theta ~ U(seed box), the distribution mcmc.flat_seed draws p0 from (mcmc.py:88-96)."""
    rng = np.random.default_rng(seed)
    box = np.array(ps.seeds, dtype=np.float64)
    out = np.empty((n, len(ps)), dtype=np.float64)
    chunk = 1 << 20
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        out[i:i + m] = rng.uniform(box[:, 0], box[:, 1], size=(m, len(ps)))
    return out


def cpu_baseline(ps, bf, theta, sample):
    """Time the CPU oracle (checker, never the product) on a bounded sample of the same workload."""
    from oracle import oracle as O
    om = O.make_model(ps, "SM_GAUSS", bestfit_fr=bf, smearing=0.02)
    cores = min(os.cpu_count() or 1, 16)   # the GPU box grants a 16-core share per GPU
    th1 = theta[:min(len(theta), 200000)]
    t0 = time.perf_counter()
    O.lnprob_batch(om, th1)
    t1 = time.perf_counter() - t0
    rate1 = len(th1) / t1
    n = sample or int(min(len(theta), max(200000, rate1 * cores * 8.0)))   # ~8 s of all-core work
    ths = theta[:n]
    t0 = time.perf_counter()
    ref = O.lnprob_batch(om, ths, threads=cores)
    tm = time.perf_counter() - t0
    # the Python reference itself never travels to the GPU box; its rate was measured in the build
    # container when the golden vectors were generated (tests/golden/golden_meta.json, 1 core)
    ref_rate = None
    try:
        meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_meta.json")))
        ref_rate = 1e6 / meta["timings"]["notebook_ln_prob_us"]
    except Exception:
        pass
    return {"value": n / tm, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d evaluations of the bench theta batch, oracle/golem_oracle.c (long double), %d threads; "
                      "single-thread rate %.3g evals/s on %d evaluations" % (n, cores, rate1, len(th1)),
            "single_thread_value": rate1,
            "reference_python_evals_per_s_1core_build_container": ref_rate}, ref


def _claim_stdout():
    """Keep fd 1 clean for the ONE JSON line: gloo and RCCL print banners to the C-level stdout, so
    everything else in this process is pointed at stderr; returns the stream the JSON line goes to."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


def main():
    a = parse()
    json_out = _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "GF_BENCH_DEVICE" in os.environ:      # rehearsal on a 1-GPU box: every rank on the same device
        local_rank = int(os.environ["GF_BENCH_DEVICE"])
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dist = None
    comm = None
    rccl_error = None
    hard_exit = False          # a stuck RCCL bootstrap thread would block interpreter shutdown
    L = _lib.lib()
    ps, bf, desc = notebook_descriptor()
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        # RCCL communicator of the library itself: unique id travels over the gloo store.  The timed
        # region has no collective, so an RCCL problem must not cost the measurement: fall back to gloo
        # for the descriptor broadcast and report the failure in the JSON line.
        rccl_error = None
        box = {}

        def _rccl_setup():
            try:
                h = C.c_void_p()
                _lib.check(L.gf_comm_create(idb, rank, world, local_rank, C.byref(h)), "gf_comm_create")
                box["comm"] = h
                # fixed physics constants: rank 0's packed descriptor is the one everybody uses
                raw = (C.c_uint8 * C.sizeof(desc)).from_buffer(desc)
                _lib.check(L.gf_comm_broadcast(h, raw, C.sizeof(desc), 0), "gf_comm_broadcast")
                box["ok"] = True
            except Exception as exc:       # noqa: BLE001
                box["err"] = "%s: %s" % (type(exc).__name__, exc)

        try:
            ids = [None]
            if rank == 0:
                buf = (C.c_uint8 * _lib.GF_COMM_ID_BYTES)()
                _lib.check(L.gf_comm_unique_id(buf), "gf_comm_unique_id")
                ids = [bytes(buf)]
            dist.broadcast_object_list(ids, src=0)
            idb = (C.c_uint8 * _lib.GF_COMM_ID_BYTES).from_buffer_copy(ids[0])
            # a communicator that cannot bootstrap must not hang the measurement: bounded wait
            import threading
            th = threading.Thread(target=_rccl_setup, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("GF_RCCL_TIMEOUT", "60")))
            if th.is_alive():
                rccl_error = "timeout: RCCL communicator setup did not finish"
                hard_exit = True
            elif "err" in box:
                rccl_error = box["err"]
            else:
                comm = box["comm"]
        except Exception as exc:           # noqa: BLE001
            rccl_error = "%s: %s" % (type(exc).__name__, exc)
            comm = None
        flags = [rccl_error]
        gathered = [None] * world
        dist.all_gather_object(gathered, rccl_error)
        if any(g is not None for g in gathered):
            rccl_error = next(g for g in gathered if g is not None)
            if comm is not None:
                L.gf_comm_destroy(comm)
                comm = None
            blob = [bytes(memoryview(desc))] if rank == 0 else [None]
            dist.broadcast_object_list(blob, src=0)
            desc = _lib.GfModelDesc.from_buffer_copy(blob[0])

    n = a.walkers * a.ensembles
    model = Model(desc, device=local_rank)
    theta = synth_theta(ps, n, seed=26 + rank)
    d_theta = model.alloc(theta.nbytes).upload(theta)
    d_out = model.alloc(8 * n)

    def step():
        model.lnprob_device(d_theta.ptr, n, d_out.ptr, None, None)

    def fence():
        model.sync()
        if dist is not None:
            dist.barrier()
        model.sync()

    for _ in range(a.warmup):
        step()
    ev0, ev1 = model.event(), model.event()
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    model.sync()                                       # this rank's K steps are complete ...
    elapsed = time.perf_counter() - t0                 # ... at this instant; the MAX over ranks below is the job's time
    fence()                                            # closing bracket: barrier + synchronize (its latency is not a step)
    kernel_ms = ev0.elapsed_ms(ev1) / a.steps          # average launch duration on the kernel's stream

    if dist is not None:
        import torch
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # after the timed region: gather one chain block (the first ensemble's lnprob) from every rank
    gathered_ok = None
    if comm is not None:
        try:
            blk = 8 * a.walkers
            d_all = model.alloc(blk * world)
            _lib.check(L.gf_comm_allgather(comm, d_out.ptr, d_all.ptr, blk), "gf_comm_allgather")
            allv = d_all.download((world, a.walkers))
            mine = d_out.download((a.walkers,))
            gathered_ok = bool(np.array_equal(allv[rank], mine, equal_nan=True))
        except Exception as exc:           # noqa: BLE001
            rccl_error = "%s: %s" % (type(exc).__name__, exc)
            gathered_ok = False

    if rank == 0:
        evals = float(n) * a.steps * world
        value = evals / elapsed
        ach = BYTES_PER_EVAL * n / (kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC pass committed under profiles/ (same command, same n)
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if int(tj.get("n", -1)) == n:
                    traffic = tj.get("traffic_bytes_per_launch")
            except Exception:
                traffic = None
        metric = "walker-lnprob evals/sec (Gaussian llh, 100 walkers) at 1/2/4/8 MI355X"
        try:                                   # BASELINE.json's own wording when the file travelled with the repo
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            pass
        out = {
            "metric": metric, "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: examples/inference.ipynb 6-dim Gaussian-llh posterior, %d-walker ensembles, "
                                   "%d independent ensembles stacked per launch per GPU (theta resident in HBM)"
                                   % (a.walkers, a.ensembles),
                       "walkers_per_ensemble": a.walkers, "ensembles_per_launch_per_gpu": a.ensembles,
                       "evals_per_step_per_gpu": n, "ndim": 6, "parallelism": "independent ensembles sharded over %d GPU(s)" % world,
                       "note": "BASELINE.json words the metric on its configs[0] (100-walker chain, CPU plumbing); the bench line "
                               "is configs[1], 4096-walker ensembles on the GPU; the 100-walker chain itself is in emcee_driven_c1"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": kernel_ms, "bytes_per_eval": BYTES_PER_EVAL,
                         "kernel": "k_lnprob_sm_fast<6, SM_GAUSS, canonical, no fr>"},
        }
        if gathered_ok is not None:
            out["rccl_gather_ok"] = gathered_ok
        if rccl_error is not None:
            out["rccl_error"] = rccl_error
        if world == 1 and not a.no_sampler:
            # emcee-driven figure (not `value`): one 4096-walker ensemble advanced by the device-resident
            # stretch-move sampler, 2 launches per step, walkers never leave HBM
            try:
                from golemflavor_amd import mcmc as mcmc_utils
                rngp = np.random.default_rng(26)
                box = np.array(ps.seeds, dtype=np.float64)
                p0 = rngp.uniform(box[:, 0], box[:, 1], size=(a.walkers, 6))
                smp = mcmc_utils.DeviceEnsembleSampler(a.walkers, 6, model, seed=26)
                smp.run_mcmc(p0, 50, storechain=False)
                t0 = time.perf_counter()
                smp.run_mcmc(None, 500, storechain=False)
                dt = time.perf_counter() - t0
                out["emcee_driven"] = {"sampler": "device-resident stretch move", "walkers": a.walkers, "chains": 1,
                                       "steps": 500, "us_per_step": 1e6 * dt / 500, "evals_per_s": a.walkers * 500 / dt,
                                       "acceptance_fraction": float(np.mean(smp.acceptance_fraction))}
                smp.close()
                # configs[0] as written: ONE 100-walker chain (one workgroup, walkers in LDS, the run is one launch),
                # and 256 such chains side by side
                for key, nch, steps in (("emcee_driven_c1", 1, 20000), ("emcee_driven_c1_x256", 256, 4000)):
                    p1 = rngp.uniform(box[:, 0], box[:, 1], size=(nch, 100, 6))
                    smp = mcmc_utils.DeviceEnsembleSampler(100, 6, model, nchains=nch, seed=26)
                    smp.run_mcmc(p1 if nch > 1 else p1[0], 100, storechain=False)
                    t0 = time.perf_counter()
                    smp.run_mcmc(None, steps, storechain=False)
                    dt = time.perf_counter() - t0
                    out[key] = {"sampler": "device-resident stretch move, one workgroup per ensemble", "walkers": 100,
                                "chains": nch, "steps": steps, "us_per_step": 1e6 * dt / steps,
                                "evals_per_s": 100.0 * nch * steps / dt,
                                "acceptance_fraction": float(np.mean(smp.acceptance_fraction))}
                    smp.close()
            except Exception as exc:       # noqa: BLE001
                out["emcee_driven"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if world == 1 and not a.no_cpu_baseline:
            cb, ref = cpu_baseline(ps, bf, theta, a.cpu_sample)
            got = d_out.download((len(ref),))
            with np.errstate(all="ignore"):
                out["parity_max_rel_vs_oracle"] = float(np.max(np.abs(got - ref) / np.abs(ref)))
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]                       # vs the oracle port on this box's cores
            refrate = cb.get("reference_python_evals_per_s_1core_build_container")
            if refrate:                                                     # vs the reference itself (timed where it can run)
                out["gpu_over_reference_python_1core"] = value / refrate
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    if comm is not None:
        L.gf_comm_destroy(comm)
    model.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if hard_exit:
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
