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
timed region and all-gathers one chain block per rank after it; host-side control (rendezvous, barrier,
max-over-ranks, the RCCL id) runs over plain TCP sockets (golemflavor_amd.dist.SocketBackend) -- there is no
PyTorch in this process, so the ROCm runtime and librccl it maps are always /opt/rocm's (the path is in the line).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant (only) kernel of the timed region: algorithmic bytes
(56 B per evaluation: 6 x 8 B read + 8 B written) over the average launch duration measured with
HIP events on the stream the kernel runs on.  `cpu_baseline` is the CPU oracle (oracle/, a
long-double restatement of the reference, kind "port") timed on this host on a bounded sample.

After the timed region (`--no-extras` skips them) the other BASELINE configurations are measured too, each bounded to a
few seconds, as sub-records of the same line.  At every N: `c4_scan_ref` / `c5_scan_ref` -- the full 64 x 2048 and
256 x 512 grid scans SHARDED over the N ranks (grid point g -> rank (g + g div N) mod N, one stacked device sampler per rank) at the
reference's own chain length (burnin 200 + 1000 stored steps, submitter/mc_texture_dag.py:33-39), chains gathered to rank 0
over RCCL / xGMI (`gf_comm_gather`) and downloaded once, reported by phase (`rccl_init_s`, `sampling_s`, `pack_s`,
`gather_bytes`, `gather_GBps`, `d2h_s`, `ranks`, `evals_per_s`; a fixed grid: these sub-records are STRONG scaling) -- and
`cpu_baseline` (rank 0).  At N = 1 also: `c3` (1e7 Haar draws), `c4_bulk` / `c5_bulk` (the 7- / 12-column flux-averaged
posterior kernel with and without the unitarity status), `c4_scan` / `c5_scan` (the same scans at 100 + 200 steps, the
round-1/2 series), `emcee_driven` (one 4096-walker ensemble) and `emcee_driven_c1_scaling` (100-walker chains, the
metric's own wording, x {1, 16, 256, 4096}).

The line is put together by `assemble_line` / `scan_record_from_phases` from plain numbers, and the cross-rank reductions go
through `reduce_step_timing` / `reduce_phases` on the control plane: tests/test_bench_line.py drives them with two ranks'
synthetic measurements over the socket control plane on a box without a GPU.
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
from golemflavor_amd import dist as gdist  # noqa: E402
from golemflavor_amd import fr as fr_utils  # noqa: E402
from golemflavor_amd.descriptor import compile_model  # noqa: E402
from golemflavor_amd.enums import Texture  # noqa: E402
from golemflavor_amd.model import Model  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_EVAL = 6 * 8 + 8     # 8*ndim read + 8 written, no fr / status blob
SCLK_PEAK_GHZ = 2.4
SETTLE_MS = 100.0                         # untimed steps before the warm-up, see main()
REF_BURNIN, REF_NSTEPS = 200, 1000       # the reference's own chain length: submitter/mc_texture_dag.py:33-39
FP64_ISSUE_PER_S = 256 * 4 * SCLK_PEAK_GHZ * 1e9 / 4.0   # wave-instructions/s: 256 CUs x 4 SIMDs, one fp64 VALU instruction per 4 cycles at 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--walkers", type=int, default=4096, help="walkers per ensemble (BASELINE config 2)")
    ap.add_argument("--ensembles", type=int, default=4096, help="independent ensembles stacked per launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true", help="skip the emcee-driven extras (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip the C3 / C4 / C5 sub-records (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="evaluations in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--scan-burnin", type=int, default=REF_BURNIN, help="burn-in steps of c4_scan_ref / c5_scan_ref")
    ap.add_argument("--scan-nsteps", type=int, default=REF_NSTEPS, help="stored steps of c4_scan_ref / c5_scan_ref")
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
    """Time the CPU oracle (checker, never the product) on a bounded sample of the same workload: with the 16 threads of a
    per-GPU share of this host (what rounds 1-3 reported) AND with every core this process may run on
    (`len(os.sched_getaffinity(0))`); `value` / `cores` are the LARGER configuration's, so `gpu_over_cpu` is against the whole
    host the box grants, and the other figure stays in the record."""
    from oracle import oracle as O
    om = O.make_model(ps, "SM_GAUSS", bestfit_fr=bf, smearing=0.02)
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:                      # pragma: no cover
        allowed = os.cpu_count() or 1
    share = min(allowed, 16)                  # a 16-core share per GPU: the figure of rounds 1-3
    th1 = theta[:min(len(theta), 200000)]
    t0 = time.perf_counter()
    O.lnprob_batch(om, th1)
    t1 = time.perf_counter() - t0
    rate1 = len(th1) / t1
    runs = []
    ref = None
    for cores in sorted({share, allowed}):
        n = sample or int(min(len(theta), max(200000, rate1 * cores * 4.0)))   # one pass: ~4 s of work at perfect scaling ...
        ths = theta[:n]
        passes, tm = 0, 0.0
        while passes == 0 or (tm < 3.0 and passes < 64 and not sample):        # ... repeated until 3 s have been timed
            t0 = time.perf_counter()
            out = O.lnprob_batch(om, ths, threads=cores)
            tm += time.perf_counter() - t0
            passes += 1
        runs.append({"cores": cores, "value": n * passes / tm, "evaluations": n * passes, "seconds": tm, "passes": passes})
        if ref is None or len(out) > len(ref):
            ref = out
    best = max(runs, key=lambda r: r["value"])
    # the Python reference itself never travels to the GPU box; its rate was measured in the build
    # container when the golden vectors were generated (tests/golden/golden_meta.json, 1 core)
    ref_rate = None
    try:
        meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_meta.json")))
        ref_rate = 1e6 / meta["timings"]["notebook_ln_prob_us"]
    except Exception:
        pass
    return {"value": best["value"], "unit": "evals/s", "cores": best["cores"], "kind": "port",
            "sample": "%d evaluations of the bench theta batch, oracle/golem_oracle.c (long double), %d threads; "
                      "single-thread rate %.3g evals/s on %d evaluations" % (best["evaluations"], best["cores"], rate1, len(th1)),
            "single_thread_value": rate1, "cores_allowed": allowed, "host_cpus": os.cpu_count(),
            "by_cores": runs,
            "reference_python_evals_per_s_1core_build_container": ref_rate,
            "note": "value = the faster of the two thread counts tried (a 16-core share, and every core this process may use); "
                    "the reference's own rate was timed in the build container (another machine): "
                    "gpu_over_reference_python_1core compares across machines"}, ref


def _claim_stdout():
    """Keep fd 1 clean for the ONE JSON line: RCCL prints banners to the C-level stdout, so
    everything else in this process is pointed at stderr; returns the stream the JSON line goes to."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


# ---------------------------------------------------------------------------------------------------------------
# sub-records: the other BASELINE configurations, device-resident, HIP-event timed, each a few seconds at most
def _timed(model, fn, reps, warm=2, warm_ms=60.0):
    """Average HIP-event time of `reps` back-to-back calls, after `warm` calls and at least `warm_ms` of them: from idle
    the chip needs tens of milliseconds under load to reach the clock it then holds (a 14 ms burst of the BSM kernel straight
    after a pause reads 10-15 % slow)."""
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    while warm_ms > 0 and (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(4):
            fn()
        model.sync()
    e0, e1 = model.event(), model.event()
    model.sync()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    model.sync()
    return e0.elapsed_ms(e1) / reps


def _profile_constants():
    """Per-bin VALU instruction counts of the flux-averaged kernel, from the committed rocprofv3 SQ_INSTS_VALU pass
    (profiles/bsm_instr.json, regenerated by profiles/run_profile_bsm.sh): they turn a rate into a fraction of the
    fp64 issue rate.  None when the file is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "bsm_instr.json")))
    except Exception:
        return None


def extra_c3(device):
    """C3: scripts/mc_unitary.py -- 1e7 Haar draws propagated, output resident on the device (24 B written per draw)."""
    n = 10_000_000
    with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1 / 3, 2 / 3, 0.0)), device=device) as m:
        d_fr = m.alloc(24 * n)
        ms = _timed(m, lambda: m.haar_draw_device(26, 0, n, None, d_fr.ptr), reps=50)
        d_ang = m.alloc(32 * n)
        ms_a = _timed(m, lambda: m.haar_draw_device(26, 0, n, d_ang.ptr, d_fr.ptr), reps=30)
    return {"workload": "gf_haar_draw_device, 1e7 draws per launch (seed 26), composition only / + the 4 angles",
            "draws": n, "kernel_ms": ms, "draws_per_s": n / ms * 1e3, "bytes_per_draw": 24,
            "GBps_algorithmic": 24 * n / ms / 1e6, "frac_of_hbm_peak": 24 * n / ms / 1e6 / HBM_PEAK_GBS,
            "with_angles": {"kernel_ms": ms_a, "draws_per_s": n / ms_a * 1e3, "bytes_per_draw": 56,
                            "frac_of_hbm_peak": 56 * n / ms_a / 1e6 / HBM_PEAK_GBS},
            "bound": "VALU issue, not HBM: 176 VALU instructions per wave of draws, a third of the issue time in the 34 v_mad_u64_u32 of "
                     "the two Philox4x32-10 blocks, whose stream is fixed (the oracle's, bit for bit); profiles/r03/haar.txt"}


def extra_bulk(device, ps, label, n=4 * 1024 * 1024):
    """C4 / C5 bulk: the flux-averaged (20-bin) posterior kernel on n walkers, theta ~ U(seed box), logLam ~ U over the
    scale range less its top 6 decades (dimension 6, texture OET, source (0,1,0)), with and without the status array."""
    dim, tex = 6, Texture.OET
    rng = np.random.default_rng(1)
    box = np.array(ps.seeds, dtype=float)
    th = rng.uniform(box[:, 0], box[:, 1], size=(n, len(ps)))
    lo, hi = Cf.SCALE_BOUNDARIES[dim]
    th[:, -1] = rng.uniform(lo, hi - 6, n)
    desc = compile_model(ps, "BSM_GAUSS", texture=tex, dimension=dim, binning=Cf.default_bin_edges(), source_ratio=(0., 1., 0.),
                         bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    out = {"workload": "%s: %d-column flux_averaged_BSMu posterior, %d walkers per launch, 20 energy bins, dimension 6, OET"
                       % (label, len(ps), n), "n": n, "nbins": 20}
    consts = _profile_constants()
    with Model(desc, device=device) as m:
        d_th = m.alloc(th.nbytes).upload(th)
        d_out, d_st = m.alloc(8 * n), m.alloc(4 * n)
        # the two modes in alternation, three bursts of 20 launches each (every burst behind _timed's warm-up); the median is
        # kept and the bursts are listed
        bursts = {"no_status": [], "with_status": []}
        for _ in range(3):
            for key, st in (("no_status", None), ("with_status", d_st.ptr)):
                bursts[key].append(_timed(m, lambda: m.lnprob_device(d_th.ptr, n, d_out.ptr, None, st), reps=20))
        for key, st in (("no_status", None), ("with_status", d_st.ptr)):
            ms = sorted(bursts[key])[1]
            rec = {"kernel_ms": ms, "evals_per_s": n / ms * 1e3, "bin_diag_per_s": 20 * n / ms * 1e3,
                   "GBps_algorithmic": n * (8 * len(ps) + 8 + (4 if st else 0)) / ms / 1e6}
            rec["frac_of_hbm_peak"] = rec["GBps_algorithmic"] / HBM_PEAK_GBS
            rec["kernel_ms_bursts"] = bursts[key]
            ipw = (consts or {}).get("%d_%s" % (len(ps), key), {}).get("valu_wave_instr_per_walker")
            if ipw:
                # wave-instructions retired per second over what 1024 SIMDs can issue (one fp64 instruction per 4 cycles)
                rec["valu_instr_per_walker"] = ipw
                rec["fp64_issue_fraction"] = (n / 64.0) * ipw / (ms * 1e-3) / FP64_ISSUE_PER_S
            out[key] = rec
        # where the unitarity verdict is not free: logLam over the FULL range of a texture that fails at its top (OEU)
        th[:, -1] = rng.uniform(lo, hi, n)
        desc2 = compile_model(ps, "BSM_GAUSS", texture=Texture.OEU, dimension=dim, binning=Cf.default_bin_edges(),
                              source_ratio=(0., 1., 0.), bestfit_fr=(1 / 3,) * 3, smearing=0.02)
    with Model(desc2, device=device) as m:
        n2 = n // 4
        d_th = m.alloc(th[:n2].nbytes).upload(th[:n2])
        d_out, d_st = m.alloc(8 * n2), m.alloc(4 * n2)
        ms = _timed(m, lambda: m.lnprob_device(d_th.ptr, n2, d_out.ptr, None, d_st.ptr), reps=3, warm=1, warm_ms=0)
        st = d_st.download((n2,), dtype=np.int32)
        out["with_status_through_the_failing_region"] = {
            "workload": "texture OEU, logLam over its whole range (the top fails the reference's unitarity assert): every "
                        "undecided (walker, bin) is re-evaluated in emulated x87 arithmetic", "n": n2, "kernel_ms": ms,
            "evals_per_s": n2 / ms * 1e3, "nonunitary_fraction": float(np.mean(st == 2))}
    out["bound"] = "fp64 VALU issue / dependency latency (HBM fraction << 1)"
    return out


def scan_setup(config, device):
    """(grid points, walkers, point factory, evaluations per grid point as a function of (burnin, nsteps))."""
    from golemflavor_amd import scan
    if config == "C4":
        pts = scan.texture_grid(6)
        nw = 2048
        make = lambda p, g: scan._TexturePoint(p, g, dimension=6, texture=Texture.OET, nwalkers=nw, device=device)  # noqa: E731
        evals = lambda burnin, nsteps: nw * (burnin + nsteps) + nw * nsteps   # noqa: E731  chain + post-processing of the stored steps
    else:
        pts = scan.sens_grid()
        nw = 512
        make = lambda p, g: scan._SensPoint(p, g, nwalkers=nw, device=device)  # noqa: E731
        evals = lambda burnin, nsteps: nw * (burnin + nsteps)                  # noqa: E731
    return pts, nw, make, evals


def reduce_phases(control, local):
    """Per-phase seconds of one scan: the MAX over ranks (a phase ends when its slowest rank ends), byte counts: the SUM.
    `local`: {name: number}; names ending in `_bytes` are summed.  Every rank calls it; the result is the same everywhere."""
    keys = sorted(local)
    parts = control.allgather_bytes(json.dumps({k: float(local[k]) for k in keys}).encode())
    rows = [json.loads(x.decode()) for x in parts]
    out = {}
    for k in sorted({k for r in rows for k in r}):
        vals = [r[k] for r in rows if k in r]
        out[k] = float(sum(vals)) if k.endswith("_bytes") else float(max(vals))
    return out


def scan_record_from_phases(config, world, n_points, walkers, burnin, nsteps, evals_total, seconds, phases, *, gather_kind,
                            rccl_init_s=None, chain_bytes_to_host=None, finite_fraction=None, nonunitary=None):
    """One `c4_scan*` / `c5_scan*` sub-record from plain numbers (no GPU needed: tests/test_bench_line.py)."""
    xgmi_s = phases.get("xgmi_s", 0.0)
    gbytes = phases.get("gather_bytes", 0.0)
    rec = {"workload": "%s: %d grid points x %d walkers, %d burn-in + %d stored steps, sharded over %d rank(s) (grid point g -> "
                       "rank (g + g div N) mod N), one stacked device sampler per rank" % (config, n_points, walkers, burnin, nsteps, world),
           "scaling": "strong (a fixed grid divided over the ranks)", "ranks": int(world), "grid_points": int(n_points),
           "walkers": int(walkers), "burnin": int(burnin), "nsteps": int(nsteps), "seconds": float(seconds),
           "evals": int(evals_total), "evals_per_s": evals_total / max(seconds, 1e-12),
           "rccl_init_s": rccl_init_s, "setup_s": phases.get("setup"), "sampling_s": phases.get("sampling"),
           "pack_s": phases.get("pack_s"), "xgmi_s": xgmi_s, "gather_bytes": int(gbytes),
           "gather_GBps": (gbytes / xgmi_s / 1e9) if (xgmi_s and gbytes) else None,
           "d2h_s": phases.get("d2h_s"), "d2h_bytes": int(phases.get("d2h_bytes", 0.0)), "gather": gather_kind,
           "sampling_evals_per_s": n_points * walkers * (burnin + nsteps) / max(phases.get("sampling") or seconds, 1e-12),
           "phases": {k: round(v, 4) for k, v in phases.items() if not k.endswith("_bytes")}}
    if chain_bytes_to_host is not None:
        rec["chain_bytes_to_host"] = int(chain_bytes_to_host)
    if finite_fraction is not None:
        rec["finite_fraction"] = float(finite_fraction)
    if nonunitary is not None:
        rec["nonunitary_proposals"] = nonunitary
    return rec


def device_gather_subphase(rccl, device, region, whole, rank, world, control):
    """The device-resident gather `north_star` names, as its OWN sub-phase (never part of a scan's `seconds`): every rank's
    block -- the bytes it has just delivered to the host segment, uploaded again -- goes GPU to GPU onto rank 0
    (`gf_comm_gather` over RCCL / xGMI, or hipIpc), timed barrier to barrier, and rank 0 checks what arrived against the
    segment (head and tail of every rank's block, bit patterns).  Every rank calls this; rank 0 gets the record."""
    nbytes = int(region.nbytes)
    rec = {"kind": getattr(rccl, "kind", "rccl"), "bytes_per_rank": nbytes, "bytes_into_root": nbytes * (world - 1), "ranks": world}
    err = ""
    stage = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=device)
    d_send = d_recv = None
    try:
        d_send = stage.alloc(nbytes).upload(np.ascontiguousarray(region))
        d_recv = stage.alloc(nbytes * world) if rank == 0 else None
    except Exception as exc:               # noqa: BLE001
        err = "%s: %s" % (type(exc).__name__, exc)
    errs = control.allgather_bytes(err.encode())        # nobody enters the gather unless everybody can
    err = next((e.decode() for e in errs if e), "")
    dt = 0.0
    if not err:
        control.barrier()
        t0 = time.perf_counter()
        try:
            rccl.gather_device(d_send.ptr, d_recv.ptr if rank == 0 else None, nbytes, 0)
        except Exception as exc:           # noqa: BLE001
            err = "%s: %s" % (type(exc).__name__, exc)
        dt = time.perf_counter() - t0
    dt = float(control.allreduce_max([dt])[0])
    errs = control.allgather_bytes(err.encode())
    err = next((e.decode() for e in errs if e), "")
    if rank == 0 and not err:
        piece = min(nbytes, 4 << 20) // 8 * 8
        ok = True
        for r in range(world):
            want = whole[r].reshape(-1).view(np.uint64)
            for off in sorted({0, nbytes - piece}):
                got = d_recv.download((piece // 8,), dtype=np.uint64, offset_bytes=r * nbytes + off)
                ok = ok and bool(np.array_equal(got, want[off // 8: off // 8 + piece // 8]))
        rec.update(seconds=dt, GBps=(nbytes * (world - 1) / dt / 1e9) if dt > 0 else None, verified=ok,
                   verified_how="head and tail (%d bytes each) of every rank's block, bit patterns against the host segment" % piece)
    if err:
        rec["error"] = err
    for d in (d_send, d_recv):
        if d is not None:
            d.free()
    stage.close()
    try:
        rec["comm_nranks"] = rccl.nranks()
    except Exception:                      # noqa: BLE001
        pass
    return rec if rank == 0 else None


def extra_scan(device, config, burnin, nsteps, rank=0, world=1, control=None, rccl=None, rccl_init_s=None, shared=False, arena=None):
    """C4 / C5 end to end at full size, the grid sharded over the ranks -- golemflavor_amd.scan's own code path: stacked device
    sampler per rank, post-processing (C4), and the delivery: `shared` (the ranks are on one node): every rank reads its own
    chains back over its own PCIe link into one host segment rank 0 maps too (scan.SharedHostGather), with the device gather
    (`gf_comm_gather` over RCCL / xGMI, or hipIpc) measured afterwards as a sub-phase of its own; otherwise the device gather
    to rank 0 and one download; no communicator: the host control plane.  EVERY rank calls this; rank 0 gets the record, the
    others None."""
    from golemflavor_amd import scan
    control = control or gdist.LocalBackend()
    pts, nw, make, evals = scan_setup(config, device)
    mine = gdist.shard(len(pts), rank, world)
    control.barrier()
    t0 = time.perf_counter()
    chains, gather_kind, local, g = None, None, {}, None
    if world > 1 and shared:
        g = scan.SharedHostGather(control, rank, world, arena=arena)
        chains = scan.run_points(pts, mine, make, burnin, nsteps, stacked=True, gather=g)
        local = dict(scan.PHASES)
        local.update({k: v for k, v in g.stats.items() if isinstance(v, (int, float)) and k not in ("ranks", "slots_per_rank")})
        gather_kind = g.stats.get("delivery", g.kind)
        if g.stats.get("note"):
            gather_kind += ": " + g.stats["note"]
    elif world == 1 or rccl is not None:
        stage = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=device)
        g = scan.DeviceGather(rccl, rank, world, stage, control=control)
        chains = scan.run_points(pts, mine, make, burnin, nsteps, stacked=True, gather=g)
        stage.close()
        local = dict(scan.PHASES)
        local.update({k: v for k, v in g.stats.items() if isinstance(v, (int, float)) and k not in ("ranks", "slots_per_rank")})
        gather_kind = "device -> host (one rank)"
        if rccl is not None:
            gather_kind = ("rccl gather to rank 0 over xGMI (gf_comm_gather), one download" if getattr(rccl, "kind", "rccl") == "rccl" else
                           "hipIpc gather to rank 0, device to device (gf_ipc_gather: RCCL unavailable), one download")
        if g.stats.get("note"):
            gather_kind += ": " + g.stats["note"]
    else:
        # no communicator (reported in the line as rccl_error): the blocks go to rank 0 through the host control plane
        loc = scan.run_points(pts, mine, make, burnin, nsteps, stacked=True)
        local = dict(scan.PHASES)
        t1 = time.perf_counter()
        chains = gdist.gather_chains_to_root(loc, len(pts), control)
        local["host_gather_s"] = time.perf_counter() - t1
        gather_kind = "host control plane (tcp) -- RCCL unavailable"
    local["rank_seconds"] = time.perf_counter() - t0
    control.barrier()
    seconds = time.perf_counter() - t0                      # barrier to barrier: the slowest rank, the delivery included
    phases = reduce_phases(control, local)
    seconds = float(control.allreduce_max([seconds])[0])
    # every rank's own numbers next to the reductions: how many links carried the result, and how long each was busy
    per_rank = control.allgather(np.array([local.get("d2h_s", 0.0), local.get("d2h_bytes", 0.0), local.get("sampling", 0.0),
                                           local["rank_seconds"]], dtype=np.float64))
    sub = None
    if world > 1 and shared and rccl is not None and getattr(g, "region", None) is not None and g.stats.get("delivery", "").startswith("shared"):
        sub = device_gather_subphase(rccl, device, g.region, g._all if rank == 0 else None, rank, world, control)
    if rank != 0:
        if isinstance(g, scan.SharedHostGather):
            g.release()
        return None
    nbytes = sum(c.nbytes for c in chains)
    finite = scan.finite_fraction(chains)
    rec = scan_record_from_phases(config, world, len(pts), nw, burnin, nsteps, len(pts) * evals(burnin, nsteps), seconds, phases,
                                  gather_kind=gather_kind, rccl_init_s=rccl_init_s, chain_bytes_to_host=nbytes,
                                  finite_fraction=finite, nonunitary=dict(scan.LAST_NONUNITARY) if scan.LAST_NONUNITARY else None)
    rec["d2h_links"] = int(np.count_nonzero(per_rank[:, 1] > 0)) if world > 1 else 1
    if world == 1:
        ar = scan._ARENA.get("arena")
        rec["destination"] = ("registered result arena (DMA straight into it)" if ar is not None and ar.registered and ar.nbytes >= nbytes else
                              "fresh memory on 2 MiB pages, through the pinned ring and the host's copy threads")
    elif isinstance(g, scan.SharedHostGather) and g.seg is not None:
        rec["destination"] = "host segment%s" % (", registered by every rank (DMA straight into it)" if getattr(g.seg, "_registered", None) else "")
    rec["d2h_s_per_rank"] = [round(float(x), 4) for x in per_rank[:, 0]]
    rec["d2h_bytes_per_rank"] = [int(x) for x in per_rank[:, 1]]
    rec["sampling_s_per_rank"] = [round(float(x), 4) for x in per_rank[:, 2]]
    rec["seconds_per_rank"] = [round(float(x), 4) for x in per_rank[:, 3]]
    if sub is not None:
        rec["device_gather_subphase"] = sub
    if isinstance(g, scan.SharedHostGather):
        g.release()
    return rec


def extra_c5_sampler(device, burnin=100, nsteps=200):
    """The C5 scan's SAMPLING alone (256 chains x 512 walkers of the 12-column posterior, chain kept on the device): what the device
    sampler delivers where nothing hides behind a read-back.  With the launch shape it chose (one workgroup per chain, or per-half-step
    grid kernels: mcmc.DeviceEnsembleSampler.launch_shape), the census of undecided proposals, and the fraction of the fp64 issue
    rate that arrives as posterior evaluations (the evaluation's own VALU count per walker from profiles/bsm_instr.json -- the
    emulated-x87 settling of undecided proposals is not counted as useful work)."""
    from golemflavor_amd import mcmc as mcmc_utils
    pts, nw, make, evals = scan_setup("C5", device)
    jobs = [make(p, g) for g, p in enumerate(pts)]
    smp = mcmc_utils.DeviceEnsembleSampler(nw, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
    smp.on_nonunitary = "-inf"
    t0 = time.perf_counter()
    smp.run_mcmc(np.stack([j.p0 for j in jobs]), burnin, storechain=False)
    t1 = time.perf_counter()
    smp.reset()
    smp.run_mcmc(None, nsteps)
    t2 = time.perf_counter()
    n_eval = len(pts) * nw * nsteps
    rec = {"workload": "C5 sampling only: %d chains x %d walkers, %d burn-in + %d stored steps, chain kept on the device" % (len(pts), nw, burnin, nsteps),
           "burnin_s": t1 - t0, "stored_run_s": t2 - t1, "seconds": t2 - t0, "us_per_half_step_stored_run": 1e6 * (t2 - t1) / (2 * nsteps),
           "evals_per_s_stored_run": n_eval / (t2 - t1), "launch_shape": smp.launch_shape(), "undecided": smp.undecided_census(),
           "nonunitary_proposals": int(smp.nonunitary_proposals)}
    ipw = ((_profile_constants() or {}).get("12_no_status", {}) or {}).get("valu_wave_instr_per_walker")
    if ipw:
        rec["valu_instr_per_walker"] = ipw
        rec["fp64_issue_fraction"] = (n_eval / 64.0) * ipw / (t2 - t1) / FP64_ISSUE_PER_S
    smp.close()
    for j in jobs:
        j.close()
    return rec


def extra_emcee(model, ps, walkers):
    """emcee-driven figures (never `value`): the device-resident stretch move."""
    from golemflavor_amd import mcmc as mcmc_utils
    out = {}
    rngp = np.random.default_rng(26)
    box = np.array(ps.seeds, dtype=np.float64)
    p0 = rngp.uniform(box[:, 0], box[:, 1], size=(walkers, 6))
    smp = mcmc_utils.DeviceEnsembleSampler(walkers, 6, model, seed=26)
    smp.run_mcmc(p0, 50, storechain=False)
    t0 = time.perf_counter()
    smp.run_mcmc(None, 500, storechain=False)
    dt = time.perf_counter() - t0
    out["emcee_driven"] = {"sampler": "device-resident stretch move", "walkers": walkers, "chains": 1,
                           "steps": 500, "us_per_step": 1e6 * dt / 500, "evals_per_s": walkers * 500 / dt,
                           "acceptance_fraction": float(np.mean(smp.acceptance_fraction))}
    smp.close()
    # BASELINE.json's metric as worded -- 100-walker chains (configs[0]) -- one workgroup per ensemble, walkers in LDS,
    # a run is one launch.  One chain is ONE wave's dependent-instruction chain: a latency, not a throughput; chains side
    # by side fill the machine.  Model: t_step(k chains) = max(L1, k / R_inf), L1 = the single-chain step latency,
    # R_inf = chain-steps per second with every CU busy.
    rows = []
    for nch, steps in ((1, 20000), (16, 20000), (256, 4000), (4096, 600)):
        p1 = rngp.uniform(box[:, 0], box[:, 1], size=(nch, 100, 6))
        smp = mcmc_utils.DeviceEnsembleSampler(100, 6, model, nchains=nch, seed=26)
        smp.run_mcmc(p1 if nch > 1 else p1[0], 100, storechain=False)
        e0, e1 = model.event(), model.event()
        t0 = time.perf_counter()
        e0.record()
        smp.run_async(None, steps, storechain=False)      # the run: enqueue ...
        e1.record()
        smp.wait()                                        # ... and wait for it (what run_mcmc does, less the state read-back)
        dt = time.perf_counter() - t0
        smp.state                                         # positions + lnprob back on the host, once per run
        dt_state = time.perf_counter() - t0 - dt
        rows.append({"chains": nch, "steps": steps, "us_per_step": 1e6 * dt / steps, "evals_per_s": 100.0 * nch * steps / dt,
                     "kernel_us_per_step": 1e3 * e0.elapsed_ms(e1) / steps, "state_readback_ms": 1e3 * dt_state,
                     "acceptance_fraction": float(np.mean(smp.acceptance_fraction))})
        smp.close()
    l1 = rows[0]["us_per_step"]
    rinf = rows[-1]["chains"] / (rows[-1]["us_per_step"] * 1e-6)
    for r in rows:
        r["model_us_per_step"] = max(l1, r["chains"] / rinf * 1e6)
    out["emcee_driven_c1_scaling"] = {
        "sampler": "device-resident stretch move, one workgroup per 100-walker ensemble (k_stretch_persist)",
        "rows": rows, "single_chain_step_latency_us": l1, "chain_steps_per_s_saturated": rinf,
        "model": "us_per_step(k) = max(L1, k / R_inf): one chain is bound by one wave's dependent instructions "
                 "(~%.0f us per step at 2 half-steps), %d chains by the machine's throughput" % (l1, rows[-1]["chains"]),
        "evals_per_s_1_chain": rows[0]["evals_per_s"], "evals_per_s_saturated": rows[-1]["evals_per_s"]}
    return out


def reduce_step_timing(control, elapsed, kernel_ms):
    """The job's step time is the MAX over ranks (contract); so is the kernel's average launch duration."""
    e, k = [float(x) for x in control.allreduce_max([elapsed, kernel_ms])]
    return e, k


def load_traffic(n):
    """HBM traffic per launch: FETCH_SIZE / WRITE_SIZE need their own rocprofv3 --pmc passes (they cannot be read inside
    this run); the number is the one of the committed pass of this same command and batch size (profiles/traffic.json,
    regenerated by profiles/run_profile.sh), None when that file is missing or does not match."""
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tp))
        if int(tj.get("n", -1)) == n:
            return tj.get("traffic_bytes_per_launch"), ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                        "of this command (not this run)")
    except Exception:                      # noqa: BLE001
        pass
    return None, None


def assemble_line(*, world, steps, warmup, walkers, ensembles, elapsed, kernel_ms, control_plane, librccl, traffic=None,
                  traffic_src=None, gathered_ok=None, rccl_error=None, rccl_init_s=None, extras=None, cpu=None, parity=None, device_gather=None,
                  overrides="", rccl_nranks=None):
    """The ONE JSON line, from plain numbers (no GPU, no library: tests/test_bench_line.py).  `elapsed`, `kernel_ms`: already
    reduced over the ranks (reduce_step_timing).  `extras`: sub-records by key.  `cpu`: the cpu_baseline record."""
    n = walkers * ensembles
    evals = float(n) * steps * world
    value = evals / elapsed
    ach = BYTES_PER_EVAL * n / (kernel_ms * 1e-3) / 1e9
    metric = "walker-lnprob evals/sec (Gaussian llh, 100 walkers) at 1/2/4/8 MI355X"
    try:                                   # BASELINE.json's own wording when the file travelled with the repo
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:                      # noqa: BLE001
        pass
    out = {
        "metric": metric, "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "settle_ms_before_warmup": SETTLE_MS,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2: examples/inference.ipynb 6-dim Gaussian-llh posterior, %d-walker ensembles, "
                               "%d independent ensembles stacked per launch per GPU (theta resident in HBM)"
                               % (walkers, ensembles),
                   "walkers_per_ensemble": walkers, "ensembles_per_launch_per_gpu": ensembles,
                   "evals_per_step_per_gpu": n, "ndim": 6, "parallelism": "independent ensembles sharded over %d GPU(s)" % world,
                   "note": "BASELINE.json words the metric on its configs[0] (100-walker chain, CPU plumbing); the bench line "
                           "is configs[1], 4096-walker ensembles on the GPU; the 100-walker chains themselves are in "
                           "emcee_driven_c1_scaling; configs[3] and [4] (the sharded grid scans) are c4_scan_ref / c5_scan_ref"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kernel_ms, "bytes_per_eval": BYTES_PER_EVAL,
                     "kernel": "k_lnprob_sm_fast<6, SM_GAUSS, canonical, no fr>"},
        "control_plane": control_plane, "librccl": librccl,
        "diagnostic_overrides": overrides,     # GF_* environment overrides the library honoured in this process ("" = none)
    }
    if rccl_init_s is not None:
        out["rccl_init_s"] = rccl_init_s
    if gathered_ok is not None:
        out["rccl_gather_ok"] = gathered_ok
    if device_gather is not None:
        out["device_gather"] = device_gather          # "rccl" | "hipIpc": the device-to-device path between the ranks' GPUs
    if rccl_nranks is not None:
        # ranks the communicator itself reports (ncclCommCount on an RCCL communicator; the world size on the hipIpc stand-in),
        # one entry per rank: "did RCCL see N ranks?" is answered by this list being [N] * N with device_gather == "rccl"
        out["rccl_nranks"] = rccl_nranks
    if rccl_error is not None:
        out["rccl_error"] = rccl_error
    for key, rec in (extras or {}).items():
        out[key] = rec
    if parity is not None:
        out["parity_max_rel_vs_oracle"] = parity
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = value / cpu["value"]                          # vs the oracle port on this box's cores
        refrate = cpu.get("reference_python_evals_per_s_1core_build_container")
        if refrate:                                                         # vs the reference itself (timed where it can run)
            out["gpu_over_reference_python_1core"] = value / refrate
    return out


def main():
    """`run`, with the control plane's failures turned into a line: a peer that stops answering (dist.ControlPlaneTimeout after
    GF_CONTROL_TIMEOUT seconds, well under the driver's limit) or goes away (ConnectionError) ends the job with an `error` in
    rank 0's JSON line and exit status 4 on every rank that noticed -- not with a run killed at its time limit."""
    a = parse()
    json_out = _claim_stdout()
    try:
        run(a, json_out)
    except (gdist.ControlPlaneTimeout, ConnectionError) as exc:
        msg = "%s: %s" % (type(exc).__name__, exc)
        sys.stderr.write("bench.py: control plane failure on rank %s: %s\n" % (os.environ.get("RANK", "0"), msg))
        if int(os.environ.get("RANK", "0")) == 0:
            json_out.write(json.dumps({"metric": "walker-lnprob evals/sec (Gaussian llh, 100 walkers) at 1/2/4/8 MI355X", "value": None,
                                       "unit": "evals/s", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "steps": a.steps,
                                       "warmup": a.warmup, "error": msg}) + "\n")
            json_out.flush()
        sys.stderr.flush()
        os._exit(4)                        # helper threads (RCCL bootstrap) must not hold the exit


def run(a, json_out):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "GF_BENCH_DEVICE" in os.environ:      # rehearsal on a 1-GPU box: every rank on the same device
        local_rank = int(os.environ["GF_BENCH_DEVICE"])
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    L = _lib.lib()                           # libgolemhip.so (and with it /opt/rocm's runtime + librccl) before anything else ROCm
    control = gdist.SocketBackend(rank, world) if world > 1 else gdist.LocalBackend()
    rccl, rccl_error, stuck, rccl_init_s, shared, arena, host_segment = None, None, False, None, False, None, None
    ps, bf, desc = notebook_descriptor()
    if world > 1:
        # RCCL communicator of the library itself; its unique id travels over the socket control plane.  The timed
        # region has no collective, so an RCCL problem must not cost the measurement: the descriptor then goes over the
        # control plane, the failure is reported in the JSON line AND in the exit status.
        t_r = time.perf_counter()
        # (open_device_gather: where the communicator cannot be set up and the ranks share a node, the chain blocks still go
        # device to device, through hipIpc; `rccl.kind` says which)
        rccl, rccl_error, stuck = gdist.open_device_gather(rank, world, local_rank, control,
                                                           timeout=float(os.environ.get("GF_RCCL_TIMEOUT", "60")))
        rccl_init_s = float(control.allreduce_max([time.perf_counter() - t_r])[0])
        # one node (the contract's case)?  Then the scans deliver over every rank's own PCIe link into one host segment -- created
        # HERE, once, for both scans: its 4 KiB shmem pages are allocated (and zeroed) by rank 0 at 17 GB/s at best, i.e. 0.7 s for
        # the C5 result, which no scan could hide; beside the job's start-up it costs nothing (reported as `host_segment`)
        shared = gdist.same_node(control) and not os.environ.get("GF_SCAN_DEVICE_GATHER")
        if shared and not a.no_extras:
            from golemflavor_amd import scan as _scan
            t_s = time.perf_counter()
            need = max(_scan.segment_bytes(64, world, 2048, a.scan_nsteps, 9), _scan.segment_bytes(256, world, 512, a.scan_nsteps, 12))
            arena = gdist.HostSegment(control, need)
            seg_create_s = time.perf_counter() - t_s
            arena.wait_allocated()                 # (rank 0: the background posix_fallocate; done before anything is timed)
            control.barrier()
            host_segment = {"bytes": int(need), "kind": arena.kind, "error": arena.error, "create_and_map_s": round(seg_create_s, 4),
                            "allocate_s": round(float(control.allreduce_max([time.perf_counter() - t_s])[0]), 4),
                            "note": "one host segment for the job's scans, created and allocated (posix_fallocate, rank 0) before the timed "
                                    "region; every scan's result goes into it, every rank through its own PCIe link"}
        # fixed physics constants: rank 0's packed descriptor is the one everybody uses
        desc = gdist.broadcast_descriptors([desc] if rank == 0 else [], rccl if getattr(rccl, "kind", None) == "rccl" else control)[0]

    n = a.walkers * a.ensembles
    model = Model(desc, device=local_rank)
    theta = synth_theta(ps, n, seed=26 + rank)
    d_theta = model.alloc(theta.nbytes).upload(theta)
    d_out = model.alloc(8 * n)

    def step():
        model.lnprob_device(d_theta.ptr, n, d_out.ptr, None, None)

    def fence():
        model.sync()
        control.barrier()
        model.sync()

    # settle the clock: from idle the chip needs tens of milliseconds under load to reach the state it then holds, and a
    # short run (--steps 20 is 3.5 ms) would otherwise be timed inside that ramp.  Untimed, ahead of the W warm-up steps.
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < SETTLE_MS:
        for _ in range(16):
            step()
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
    elapsed, kernel_ms = reduce_step_timing(control, elapsed, kernel_ms)

    # after the timed region: one chain block (the first ensemble's lnprob) from every rank to rank 0 over RCCL
    gathered_ok, rccl_nranks = None, None
    if rccl is not None:
        try:
            blk = 8 * a.walkers
            d_all = model.alloc(blk * world) if rank == 0 else None
            rccl.gather_device(d_out.ptr, d_all.ptr if rank == 0 else None, blk, 0)
            gathered_ok = True
            if rank == 0:
                allv = d_all.download((world, a.walkers))
                mine = d_out.download((a.walkers,))
                gathered_ok = bool(np.array_equal(allv[0], mine, equal_nan=True) and np.all(np.isfinite(allv) | np.isinf(allv)))
        except Exception as exc:           # noqa: BLE001
            rccl_error = "%s: %s" % (type(exc).__name__, exc)
            gathered_ok = False
        oks = control.allgather_bytes(b"1" if gathered_ok else b"0")
        gathered_ok = all(x == b"1" for x in oks)
        try:                                 # how many ranks the COMMUNICATOR saw (ncclCommCount), as every rank reports it
            seen = rccl.nranks()
        except Exception:                  # noqa: BLE001
            seen = -1
        rccl_nranks = [int(x) for x in control.allgather(np.array([seen], dtype=np.int64)).reshape(-1)]

    # sub-records.  The sharded scans run on EVERY rank (rank 0 assembles); the rest is rank 0's
    extras = {}
    # Where the scans' results go: memory REGISTERED with the HIP runtime once per job, so that every read-back is a DMA straight into
    # it at the speed of the link (gf_host_register, ABI 5; profiles/r04/host_register.txt) -- at N > 1 every rank registers its
    # mapping of the job's host segment, at N = 1 a process-lifetime arena (scan.ResultArena).  Set up here, outside every timed
    # region, and reported (`host_segment` / `result_arena`); what a scan costs into FRESH memory is reported beside it at N = 1.
    result_arena = None
    if not a.no_extras and not os.environ.get("GF_BENCH_NO_ARENA"):
        from golemflavor_amd import scan as _scan
        if arena is not None and arena.error is None:
            ok, secs = arena.register()
            regs = control.allgather(np.array([1.0 if ok else 0.0, secs], dtype=np.float64))
            host_segment["registered_ranks"] = int(regs[:, 0].sum())
            host_segment["register_s"] = round(float(regs[:, 1].max()), 4)
            if getattr(arena, "register_error", None):
                host_segment["register_error"] = arena.register_error
        elif world == 1:
            need = max(_scan.segment_bytes(64, 1, 2048, max(a.scan_nsteps, 200), 9), _scan.segment_bytes(256, 1, 512, max(a.scan_nsteps, 200), 12))
            result_arena = _scan.ResultArena(need)
            _scan.set_result_arena(result_arena)
            extras["result_arena"] = {"bytes": int(result_arena.nbytes), "registered": bool(result_arena.registered),
                                      "set_up_s": round(result_arena.seconds, 4), "error": result_arena.register_error,
                                      "note": "one registered block of host memory for the job's scans (2 MiB pages, mapped and pinned "
                                              "once, outside every timed region): their read-backs are DMA straight into it"}

    def guarded(key, fn, collective=False):
        try:
            rec = fn()
        except Exception as exc:           # noqa: BLE001
            if collective and world > 1:
                raise                      # a rank that left a collective half-way cannot be papered over
            rec = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if rec is not None:
            extras[key] = rec

    if world == 1 and rank == 0 and not a.no_sampler:
        try:
            extras.update(extra_emcee(model, ps, a.walkers))
        except Exception as exc:           # noqa: BLE001
            extras["emcee_driven"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if not a.no_extras:
        if world == 1:
            guarded("c3", lambda: extra_c3(local_rank))
            guarded("c4_bulk", lambda: extra_bulk(local_rank, Cf.texture_paramset(6), "C4"))
            guarded("c5_bulk", lambda: extra_bulk(local_rank, Cf.fr_paramsets(6, (0.4444, 0.0))[1], "C5"))
            guarded("c4_scan", lambda: extra_scan(local_rank, "C4", 100, 200))
            guarded("c5_scan", lambda: extra_scan(local_rank, "C5", 100, 200))
            guarded("c5_sampler", lambda: extra_c5_sampler(local_rank))
        for key, cfg in (("c4_scan_ref", "C4"), ("c5_scan_ref", "C5")):
            guarded(key, lambda cfg=cfg: extra_scan(local_rank, cfg, a.scan_burnin, a.scan_nsteps, rank, world, control, rccl,
                                                    rccl_init_s, shared, arena), collective=True)
        if result_arena is not None:
            # the same two scans into FRESH memory (what a process pays that produces one result and exits)
            _scan.set_result_arena(None)
            for key, cfg in (("c4_scan_ref", "C4"), ("c5_scan_ref", "C5")):
                if isinstance(extras.get(key), dict) and "error" not in extras[key]:
                    try:
                        r = extra_scan(local_rank, cfg, a.scan_burnin, a.scan_nsteps)
                        extras[key]["seconds_into_fresh_memory"] = r["seconds"]
                    except Exception as exc:   # noqa: BLE001
                        extras[key]["seconds_into_fresh_memory"] = "%s: %s" % (type(exc).__name__, exc)
            result_arena.close()
        if host_segment is not None:
            extras["host_segment"] = host_segment

    if rank == 0:
        cb, parity = None, None
        if not a.no_cpu_baseline:
            cb, ref = cpu_baseline(ps, bf, theta, a.cpu_sample)
            got = d_out.download((len(ref),))
            with np.errstate(all="ignore"):
                parity = float(np.max(np.abs(got - ref) / np.abs(ref)))
        traffic, traffic_src = load_traffic(n)
        out = assemble_line(world=world, steps=a.steps, warmup=a.warmup, walkers=a.walkers, ensembles=a.ensembles,
                            elapsed=elapsed, kernel_ms=kernel_ms,
                            control_plane="tcp sockets (golemflavor_amd.dist.SocketBackend)" if world > 1 else "none (1 rank)",
                            librccl=gdist.rccl_library_info(), traffic=traffic, traffic_src=traffic_src,
                            gathered_ok=gathered_ok, rccl_error=rccl_error, rccl_init_s=rccl_init_s, extras=extras, cpu=cb,
                            device_gather=getattr(rccl, "kind", None), rccl_nranks=rccl_nranks,
                            parity=parity, overrides=_lib.diagnostic_overrides())
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    if rccl is not None:
        rccl.close()
    model.close()
    try:
        control.barrier()
    except Exception:                      # noqa: BLE001
        pass
    control.close()
    if rccl_error is not None:
        # the measurement is complete and printed; a broken RCCL path still fails the command
        sys.stderr.write("bench.py: RCCL problem: %s\n" % rccl_error)
        sys.stderr.flush()
        if stuck:
            os._exit(3)                    # a helper thread is still inside ncclCommInitRank: it would block a normal exit
        sys.exit(3)


if __name__ == "__main__":
    main()
