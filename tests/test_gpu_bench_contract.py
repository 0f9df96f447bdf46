"""bench.py's one JSON line: the keys the driver reads, consistent with each other (a short run of the real command)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "7", "--warmup", "2", "--no-sampler", "--no-extras",
           "--cpu-sample", "20000"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                        # ONE line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["evals_per_step_per_gpu"]
    assert d["value"] == pytest.approx(n / (d["ms_per_step"] * 1e-3), rel=1e-6)          # whole-job rate = evals / step time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.3 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(r["bytes_per_eval"] * n / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02                                     # the kernel fits inside the step
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["bytes_per_eval"] * n
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == d["unit"]
    assert d["parity_max_rel_vs_oracle"] < 1e-10


@pytest.mark.gpu
def test_bench_line_sub_records_at_reduced_scan_length():
    """The sub-records of the line through the real command, the sharded scans at a reduced chain length (the full
    reference length is exercised by tests/test_gpu_mcmc.py::test_reference_length_scan_on_one_gpu): every BASELINE
    configuration is present, none carries an `error`, and the scan records have the per-phase keys of the N-rank line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-sample", "20000",
           "--scan-burnin", "20", "--scan-nsteps", "40"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("c3", "c4_bulk", "c5_bulk", "c4_scan", "c5_scan", "c4_scan_ref", "c5_scan_ref", "emcee_driven", "emcee_driven_c1_scaling",
                "cpu_baseline", "roofline"):
        assert key in d and "error" not in d[key], (key, d.get(key))
    assert d["diagnostic_overrides"] == ""
    for key, npts, nw in (("c4_scan_ref", 64, 2048), ("c5_scan_ref", 256, 512)):
        s = d[key]
        assert s["ranks"] == 1 and s["grid_points"] == npts and s["walkers"] == nw and s["burnin"] == 20 and s["nsteps"] == 40
        assert s["scaling"].startswith("strong")
        for k in ("seconds", "evals_per_s", "setup_s", "sampling_s", "pack_s", "xgmi_s", "gather_bytes", "d2h_s", "gather", "phases"):
            assert k in s, (key, k)
        assert s["evals_per_s"] == pytest.approx(s["evals"] / s["seconds"])
        assert s["finite_fraction"] > 0.9
    assert d["c5_scan_ref"]["nonunitary_proposals"]["settled"].startswith("on the device")
    # round 4 (ABI 5): the job's scans deliver into ONE block of host memory registered with the runtime before anything is timed;
    # what the reference-length scans cost into fresh memory is reported beside them
    ar = d["result_arena"]
    assert ar["registered"] is True and ar["error"] is None and ar["bytes"] >= 256 * 512 * 200 * 12 * 8 and ar["set_up_s"] > 0
    for key in ("c4_scan", "c5_scan", "c4_scan_ref", "c5_scan_ref"):
        assert d[key]["destination"].startswith("registered result arena"), d[key]["destination"]
    for key in ("c4_scan_ref", "c5_scan_ref"):
        assert isinstance(d[key]["seconds_into_fresh_memory"], float) and d[key]["seconds_into_fresh_memory"] > 0
    assert d["c5_scan_ref"]["nonunitary_proposals"]["host_thread_times"]["blocks"] >= 1
    fr = d["c4_bulk"]["with_status_through_the_failing_region"]
    assert fr["evals_per_s"] > 3e8 and 0.1 < fr["nonunitary_fraction"] < 0.3          # round 2: 1.5e8


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_bench_line_of_several_ranks_on_one_gpu(world, tmp_path):
    """The N > 1 line through the real command and launcher, `world` ranks on the ONE GPU (GF_BENCH_DEVICE=0), scans at a reduced
    chain length: RCCL refuses the duplicate device (reported, exit status 3, as designed), the ranks fall back to hipIpc for the
    device path, and the scans deliver as they would on `world` GPUs -- every rank reads its own chains back into the job's host
    segment (`host_segment`), `d2h_links` == world with every rank's own byte count, and the device gather runs afterwards as a
    sub-phase of its own whose result is checked against the segment.  What RCCL over xGMI does with N ranks this cannot show."""
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, GF_BENCH_DEVICE="0", GF_RCCL_TIMEOUT="30", PYTHONDONTWRITEBYTECODE="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "2",
           "--cpu-sample", "20000", "--scan-burnin", "20", "--scan-nsteps", "40"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, (res.returncode, res.stdout[-500:], res.stderr[-1500:])
    d = json.loads(lines[0])
    assert res.returncode != 0 and "ncclCommInitRank" in d["rccl_error"]         # two ranks on one device: RCCL says no, the line says so
    assert d["n_gpus"] == world and d["device_gather"] == "hipIpc" and d["rccl_nranks"] == [world] * world
    assert d["host_segment"]["error"] is None and d["host_segment"]["bytes"] >= 256 * 512 * 40 * 12 * 8
    assert d["host_segment"]["registered_ranks"] == world                       # ABI 5: every rank's mapping registered: DMA straight into it
    n = d["config"]["evals_per_step_per_gpu"]
    assert d["value"] == pytest.approx(world * n / (d["ms_per_step"] * 1e-3), rel=1e-6)   # the whole job: every rank's evaluations
    for key, npts, per in (("c4_scan_ref", 64, 2048 * 40 * 9 * 8), ("c5_scan_ref", 256, 512 * 40 * 12 * 8)):
        s = d[key]
        assert "error" not in s and s["ranks"] == world and s["gather"].startswith("shared host segment")
        assert s["d2h_links"] == world and s["gather_bytes"] == 0 and "registered by every rank" in s["destination"]
        from golemflavor_amd import dist as gdist
        assert s["d2h_bytes_per_rank"] == [len(gdist.shard(npts, r, world)) * per for r in range(world)]
        assert s["d2h_bytes"] == npts * per == s["chain_bytes_to_host"]        # the result, once: summed over the ranks' links
        sub = s["device_gather_subphase"]
        assert sub["kind"] == "hipIpc" and sub["verified"] is True and sub["comm_nranks"] == world
        assert sub["bytes_into_root"] == sub["bytes_per_rank"] * (world - 1)
        assert s["finite_fraction"] > 0.9
