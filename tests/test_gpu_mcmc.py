"""GPU: the emcee-driven call surface end to end (BASELINE config 1 plumbing) and the RCCL binding."""
import ctypes as C
import os

import numpy as np
import pytest

from common import BIN_EDGES, bsm_args, notebook_sets
from golemflavor_amd import _lib
from golemflavor_amd import configs as Cf
from golemflavor_amd import dist as gdist
from golemflavor_amd import llh as llh_utils
from golemflavor_amd import mcmc as mcmc_utils
from golemflavor_amd.enums import Texture

pytestmark = pytest.mark.gpu


def test_lnprob_callable_keeps_reference_conventions(golden):
    asimov, ps = notebook_sets(golden)
    f = llh_utils.notebook_ln_prob(asimov, ps)
    v = f([0.307, 0.9564, 0.538, 4.08404, 0.9, 0.1])
    assert isinstance(v, float) and v == pytest.approx(-355.3852856116068, rel=1e-12)
    assert f([1.2, 0.9564, 0.538, 4.08404, 0.9, 0.1]) == -np.inf
    with pytest.raises(AssertionError):
        f([0.3, 0.9, 0.5])
    out = f(golden["g6_theta"][:10])
    assert out.shape == (10,) and np.allclose(out, golden["g6_lnprob"][:10], rtol=1e-12)
    assert llh_utils.lnprior([0.307, 0.9564, 0.538, 4.08404, 0.9, 0.1], ps) == pytest.approx(10.781874524028385, rel=1e-13)
    f.close()


def test_config1_notebook_chain_through_mcmc(golden, capsys):
    """100 walkers x 6 dims through mcmc.mcmc(): one launch per half-ensemble; acceptance fraction and
    posterior in line with the reference notebook (mean acceptance 0.427, examples/inference.ipynb:429-438)."""
    asimov, ps = notebook_sets(golden)
    f = llh_utils.notebook_ln_prob(asimov, ps)
    np.random.seed(26)
    p0 = mcmc_utils.flat_seed(ps, nwalkers=100)
    samples = mcmc_utils.mcmc(p0=p0, ln_prob=f, ndim=6, nwalkers=100, burnin=400, nsteps=1200, threads=1,
                              device_resident=False)
    out = capsys.readouterr().out
    assert samples.shape == (100 * 1200, 6)
    assert f.ncalls == 2 + 2 * 1600 and f.nevals == 200 + 100 * 1600
    acc = float(out.split("sum of acceptance fraction")[1].split()[0]) / 100
    assert 0.36 < acc < 0.50
    # mixing parameters stay within their (truncated) Gaussian priors (sigma 0.013, 0.00147 around the nominal values)
    assert samples[:, 0].mean() == pytest.approx(0.307, abs=0.02) and samples[:, 0].std() < 0.02
    assert samples[:, 1].mean() == pytest.approx((1 - 0.02206) ** 2, abs=0.001) and samples[:, 1].std() < 0.003
    # injected source (1,0,0): the source composition posterior prefers electron-rich sources
    src = samples[:, 4:6]
    fe = np.sqrt(src[:, 0]) * (1 + src[:, 1]) / 2
    assert fe.mean() > 0.6
    f.close()


def test_bsm_callable_raises_on_nonunitary_like_reference():
    asimov, ps = Cf.fr_paramsets(6, (0.4444444444444444, 0.0))      # fr_to_angles((1,1,1))
    args = bsm_args(6, Texture.OEU, (1 / 3, 2 / 3, 0))
    f = llh_utils.bsm_ln_prob(args, asimov, ps, smearing=0.02)
    th = np.array([0.307, 0.9564, 0.538, 4.08404, 7.4e-23, 2.494e-21, 1.0, 0.5, 1.0, 6.9, 2.5, -50.0])
    v = f(th)
    assert np.isfinite(v) or v == -np.inf
    th_bad = th.copy()
    th_bad[11] = -30.0                                             # the reference raises here
    with pytest.raises(AssertionError, match="not unitary"):
        f(th_bad)
    g = llh_utils.bsm_ln_prob(args, asimov, ps, smearing=0.02, on_nonunitary="-inf")
    assert g(th_bad) == -np.inf
    f.close()
    g.close()


def test_rccl_binding_world_size_1():
    """The gf_comm_* entry points on a single rank: create, broadcast, all-gather, barrier."""
    b = gdist.RcclBackend(rank=0, world=1, device=0)
    payload = bytes(range(256)) * 5
    assert b.broadcast_bytes(payload, 0) == payload
    from golemflavor_amd.descriptor import compile_model
    from golemflavor_amd.model import Model
    with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY")) as m:
        x = np.arange(1000, dtype=np.float64).reshape(250, 4)
        got = b.allgather(x, m)
        assert got.shape == (1, 250, 4) and np.array_equal(got[0], x)
        chains = gdist.gather_chains({0: x, 1: x + 1}, 2, b, allgather=lambda a: b.allgather(a, m))
        assert np.array_equal(chains[1], x + 1)
        # gather to root (ABI 3): on one rank the block lands in the root's buffer, equal to the local block
        d_send, d_recv = m.alloc(x.nbytes).upload(x), m.alloc(x.nbytes)
        b.gather_device(d_send.ptr, d_recv.ptr, x.nbytes, 0)
        assert np.array_equal(d_recv.download(x.shape), x)
        b.gather_device(d_send.ptr, d_send.ptr, x.nbytes, 0)           # in place: nothing to copy
        assert np.array_equal(d_send.download(x.shape), x)
        import ctypes as C
        L = _lib.lib()
        assert L.gf_comm_gather(b._h, d_send.ptr, None, x.nbytes, 0) == _lib.GF_ERR_INVALID_ARG    # the root needs a destination
        assert L.gf_comm_gather(b._h, d_send.ptr, d_recv.ptr, x.nbytes, 1) == _lib.GF_ERR_INVALID_ARG  # root outside the world
    b.barrier()
    b.close()


def test_grid_scans_c4_c5_smoke(capsys):
    """BASELINE configs C4 / C5 end to end on one rank, reduced sizes: sharded grid -> device sampler ->
    (C4) flux-averaged post-processing -> gathered chains."""
    import json
    from golemflavor_amd import scan
    scan.main(["--config", "C4", "--points", "3", "--nwalkers", "32", "--burnin", "10", "--nsteps", "15"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["chains_shape"] == [3, 32 * 15, 9] and out["finite_fraction"] > 0.9
    scan.main(["--config", "C5", "--points", "2", "--nwalkers", "32", "--burnin", "5", "--nsteps", "10"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["chains_shape"] == [2, 32 * 10, 12] and out["finite_fraction"] == 1.0 and out["gather"] == "device -> host"
    assert "librccl" in out["librccl"]                                 # the RCCL this process mapped is on record


def test_grid_scan_one_sampler_per_point_path(capsys):
    """--no-stack keeps one sampler per grid point, each on its own stream (the A/B baseline of the stacked path)."""
    import json
    from golemflavor_amd import scan
    for cfg, shape in (("C4", [3, 32 * 12, 9]), ("C5", [3, 32 * 12, 12])):
        scan.main(["--config", cfg, "--points", "3", "--nwalkers", "32", "--burnin", "6", "--nsteps", "12", "--no-stack"])
        out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert out["stacked"] is False and out["chains_shape"] == shape and out["finite_fraction"] > 0.9


def test_grid_scan_writes_reference_named_files(capsys, tmp_path):
    """--datadir: one .npy per grid point under the reference's naming scheme (misc.py:44-51)."""
    import json
    from golemflavor_amd import scan
    scan.main(["--config", "C4", "--points", "3", "--nwalkers", "32", "--burnin", "5", "--nsteps", "10",
               "--datadir", str(tmp_path / "chains"), "--outfile", str(tmp_path / "all")])
    capsys.readouterr()
    files = sorted(os.listdir(str(tmp_path / "chains")))
    assert files == ["mc_texture_DIM6_sfr_0.14_0.86_0.00_OET_logLam-56.000.npy", "mc_texture_DIM6_sfr_0.29_0.71_0.00_OET_logLam-56.000.npy",
                     "mc_texture_DIM6_sfr_0_1_0_OET_logLam-56.000.npy"], files
    allc = np.load(str(tmp_path / "all.npy"))
    lo = Cf.SCALE_BOUNDARIES[6][0]
    first = np.load(str(tmp_path / "chains" / ("mc_texture_DIM6_sfr_0_1_0_OET_logLam%+.3f.npy" % lo)))
    assert first.shape == (320, 9) and np.array_equal(first, allc[0], equal_nan=True)


def test_grid_scan_gathers_over_rccl(capsys, monkeypatch, tmp_path):
    """The chain gather of a scan through the library's RCCL communicator (one rank here; the same code path
    gathers to rank 0 over xGMI on N ranks): result identical to the local gather."""
    import json
    from golemflavor_amd import scan
    args = ["--config", "C4", "--points", "3", "--nwalkers", "32", "--burnin", "10", "--nsteps", "15"]
    scan.main(args + ["--outfile", str(tmp_path / "local")])
    capsys.readouterr()
    monkeypatch.setenv("GF_SCAN_RCCL", "1")
    scan.main(args + ["--outfile", str(tmp_path / "rccl")])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["gather"] == "rccl device gather to rank 0" and out["rccl_error"] is None
    a, b = np.load(str(tmp_path / "local.npy")), np.load(str(tmp_path / "rccl.npy"))
    assert a.shape == (3, 32 * 15, 9) and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("config", ["C4", "C5"])
def test_reference_length_scan_on_one_gpu(config):
    """BASELINE configs C4 / C5 at FULL size and at the reference's own chain length (burnin 200 + 1000 stored steps,
    submitter/mc_texture_dag.py:33-39), one rank: the chain buffer grows through its doublings (64 -> 1024 slots, repacked on
    the device), C4's 131 M stored samples go through the flux-averaged post-processing with the unitarity verdict, and
    7-13 GB of rows cross PCIe in chunks -- bench.py's c4_scan_ref / c5_scan_ref code path."""
    import bench
    rec = bench.extra_scan(0, config, bench.REF_BURNIN, bench.REF_NSTEPS)
    assert "error" not in rec
    npts, nw, width = (64, 2048, 9) if config == "C4" else (256, 512, 12)
    assert rec["ranks"] == 1 and rec["grid_points"] == npts and rec["burnin"] == 200 and rec["nsteps"] == 1000
    assert rec["chain_bytes_to_host"] == npts * nw * 1000 * width * 8
    assert rec["finite_fraction"] > (0.6 if config == "C4" else 0.999)      # C4's top scales sit in the failing region (NaN rows)
    assert rec["sampling_s"] > 0 and rec["d2h_s"] >= 0 and rec["seconds"] < 60.0
    if config == "C4":
        assert rec["d2h_s"] > 0                                   # C5's chain is read back during the run: what is left is a tail
    assert rec["evals_per_s"] > 1e8


_TWO_RANK_CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from golemflavor_amd import scan
out = sys.argv[2]
scan.main(["--config", sys.argv[3], "--points", "5", "--nwalkers", "32", "--burnin", "6", "--nsteps", "9"] +
          (["--outfile", out] if int(os.environ["RANK"]) == 0 else []))
"""


@pytest.mark.parametrize("config", ["C4", "C5"])
def test_scan_gathers_over_rccl_between_two_gpus(config, tmp_path, capsys):
    """The N > 1 data path on hardware: two ranks on two DISTINCT devices, descriptors and the RCCL id over the socket control
    plane, chains gathered to rank 0 with gf_comm_gather over xGMI (5 grid points on 2 ranks: ragged), one download.  The
    result must be the single-rank scan's, bit for bit.  Skipped on a one-GPU box -- where RCCL with more than one rank has
    never run (multi-GPU unmeasured)."""
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import socket
    import subprocess
    import sys
    from golemflavor_amd import scan
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scan.main(["--config", config, "--points", "5", "--nwalkers", "32", "--burnin", "6", "--nsteps", "9", "--outfile", str(tmp_path / "one")])
    capsys.readouterr()
    script = tmp_path / "child.py"
    script.write_text(_TWO_RANK_CHILD)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(tmp_path / "two"), config], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["ranks"] == 2 and line["gather"] == "rccl device gather to rank 0" and line["rccl_error"] is None
    a, b = np.load(str(tmp_path / "one.npy")), np.load(str(tmp_path / "two.npy"))
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("config,world", [("C4", 2), ("C5", 2), ("C5", 3)])
def test_scan_gathers_between_processes_on_one_gpu_over_hipipc(config, world, tmp_path, capsys):
    """The N > 1 DEVICE data path on the hardware this suite always has: `world` ranks in separate processes on ONE GPU.  RCCL
    refuses more than one rank per device (the error is reported and the exit status is 3, as designed), so the chain blocks
    take the fallback of `dist.open_device_gather`: shared between the processes as hipIpc handles over the socket control
    plane and copied device to device by rank 0 (gf_ipc_export / gf_ipc_gather), then downloaded once -- scan.DeviceGather's
    packing, slot mapping (5 grid points: ragged over 2 and 3 ranks) and root-only receive buffer with real device buffers.
    The result must be the single-rank scan's, bit for bit."""
    import json
    import socket
    import subprocess
    import sys
    from golemflavor_amd import scan
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scan.main(["--config", config, "--points", "5", "--nwalkers", "32", "--burnin", "6", "--nsteps", "9", "--outfile", str(tmp_path / "one")])
    capsys.readouterr()
    script = tmp_path / "child.py"
    script.write_text(_TWO_RANK_CHILD)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", GF_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GF_RCCL_TIMEOUT="30", PYTHONDONTWRITEBYTECODE="1",
                   GF_SCAN_DEVICE_GATHER="1")     # (round 4: ranks of one node deliver through a shared host segment unless told otherwise)
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(tmp_path / "many"), config], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 3 for p in procs), [(p.returncode, o[1][-1500:]) for p, o in zip(procs, outs)]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["ranks"] == world and line["gather"] == "hipIpc device gather to rank 0" and "ncclCommInitRank" in line["rccl_error"]
    assert line["gather_stats"]["gather_bytes"] > 0 and line["gather_stats"]["ranks"] == world
    # bytes into the root: every block but its own, counted once
    assert line["gather_stats"]["gather_bytes"] == line["gather_stats"]["block_bytes"] * (world - 1)
    a, b = np.load(str(tmp_path / "one.npy")), np.load(str(tmp_path / "many.npy"))
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("config,world", [("C4", 2), ("C5", 2), ("C4", 3), ("C5", 3)])
def test_scan_delivers_over_every_rank_s_own_link_into_a_shared_host_segment(config, world, tmp_path, capsys):
    """Round 4: the multi-rank delivery.  `world` ranks in separate processes (on the one GPU this box has), no communicator at
    all: every rank reads its OWN chains back -- C5's while they are sampled, C4's rows group by group behind the post-processing
    -- into its part of one host segment (dist.HostSegment), which rank 0 maps too; a barrier is all that crosses the control
    plane.  Five grid points: ragged over 2 and 3 ranks.  The result must be the single-rank scan's, bit for bit, the exit
    status 0 (no RCCL was asked for), and the accounting must show `world` links with each rank's own bytes."""
    import json
    import socket
    import subprocess
    import sys
    from golemflavor_amd import scan
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scan.main(["--config", config, "--points", "5", "--nwalkers", "32", "--burnin", "6", "--nsteps", "9", "--outfile", str(tmp_path / "one")])
    capsys.readouterr()
    script = tmp_path / "child.py"
    script.write_text(_TWO_RANK_CHILD)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", GF_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(tmp_path / "many"), config], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[1][-1500:]) for p, o in zip(procs, outs)]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["ranks"] == world and line["gather"].startswith("shared host segment") and line["rccl_error"] is None
    st = line["gather_stats"]
    width = 9 if config == "C4" else 12
    assert st["gather_bytes"] == 0 and st["d2h_bytes"] == len(gdist.shard(5, 0, world)) * 9 * 32 * width * 8    # rank 0's own points only
    a, b = np.load(str(tmp_path / "one.npy")), np.load(str(tmp_path / "many.npy"))
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def test_scan_with_datadir_gathers_nothing(tmp_path, capsys):
    """--datadir on two ranks: every rank writes the files of its own grid points, as the reference's jobs do
    (golemflavor/mcmc.py:108-126) -- no chain crosses anything; the line's finite_fraction comes from a two-number reduction.
    The files are the single-rank run's, bit for bit."""
    import json
    import socket
    import subprocess
    import sys
    from golemflavor_amd import scan
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--config", "C4", "--points", "5", "--nwalkers", "32", "--burnin", "6", "--nsteps", "9"]
    scan.main(args + ["--datadir", str(tmp_path / "one")])
    one = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    child = tmp_path / "child.py"
    child.write_text("import sys\nsys.path.insert(0, sys.argv[1])\nfrom golemflavor_amd import scan\nscan.main(sys.argv[2:])\n")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", GF_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(child), root] + args + ["--datadir", str(tmp_path / "two")],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[1][-1500:]) for p, o in zip(procs, outs)]
    line = json.loads([l for l in outs[0][0].strip().splitlines() if l.startswith("{")][-1])
    assert line["gather"].startswith("none") and line["chains_shape"] == [5, 9 * 32, 9] == one["chains_shape"]
    assert line["finite_fraction"] == pytest.approx(one["finite_fraction"])
    f1, f2 = sorted(os.listdir(str(tmp_path / "one"))), sorted(os.listdir(str(tmp_path / "two")))
    assert f1 == f2 and len(f1) == 5
    for f in f1:
        assert np.array_equal(np.load(str(tmp_path / "one" / f)), np.load(str(tmp_path / "two" / f)), equal_nan=True)


def test_scan_rows_to_host_equal_rows_on_device():
    """gf_sampler_postprocess_rows hands the scan's rows to the host group by group while later chains are still being
    post-processed; they must be the rows gf_sampler_postprocess_rows_device assembles (19 chains: ragged last group)."""
    from golemflavor_amd import scan
    pts = scan.texture_grid(6)[:64:3][:19]
    jobs = [scan._TexturePoint(p, g, dimension=6, texture=Texture.OEU, nwalkers=32, device=0) for g, p in enumerate(pts)]
    s = mcmc_utils.DeviceEnsembleSampler(32, 6, [j.f for j in jobs], seed=4)
    s.run_mcmc(np.stack([j.p0 for j in jobs]), 9)
    models = [j.post_model for j in jobs]
    host = s.postprocess_rows(models=models)
    m0 = jobs[0].f.model
    d_rows = m0.alloc(host.nbytes)
    s.postprocess_rows_to_device(d_rows.ptr, models=models)
    dev = d_rows.download(host.shape)
    d_rows.free()
    assert host.shape == (19, 9 * 32, 9) and np.array_equal(host, dev, equal_nan=True)
    assert np.isnan(host[:, :, 0]).any() and np.isfinite(host[:, :, 0]).any()      # the grid reaches the failing region
    s.close()
    for j in jobs:
        j.close()


def test_grid_point_chain_does_not_depend_on_the_sharding():
    """Every chain of a stacked scan draws from the random stream of its GLOBAL grid index (gf_sampler_set_stream_ids):
    rank 1's two grid points of world 2 (`dist.shard`: points 1 and 2 under the skewed round robin) give bitwise the same chains
    whether their rank holds all four points (world 1) or just those two, and so does the one-sampler-per-point path's keying by
    grid index."""
    from golemflavor_amd import scan
    pts = scan.sens_grid(n_scales=2, n_sources=1)[:4]
    make = lambda p, g: scan._SensPoint(p, g, nwalkers=32, device=0)   # noqa: E731
    full = scan.run_points(pts, [0, 1, 2, 3], make, 5, 12)
    mine = gdist.shard(4, 1, 2)
    half = scan.run_points(pts, mine, make, 5, 12)
    assert sorted(half) == mine == [1, 2]
    for g in mine:
        assert np.array_equal(full[g], half[g])
    assert not np.array_equal(full[0][:, :4], full[1][:, :4])          # distinct streams per grid point


def test_rccl_error_text_reaches_python():
    """GF_ERR_COMM carries the failing RCCL call's text (gf_comm_last_error is bound)."""
    import ctypes as C
    L = _lib.lib()
    h = C.c_void_p()
    bad = (C.c_uint8 * _lib.GF_COMM_ID_BYTES)()
    assert L.gf_comm_create(bad, 3, 2, 0, C.byref(h)) == _lib.GF_ERR_INVALID_ARG       # rank outside the world
    info = gdist.rccl_library_info()
    assert "librccl" in info and int(info.split()[0]) > 20000


def test_integration_md_stub_is_executable(golden, oracle):
    """INTEGRATION.md section 2 is the ctypes stub a GolemFlavor maintainer would add.  Run exactly that text
    (with this package's Param/enums/fr standing in for the reference's, same names and values) against the oracle."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# golemflavor/hip.py.*?)```", text, re.S).group(1)
    code = code.replace("from golemflavor.", "from golemflavor_amd.").replace("from golemflavor import", "from golemflavor_amd import")
    code = code.replace('C.CDLL("libgolemhip.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    asimov, ps = notebook_sets(golden)
    f = ns["HipLnProb"](asimov, ps)
    th = np.ascontiguousarray(golden["g6_theta"][:512])
    got = f(th)
    om = oracle.make_model(ps, "SM_GAUSS", bestfit_fr=golden["g6_bestfit_fr"], smearing=0.02)
    ref = oracle.lnprob_batch(om, th)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-10 * np.abs(ref[fin]).max()
    assert isinstance(f(th[0]), float) and f(th[0]) == got[0]
    with pytest.raises(AssertionError):
        f(np.zeros(5))


def test_reference_named_entry_points(golden):
    """llh.ln_prob / llh.triangle_llh / llh.lnprior / fr.flux_averaged_BSMu with the reference's names and
    signatures (llh.py:65-130, fr.py:403-458), bound with functools.partial exactly as scripts/fr.py:182-187 does,
    against the 12-dim golden rows."""
    from functools import partial
    from common import TEX_BY_VALUE
    from golemflavor_amd import fr as fr_utils
    rows = golden["g9_rows"]
    key = rows[0, :5]
    sel = np.all(rows[:, :5] == key, axis=1) & (golden["g9_status"] == 0)
    dim, tex, src = int(key[0]), TEX_BY_VALUE[int(key[1])], key[2:5]
    inj = fr_utils.fr_to_angles(golden["g9_injected"])
    asimov, ps = Cf.fr_paramsets(dim, inj)
    args = bsm_args(dim, tex, src)
    th = np.ascontiguousarray(rows[sel][:, 5:])
    ref = golden["g9_lnprob"][sel]
    exact = golden["g9_fr_exact"][sel]
    # rows the device flags although the stored reference run passed (within a factor of order one of its 1e-7
    # threshold and a different last bit of 10**logLam, DESIGN.md "Unitarity status") would raise here exactly as a
    # reference failure does: leave them out
    m = llh_utils._bound(args, asimov, ps).model
    keep = (m.lnprob(th)[1] != _lib.GF_ST_NON_UNITARY) & (m.propagate(th)[1] != _lib.GF_ST_NON_UNITARY)   # propagate has no box
    th, ref, exact = th[keep], ref[keep], exact[keep]
    assert keep.mean() > 0.7
    f = partial(llh_utils.ln_prob, args=args, asimov_paramset=asimov, llh_paramset=ps)
    got = f(th)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin) and fin.sum() > 5
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-10 * np.abs(ref[fin]).max()
    one = f(th[0])
    assert isinstance(one, float) and (one == got[0] or (np.isinf(one) and np.isinf(got[0])))
    with pytest.raises(AssertionError):
        f(th[0][:5])
    # ln_prob = lnprior + triangle_llh (llh.py:124-130)
    lp = llh_utils.lnprior(th, ps)
    tl = llh_utils.triangle_llh(th, args, asimov, ps)
    both = fin & np.isfinite(tl)
    assert both.sum() > 5 and np.abs((lp + tl)[both] - got[both]).max() <= 1e-9 * np.abs(got[both]).max()
    # flux_averaged_BSMu: the composition behind it (fr.py:403-458), exact values from the 60-digit golden
    frs = fr_utils.flux_averaged_BSMu(th, args, 2.0, ps)
    has = np.isfinite(exact[:, 0])
    assert np.abs(frs[has] - exact[has]).max() <= 1e-11
    assert fr_utils.flux_averaged_BSMu(th[0], args, 2.0, ps).shape == (3,)
    # the bound state is compiled once per distinct posterior
    n0 = len(llh_utils._BOUND)
    f(th[:3])
    assert len(llh_utils._BOUND) == n0


def test_tutorial_chain_reproduces_stored_notebook_outputs(golden, capsys):
    """examples/tutorial.ipynb:611-619 stores the outputs of its chain (60 walkers, 1000 + 10000 steps): sum of
    acceptance fractions 42.9055 (mean 0.715) and autocorrelation times 30-31.  emcee is un-vendored and unseeded
    there, so the check is distributional."""
    asimov, ps = Cf.tutorial_paramsets(golden["g10_asimov_angles"])
    f = llh_utils.tutorial_ln_prob(asimov, ps)
    np.random.seed(26)
    p0 = mcmc_utils.flat_seed(ps, nwalkers=60)
    samples = mcmc_utils.mcmc(p0=p0, ln_prob=f, ndim=2, nwalkers=60, burnin=1000, nsteps=10000, seed=4)
    out = capsys.readouterr().out
    assert samples.shape == (600000, 2)
    acc = float(out.split("sum of acceptance fraction")[1].split()[0])
    assert 41.0 < acc < 45.0                                   # notebook: 42.9055
    tau = np.array(out.split("autocorrelation")[1].replace("[", " ").replace("]", " ").split()[:2], dtype=float)
    assert np.all(tau > 22) and np.all(tau < 40)               # notebook: [30.39 30.96]
    # the posterior sits on the injected flavor angles
    assert np.allclose(samples.mean(axis=0), golden["g10_asimov_angles"], atol=0.01)
    f.close()


def test_no_device_memory_leak_over_model_and_sampler_lifecycles(golden):
    """A scan creates and destroys hundreds of models and samplers: device memory must come back (the per-device
    pool keeps a stream and a 3-KB constant block per retired model, nothing else)."""
    hip = C.CDLL("libamdhip64.so")
    free0, free1, total = C.c_size_t(), C.c_size_t(), C.c_size_t()
    asimov, ps = notebook_sets(golden)
    rng = np.random.default_rng(0)
    p0 = rng.uniform(*np.array(ps.seeds, dtype=float).T, size=(2, 64, 6))

    def cycle(k):
        f = llh_utils.notebook_ln_prob(asimov, ps)
        f(p0[0])                                              # host-buffer path: staging buffers
        s = mcmc_utils.DeviceEnsembleSampler(64, 6, f, nchains=2, seed=k)
        s.run_mcmc(p0, 40, thin=2)
        s.postprocess(want_fr=True, nbins=8)
        s.walker_mean()
        s.close()
        s2 = mcmc_utils.DeviceEnsembleSampler(64, 6, [f, f], seed=k)   # per-chain posteriors
        s2.run_mcmc(p0, 20)
        s2.close()
        f.close()

    for k in range(5):                                        # warm the pool and the runtime's own caches
        cycle(k)
    assert hip.hipMemGetInfo(C.byref(free0), C.byref(total)) == 0
    for k in range(150):
        cycle(100 + k)
    assert hip.hipMemGetInfo(C.byref(free1), C.byref(total)) == 0
    leaked = int(free0.value) - int(free1.value)
    assert leaked < 32 << 20, "device memory shrank by %.1f MiB over 150 lifecycles" % (leaked / 2 ** 20)


def test_handles_are_independent_across_threads(golden):
    """"Thread-safe per handle": different models / samplers driven from different host threads at the same time
    (ctypes drops the GIL inside every call) give what they give alone."""
    import threading
    asimov, ps = notebook_sets(golden)
    from golemflavor_amd.descriptor import compile_model
    from golemflavor_amd.model import Model
    th = np.ascontiguousarray(golden["g6_theta"][:3000])
    bfs = [(1 / 3, 1 / 3, 1 / 3), (0.2, 0.45, 0.35), (0.5, 0.3, 0.2), (0.55, 0.18, 0.27)]
    models = [Model(compile_model(ps, "SM_GAUSS", bestfit_fr=bf, smearing=0.05)) for bf in bfs]
    want = [m.lnprob(th, want_status=False) for m in models]
    rng = np.random.default_rng(1)
    p0 = rng.uniform(*np.array(ps.seeds, dtype=float).T, size=(64, 6))

    def chain_of(m, k):
        s = mcmc_utils.DeviceEnsembleSampler(64, 6, m, seed=10 + k)
        s.run_mcmc(p0, 60)
        c = s.chain
        s.close()
        return c
    want_chain = [chain_of(m, k) for k, m in enumerate(models)]
    errors = []

    def work(k):
        try:
            for rep in range(30):
                n = 64 * (1 + (rep * 7 + k) % 40) + (rep % 3)
                got = models[k].lnprob(th[:n], want_status=False)
                assert np.array_equal(got, want[k][:n], equal_nan=True)
            assert np.array_equal(chain_of(models[k], k), want_chain[k])
        except Exception as exc:       # noqa: BLE001
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(models))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for m in models:
        m.close()


def test_bsm_samplers_capture_graphs_concurrently():
    """Two BSM samplers (per-half-step launches replayed from hipGraphs, thread-local capture) advanced from two
    host threads at once reproduce their single-threaded chains."""
    import threading
    asimov, ps = Cf.fr_paramsets(6, (0.4444444444444444, 0.0))
    fs = [llh_utils.bsm_ln_prob(bsm_args(6, tex, (0., 1., 0.)), asimov, ps, smearing=0.3, on_nonunitary="-inf")
          for tex in (Texture.OET, Texture.OUT)]
    rng = np.random.default_rng(3)
    box = np.array(ps.seeds, dtype=float)
    p0 = rng.uniform(box[:, 0], box[:, 1], size=(64, 12))
    p0[:, 11] = rng.uniform(-52, -44, 64)

    def chain_of(f, k):
        s = mcmc_utils.DeviceEnsembleSampler(64, 12, f, seed=k)
        s.on_nonunitary = "-inf"
        s.run_mcmc(p0, 70)                                    # 64 steps from the graph + 6 eager
        c = s.chain
        s.close()
        return c
    want = [chain_of(f, k) for k, f in enumerate(fs)]
    got, errors = [None, None], []

    def work(k):
        try:
            for _ in range(3):
                got[k] = chain_of(fs[k], k)
        except Exception as exc:       # noqa: BLE001
            errors.append((k, repr(exc)))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for k in range(2):
        assert np.array_equal(got[k], want[k])
    for f in fs:
        f.close()


def test_graph_capture_survives_allocation_traffic_from_another_thread(golden):
    """While one thread's BSM sampler captures and replays graphs, another thread creates and destroys models and
    grows its staging buffers (hipMalloc / hipFree / hipHostMalloc / hipHostFree): neither may disturb the other."""
    import threading
    asimov12, ps12 = Cf.fr_paramsets(6, (0.4444444444444444, 0.0))
    f = llh_utils.bsm_ln_prob(bsm_args(6, Texture.OET, (0., 1., 0.)), asimov12, ps12, smearing=0.3, on_nonunitary="-inf")
    rng = np.random.default_rng(3)
    box = np.array(ps12.seeds, dtype=float)
    p0 = rng.uniform(box[:, 0], box[:, 1], size=(64, 12))
    p0[:, 11] = rng.uniform(-52, -44, 64)

    def chain():
        s = mcmc_utils.DeviceEnsembleSampler(64, 12, f, seed=1)
        s.on_nonunitary = "-inf"
        s.run_mcmc(p0, 70)
        c = s.chain
        s.close()
        return c
    want = chain()
    asimov, ps = notebook_sets(golden)
    th = np.ascontiguousarray(golden["g6_theta"][:5000])
    ref = golden["g6_lnprob"][:5000]
    errors, stop = [], threading.Event()

    def sampler_thread():
        try:
            for _ in range(6):
                assert np.array_equal(chain(), want)
        except Exception as exc:       # noqa: BLE001
            errors.append(("sampler", repr(exc)))
        finally:
            stop.set()

    def churn_thread():
        try:
            k = 0
            while not stop.is_set() and k < 400:
                g = llh_utils.notebook_ln_prob(asimov, ps)
                for n in (10, 3000, 100, 5000):                 # staging grows twice per model
                    got = g.model.lnprob(th[:n], want_status=False)
                    fin = np.isfinite(ref[:n])
                    assert np.array_equal(np.isfinite(got), fin) and np.allclose(got[fin], ref[:n][fin], rtol=1e-10)
                g.close()
                k += 1
        except Exception as exc:       # noqa: BLE001
            errors.append(("churn", repr(exc)))

    ts = [threading.Thread(target=sampler_thread), threading.Thread(target=churn_thread)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=180)
    assert not errors, errors
    f.close()
