"""CPU: the pieces of bench.py / scan.py / dist.py that put the N-rank line together, without a GPU.

* bench.py's line assembly and cross-rank reductions, fed with two ranks' synthetic measurements over the socket control
  plane (there is no CPU evaluation path, so bench.py itself cannot run here);
* scan.DeviceGather's packing and indexing with a fake RCCL object, ragged grids included;
* dist.open_rccl with a failure injected on rank 0 (and on another rank): the control plane stays in step.
"""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from golemflavor_amd import dist as gdist  # noqa: E402
from golemflavor_amd import scan  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_BENCH_CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
from golemflavor_amd import dist as gdist
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
control = gdist.SocketBackend(rank, world, addr="127.0.0.1", port=port, token="bench-line", timeout=60)
# this rank's synthetic measurements: rank 1 is the slow one
elapsed, kernel_ms = bench.reduce_step_timing(control, 0.0034 + 0.0002 * rank, 0.170 + 0.004 * rank)
local = {"setup": 0.010 + 0.001 * rank, "sampling": 0.30 - 0.02 * rank, "pack_s": 0.05, "xgmi_s": 0.020 if rank == 0 else 0.001,
         "d2h_s": 0.19 if rank == 0 else 0.0, "gather_bytes": 4718592000 if rank == 0 else 0, "block_bytes": 4718592000,
         "d2h_bytes": 9437184000 if rank == 0 else 0}
phases = bench.reduce_phases(control, local)
seconds = float(control.allreduce_max([0.61 + 0.01 * rank])[0])
if rank == 0:
    rec = bench.scan_record_from_phases("C4", world, 64, 2048, bench.REF_BURNIN, bench.REF_NSTEPS,
                                        64 * (2048 * 1200 + 2048 * 1000), seconds, phases,
                                        gather_kind="rccl gather to rank 0 over xGMI (gf_comm_gather), one download",
                                        rccl_init_s=1.2, chain_bytes_to_host=9437184000, finite_fraction=1.0)
    cpu = {"value": 1.0e7, "unit": "evals/s", "cores": 16, "kind": "port", "sample": "synthetic"}
    line = bench.assemble_line(world=world, steps=20, warmup=5, walkers=4096, ensembles=4096, elapsed=elapsed, kernel_ms=kernel_ms,
                               control_plane="tcp sockets", librccl="22707 /opt/rocm/lib/librccl.so", gathered_ok=True,
                               rccl_init_s=1.2, extras={"c4_scan_ref": rec}, cpu=cpu, parity=7e-16)
    print(json.dumps(line))
control.barrier()
control.close()
"""


def test_bench_line_from_two_ranks_over_the_socket_control_plane(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(_BENCH_CHILD)
    port = _free_port()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-3000:]
        outs.append(so)
    assert outs[1].strip() == ""                                   # only rank 0 prints
    lines = [l for l in outs[0].splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    n = 4096 * 4096
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["ms_per_step"] == pytest.approx(1e3 * 0.0036 / 20)                       # the MAX over the two ranks
    assert d["value"] == pytest.approx(2 * n * 20 / 0.0036)                           # whole job: both ranks' evaluations
    r = d["roofline"]
    assert r["kernel_ms"] == pytest.approx(0.174) and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert r["achieved"] == pytest.approx(56 * n / 0.174e-3 / 1e9) and r["frac"] == pytest.approx(r["achieved"] / 8000.0)
    assert d["cpu_baseline"]["kind"] == "port" and d["gpu_over_cpu"] == pytest.approx(d["value"] / 1.0e7)
    s = d["c4_scan_ref"]
    assert s["ranks"] == 2 and s["burnin"] == 200 and s["nsteps"] == 1000 and s["scaling"].startswith("strong")
    assert s["seconds"] == pytest.approx(0.62) and s["sampling_s"] == pytest.approx(0.30) and s["setup_s"] == pytest.approx(0.011)
    assert s["gather_bytes"] == 4718592000 and s["xgmi_s"] == pytest.approx(0.020)
    assert s["gather_GBps"] == pytest.approx(4718592000 / 0.020 / 1e9) and s["d2h_s"] == pytest.approx(0.19)
    assert s["evals_per_s"] == pytest.approx(s["evals"] / 0.62) and s["rccl_init_s"] == 1.2


# ---------------------------------------------------------------------------------------------------------------
class _FakeBuf:
    def __init__(self, nbytes):
        assert nbytes % 8 == 0
        self.arr = np.full(nbytes // 8, np.nan)
        self.ptr = self.arr                                       # what the product passes on as the "device pointer"
        self.nbytes = nbytes

    def download(self, shape, dtype=np.float64, out=None):
        n = int(np.prod(shape))
        if out is None:
            return self.arr[:n].reshape(shape).copy()
        assert out.shape == tuple(shape)
        out[...] = self.arr[:n].reshape(shape)
        return out

    def free(self):
        self.arr = None


class _FakeModel:
    def alloc(self, nbytes):
        return _FakeBuf(nbytes)


class _FakeRccl:
    """gather_device among threads: every rank deposits its block, the root assembles them in rank order."""

    def __init__(self, rank, world, shared, barrier):
        self.rank, self.world, self.shared, self.barrier = rank, world, shared, barrier

    def gather_device(self, d_send, d_recv, nbytes, root=0):
        assert (d_recv is not None) == (self.rank == root)         # only the root holds world x the block
        self.shared[self.rank] = d_send[:nbytes // 8].copy()
        self.barrier.wait()
        if self.rank == root:
            assert d_recv.nbytes == nbytes * self.world
            for r in range(self.world):
                d_recv[r * (nbytes // 8):(r + 1) * (nbytes // 8)] = self.shared[r]
        self.barrier.wait()


class _FakeSampler:
    """Stands in for mcmc.DeviceEnsembleSampler: the chain of grid point g is filled with g + sample index / 1e6."""

    def __init__(self, order, nstored, nwalkers, ndim, with_rows):
        self.order, self.nstored, self.nw, self.ndim, self.with_rows = order, nstored, nwalkers, ndim, with_rows

    def _fill(self, d, width):
        per = self.nstored * self.nw
        for i, g in enumerate(self.order):
            blk = np.arange(per * width, dtype=np.float64).reshape(per, width) / 1e6 + g
            d[i * per * width:(i + 1) * per * width] = blk.ravel()

    def chain_to_device(self, d):
        assert not self.with_rows
        self._fill(d, self.ndim)

    def postprocess_rows_to_device(self, d, models=None):
        assert self.with_rows and len(models) == len(self.order)
        self._fill(d, 3 + self.ndim)


class _Job:
    def __init__(self, nwalkers, ndim, post):
        self.nwalkers, self.ndim, self.post_model = nwalkers, ndim, post


@pytest.mark.parametrize("n_points,world,with_rows", [(5, 2, False), (7, 3, True), (4, 3, False), (8, 2, True), (3, 3, False), (9, 4, True)])
def test_device_gather_packing_and_indexing_with_a_fake_rccl(n_points, world, with_rows, monkeypatch):
    """Grid point g must come back from slot [owner(g), g div world] whatever the grid / world sizes: grids that do not
    divide evenly, ranks with fewer points than slots.  (A rank with NO point never reaches the gather: scan.main and
    bench.py use the device gather only when there are at least as many grid points as ranks.)"""
    monkeypatch.setenv("GF_SCAN_PREFAULT", "1")
    nstored, nw, ndim = 3, 4, 2
    shared, barrier = [None] * world, threading.Barrier(world)
    results, stats, errors = [None] * world, [None] * world, []

    def run(rank):
        try:
            order = gdist.shard(n_points, rank, world)
            jobs = {g: _Job(nw, ndim, object() if with_rows else None) for g in order}
            g = scan.DeviceGather(_FakeRccl(rank, world, shared, barrier), rank, world, _FakeModel())
            if n_points % 2:                                       # with and without the result array allocated ahead of the run
                g.prepare(jobs[order[0]], len(order), n_points, nstored)     # (opt-in: GF_SCAN_PREFAULT, set by the test below)
                assert g._dest is not None
            results[rank] = g.run(_FakeSampler(order, nstored, nw, ndim, with_rows), jobs, order, n_points)
            stats[rank] = dict(g.stats)
        except Exception as exc:           # noqa: BLE001
            errors.append(exc)
            barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=30)
    assert not errors, errors
    width = 3 + ndim if with_rows else ndim
    per = nstored * nw
    assert all(results[r] is None for r in range(1, world))        # only the root receives
    out = results[0]
    assert len(out) == n_points
    for g in range(n_points):
        want = np.arange(per * width, dtype=np.float64).reshape(per, width) / 1e6 + g
        assert np.array_equal(out[g], want), g
    slots = gdist.slots_per_rank(n_points, world)
    blk = slots * per * width * 8
    assert stats[0]["block_bytes"] == blk
    # bytes that crossed into the root are counted ONCE, on the root (round 3 counted them on every rank, and the line, which
    # sums `*_bytes` over the ranks, quoted the xGMI rate `world` times too high)
    assert stats[0]["gather_bytes"] == blk * (world - 1) and all(stats[r]["gather_bytes"] == 0 for r in range(1, world))
    assert stats[0]["d2h_bytes"] == blk * world and all(stats[r]["d2h_bytes"] == 0 for r in range(1, world))
    # ... and the reduction bench.py applies to what the producer REALLY holds on every rank gives the true crossing
    import bench
    reduced = [None] * world
    bar2, slots2 = threading.Barrier(world), [None] * world

    class _Ctl:
        def __init__(self, r):
            self.rank, self.world = r, world

        def allgather_bytes(self, payload):
            slots2[self.rank] = bytes(payload)
            bar2.wait()
            out = list(slots2)
            bar2.wait()
            return out

    def red(r):
        local = {k: v for k, v in stats[r].items() if isinstance(v, (int, float)) and k not in ("ranks", "slots_per_rank")}
        reduced[r] = bench.reduce_phases(_Ctl(r), local)

    th = [threading.Thread(target=red, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=30)
    for r in range(world):
        assert reduced[r]["gather_bytes"] == blk * (world - 1), reduced[r]
        assert reduced[r]["d2h_bytes"] == blk * world
    rec = bench.scan_record_from_phases("C5", world, n_points, nw, 1, nstored, 1, 1.0, dict(reduced[0], xgmi_s=0.5), gather_kind="fake")
    assert rec["gather_bytes"] == blk * (world - 1) and rec["gather_GBps"] == pytest.approx(blk * (world - 1) / 0.5 / 1e9)


# ---------------------------------------------------------------------------------------------------------------
_RCCL_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from golemflavor_amd import dist as gdist
from golemflavor_amd._lib import GfModelDesc
rank, world, port, mode = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
b = gdist.SocketBackend(rank, world, addr="127.0.0.1", port=port, token="rccl-inject", timeout=60)

class FakeComm:
    def __init__(self, uid): self.uid = uid
    def close(self): pass

def make_id():
    if mode == "rank0_id_fails":
        raise RuntimeError("ncclGetUniqueId: unhandled system error (injected)")
    if mode == "rank0_id_hangs":
        import time; time.sleep(3600)
    if mode == "rank0_short_id":
        return b"x" * 17
    return bytes(range(128))

def factory(r, w, d, uid):
    assert len(uid) == 128 and uid == bytes(range(128))
    if mode == "rank1_init_fails" and r == 1:
        raise RuntimeError("ncclCommInitRank: invalid usage (injected)")
    return FakeComm(uid)

tmo = 1.0 if mode == "rank0_id_hangs" else 20.0
comm, err, stuck = gdist.open_rccl(rank, world, 0, b, timeout=tmo, make_id=make_id, backend_factory=factory)
out = {"rank": rank, "comm": comm is not None, "err": err, "stuck": stuck}
# the collectives that follow in bench.py / scan.py must still line up on every rank
d = GfModelDesc(); d.ndim = 7 if rank == 0 else 0; d.smearing = 0.02 if rank == 0 else 0.0
descs = gdist.broadcast_descriptors([d, d, d] if rank == 0 else [], b)
out["descs"] = [(x.ndim, x.smearing) for x in descs]
n = 5
local = {g: np.full((2, 3), float(g)) for g in gdist.shard(n, rank, world)}
out["chains"] = [float(c[0, 0]) for c in gdist.gather_chains(local, n, b)]
root = gdist.gather_chains_to_root(local, n, b)
out["root"] = None if root is None else [float(c[1, 2]) for c in root]
out["max"] = b.allreduce_max([float(rank)]).tolist()
b.barrier()
b.close()
print(json.dumps(out))
sys.stdout.flush()
os._exit(0)          # a deliberately hung helper thread must not hold the exit
"""


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["ok", "rank0_id_fails", "rank0_id_hangs", "rank0_short_id", "rank1_init_fails"])
def test_open_rccl_failures_leave_the_control_plane_in_step(world, mode, tmp_path):
    """A rank-0 failure (or hang) before the id exchange used to desynchronise the sockets: peers read error strings as
    the unique id and later collectives parsed garbage.  Every rank must now agree on the outcome and the descriptor
    broadcast / chain gathers that follow must still work."""
    script = tmp_path / "child.py"
    script.write_text(_RCCL_CHILD)
    port = _free_port()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(port), mode], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    for r, o in enumerate(outs):
        assert o["rank"] == r
        assert o["comm"] == (mode == "ok")
        assert o["stuck"] is False
        if mode == "ok":
            assert o["err"] is None
        elif mode == "rank0_id_fails":
            assert "ncclGetUniqueId" in o["err"]
        elif mode == "rank0_id_hangs":
            assert "timeout" in o["err"]
        elif mode == "rank0_short_id":
            assert "17 bytes" in o["err"]
        else:
            assert "ncclCommInitRank" in o["err"]
        assert o["descs"] == [[7, 0.02]] * 3
        assert o["chains"] == [0.0, 1.0, 2.0, 3.0, 4.0]
        assert o["root"] == ([0.0, 1.0, 2.0, 3.0, 4.0] if r == 0 else None)
        assert o["max"] == [world - 1.0]


def test_rccl_backend_refuses_an_id_of_the_wrong_length():
    with pytest.raises(ValueError):
        gdist.RcclBackend(0, 1, 0, uid=b"short")


# ---------------------------------------------------------------------------------------------------------------
# round 4: delivery over N links -- the shared host segment, with real processes over the socket control plane
_SEGMENT_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from golemflavor_amd import dist as gdist
from golemflavor_amd import scan
rank, world, port, n_points, mode = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
ctl = gdist.SocketBackend(rank, world, addr="127.0.0.1", port=port, token="segment", timeout=60)
nstored, nw, ndim = 3, 4, 2
with_rows = mode.startswith("rows")

class Job:
    def __init__(self): self.nwalkers, self.ndim, self.post_model = nw, ndim, (object() if with_rows else None)

class Sampler:
    # stands in for mcmc.DeviceEnsembleSampler: grid point g's block is g + sample index / 1e6 (NaN in one place: bit patterns travel)
    def __init__(self, order): self.order, self.nstored = order, nstored
    def _blk(self, g, width):
        b = np.arange(nstored * nw * width, dtype=np.float64).reshape(nstored * nw, width) / 1e6 + g
        b[1, 0] = np.nan
        return b
    def postprocess_rows(self, models=None, out=None):
        assert with_rows and len(models) == len(self.order)
        for i, g in enumerate(self.order): out[i] = self._blk(g, 3 + ndim)
        return out
    def chain_to_host(self, out):
        for i, g in enumerate(self.order): out[i] = self._blk(g, ndim).reshape(out.shape[1:])
        return out

order = gdist.shard(n_points, rank, world)
jobs = {g: Job() for g in order}
if mode.endswith("nosegment"):
    os.environ["GF_SEGMENT_DIR"] = "/nonexistent-dir"        # the file backing is refused ...
    real = os.memfd_create
    def broken(*a, **k): raise OSError("memfd_create refused (injected)")
    os.memfd_create = broken                                   # ... and so is the memfd: every rank must fall back together
g = scan.SharedHostGather(ctl, rank, world)
g.prepare(jobs[order[0]], len(order), n_points, nstored)
os.environ["GF_SCAN_NO_STREAMED_CHAIN"] = "1"
assert not g.streams_chain(jobs[order[0]])
out = g.run(Sampler(order), jobs, order, n_points)
res = {"rank": rank, "stats": {k: v for k, v in g.stats.items() if isinstance(v, (int, float, str))}, "got": None}
if out is not None:
    width = 3 + ndim if with_rows else ndim
    ok = len(out) == n_points
    for gp in range(n_points):
        want = np.arange(nstored * nw * width, dtype=np.float64).reshape(nstored * nw, width) / 1e6 + gp
        want[1, 0] = np.nan
        ok = ok and np.array_equal(np.asarray(out[gp]).reshape(want.shape).view(np.uint64), want.view(np.uint64))
    res["got"] = bool(ok)
    res["leftover"] = [f for f in os.listdir("/dev/shm") if f.startswith("gf_segment_%d_" % os.getpid())]
g.release()
ctl.barrier()
ctl.close()
print(json.dumps(res))
"""


def _spawn(script_text, tmp_path, world, *args, env=None, timeout=120):
    script = tmp_path / "child.py"
    script.write_text(script_text)
    port = _free_port()
    e = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    e.update(env or {})
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), str(port)] + [str(x) for x in args],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e) for r in range(world)]
    return [(p,) + p.communicate(timeout=timeout) for p in procs]


@pytest.mark.parametrize("world,n_points,mode", [(2, 5, "chain"), (3, 7, "rows"), (3, 3, "chain"), (2, 4, "rows-nosegment"), (3, 5, "chain-nosegment")])
def test_shared_host_segment_delivers_every_rank_s_points_to_rank_0(world, n_points, mode, tmp_path):
    """scan.SharedHostGather over real processes: every rank writes its own grid points into its part of ONE host segment
    (dist.HostSegment: a file under /dev/shm whose name is gone once everybody has mapped it), rank 0 reads the whole grid
    out of the same pages -- bit patterns included, ragged grids included -- and nothing but a barrier crosses the control
    plane.  `-nosegment`: the segment cannot be created; all ranks agree on that inside the constructor and the blocks go
    to rank 0 over the control plane instead."""
    outs = _spawn(_SEGMENT_CHILD, tmp_path, world, n_points, mode)
    recs = []
    for p, so, se in outs:
        assert p.returncode == 0, se[-3000:]
        recs.append(json.loads(so.strip().splitlines()[-1]))
    assert recs[0]["got"] is True and all(r["got"] is None for r in recs[1:])
    per = 3 * 4 * ((3 + 2) if mode.startswith("rows") else 2) * 8
    for r, rec in enumerate(recs):
        st = rec["stats"]
        assert st["d2h_bytes"] == len(gdist.shard(n_points, r, world)) * per           # every rank counts its OWN link's bytes
        assert st["gather_bytes"] == 0 and st["xgmi_s"] == 0.0
        if mode.endswith("nosegment"):
            assert st["delivery"].startswith("private read-back") and "injected" in st["delivery"]
        else:
            assert st["delivery"].startswith("shared host segment (file under /dev/shm)")
    assert recs[0].get("leftover") == []                                                # no name left behind
    assert sum(r["stats"]["d2h_bytes"] for r in recs) == n_points * per                 # what bench.reduce_phases sums: the result, once


_TIMEOUT_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
from golemflavor_amd import dist as gdist
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctl = gdist.SocketBackend(rank, world, addr="127.0.0.1", port=port, token="timeout", timeout=60)
ctl.barrier()
out = {"rank": rank, "error": None}
try:
    if rank == 2:
        time.sleep(30)                     # alive, socket open, not answering: the case a dead-peer check cannot see
    else:
        ctl.allreduce_max([1.0])
except Exception as exc:
    out["error"] = "%s: %s" % (type(exc).__name__, exc)
print(json.dumps(out), flush=True)
os._exit(4 if out["error"] else 0)
"""


def test_a_silent_peer_becomes_an_error_not_a_hang(tmp_path):
    """Three ranks, rank 2 stops answering in the middle of a collective (it neither dies nor closes its socket).  With
    GF_CONTROL_TIMEOUT = 2 s rank 0 gives up on it (ControlPlaneTimeout) and leaves; rank 1, waiting for rank 0's relay, sees the
    connection go.  Both exit non-zero with the reason in their line, long before the silent rank wakes up -- bench.py wraps
    the same exceptions into its JSON line (`error`) and exit status 4."""
    import time
    t0 = time.time()
    outs = _spawn(_TIMEOUT_CHILD, tmp_path, 3, env={"GF_CONTROL_TIMEOUT": "2"}, timeout=60)
    recs = []
    for p, so, se in outs[:2]:
        assert p.returncode == 4, (p.returncode, se[-2000:])
        recs.append(json.loads(so.strip().splitlines()[-1]))
    assert "ControlPlaneTimeout" in recs[0]["error"] and "did not answer within 2 s" in recs[0]["error"]
    assert recs[1]["error"] is not None and ("ConnectionError" in recs[1]["error"] or "Timeout" in recs[1]["error"])
    assert time.time() - t0 < 45
    assert gdist.SocketBackend.OP_TIMEOUT < 600                    # the default stays under the driver's limit for a bench run


_STUCK_CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from golemflavor_amd import dist as gdist
rank, world, port, mode = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
b = gdist.SocketBackend(rank, world, addr="127.0.0.1", port=port, token="stuck", timeout=60)

class FakeComm:
    kind = "rccl"
    def close(self): pass

class FakeIpc:
    kind = "hipIpc"
    def __init__(self, r, w, d, c): self.control = c
    def probe(self):
        errs = self.control.allgather_bytes(b"no peer access (injected)" if (mode == "probe_fails" and rank == 1) else b"")
        return next((e.decode() for e in errs if e), None)

def factory(r, w, d, uid):
    if mode == "ok":
        return FakeComm()
    if r == world - 1 or mode in ("all_fail", "probe_fails"):
        raise RuntimeError("ncclCommInitRank: invalid usage (injected)")      # one rank's init errors out at once ...
    time.sleep(3600)                                                          # ... its peers sit in the bootstrap until the timeout

comm, err, stuck = gdist.open_device_gather(rank, world, 0, b, timeout=1.5, make_id=lambda: bytes(range(128)),
                                            backend_factory=factory, ipc_factory=FakeIpc)
out = {"rank": rank, "kind": getattr(comm, "kind", None), "err": err, "stuck": stuck}
# the collectives that follow in bench.py / scan.py must still line up on every rank
out["max"] = b.allreduce_max([float(rank)]).tolist()
n = 5
local = {g: np.full((2, 3), float(g)) for g in gdist.shard(n, rank, world)}
root = gdist.gather_chains_to_root(local, n, b)
out["root"] = None if root is None else [float(c[1, 2]) for c in root]
b.barrier()
b.close()
print(json.dumps(out), flush=True)
os._exit(0)
"""


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["ok", "one_fails_others_hang", "all_fail", "probe_fails"])
def test_open_device_gather_decides_the_fallback_from_shared_state(world, mode, tmp_path):
    """The usual RCCL failure: one rank's ncclCommInitRank errors out, its peers time out in the bootstrap -- `stuck` differs
    between the ranks.  Round 3 let only the non-stuck rank enter `same_node` (an all-gather) and the control plane was off by
    one collective from then on.  Now the ranks exchange their stuck flags first: anybody stuck -> nobody tries hipIpc; nobody
    stuck -> everybody takes same_node and the probe.  Either way the collectives that follow line up."""
    outs = _spawn(_STUCK_CHILD, tmp_path, world, mode)
    recs = []
    for p, so, se in outs:
        assert p.returncode == 0, se[-3000:]
        recs.append(json.loads(so.strip().splitlines()[-1]))
    for r, o in enumerate(recs):
        assert o["max"] == [world - 1.0] and o["root"] == ([0.0, 1.0, 2.0, 3.0, 4.0] if r == 0 else None)
        if mode == "ok":
            assert o["kind"] == "rccl" and o["err"] is None
        elif mode == "one_fails_others_hang":
            assert o["kind"] is None and o["err"] is not None            # somebody is stuck: no hipIpc anywhere
            assert o["stuck"] == (r != world - 1)
        elif mode == "all_fail":
            assert o["kind"] == "hipIpc" and "ncclCommInitRank" in o["err"] and o["stuck"] is False
        else:
            assert o["kind"] is None and "hipIpc probe: no peer access" in o["err"]


def test_node_identity_is_the_boot_id():
    ident = gdist.node_identity()
    boot = open("/proc/sys/kernel/random/boot_id").read().strip().encode()
    assert ident == boot and len(ident) == 36
    assert gdist.same_node(gdist.LocalBackend())


def test_host_segment_registration_is_local_and_survives_having_no_gpu():
    """`HostSegment.register` (ABI 5) is not a collective: a rank whose registration fails -- here: no GPU -- says so and keeps the
    pinned ring; the segment works as before, and `close` after a failed registration is quiet."""
    from golemflavor_amd import dist as gdist
    seg = gdist.HostSegment(gdist.LocalBackend(), 1 << 20)
    assert seg.error is None
    ok, secs = seg.register()
    a = seg.array((1 << 17,))
    a[:] = 3.0
    assert float(a.sum()) == 3.0 * (1 << 17)
    if not ok:
        assert seg.register_error and secs >= 0.0
    assert seg.register() == (ok, 0.0)                      # asked once
    del a
    seg.close()
