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
         "d2h_s": 0.19 if rank == 0 else 0.0, "gather_bytes": 4718592000 if rank == 0 else 0, "block_bytes": 4718592000}
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
    """Grid point g must come back from slot [g mod world, g div world] whatever the grid / world sizes: grids that do not
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
    assert stats[0]["block_bytes"] == slots * per * width * 8
    assert stats[0]["gather_bytes"] == stats[0]["block_bytes"] * (world - 1)


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
