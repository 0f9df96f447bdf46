"""CPU: the N > 1 path on gloo, world_size 2 -- sharding, descriptor broadcast, gather of chains."""
import os
import socket

import numpy as np
import pytest

from golemflavor_amd import dist as gdist


def test_shard_partition_is_exact():
    for n in (0, 1, 7, 64, 256):
        for world in (1, 2, 3, 8):
            seen = sorted(g for r in range(world) for g in gdist.shard(n, r, world))
            assert seen == list(range(n))
            assert max([len(gdist.shard(n, r, world)) for r in range(world)] + [0]) == gdist.slots_per_rank(n, world)
            for r in range(world):
                pts = gdist.shard(n, r, world)
                assert all(gdist.owner(g, world) == r for g in pts)
                assert [g // world for g in pts] == list(range(len(pts)))          # a point's slot on its rank is g div world
    # the skew spreads what plain g mod world piles up: the C5 grid's fastest axis is the scale (8 values), and on 8 ranks its 32
    # top-scale points -- the expensive ones -- must not all land on one rank
    top = [g for g in range(256) if g % 8 == 7]
    per_rank = [sum(gdist.owner(g, 8) == r for g in top) for r in range(8)]
    assert per_rank == [4] * 8
    with pytest.raises(ValueError):
        gdist.shard(4, 2, 2)


def test_gather_chains_for_any_grid_and_world():
    """Property (hypothesis): for any number of grid points and ranks -- including ranks that own no point and
    grids that do not divide evenly -- every rank ends up with every point's block, in grid order.  The ranks
    run as threads; the all-gather is a barrier plus a shared slot table."""
    import threading
    from hypothesis import given, settings, strategies as st

    class ThreadBackend:
        def __init__(self, rank, world, slots, barrier):
            self.rank, self.world, self._slots, self._barrier = rank, world, slots, barrier

        def allgather(self, arr):
            self._slots[self.rank] = np.array(arr)
            self._barrier.wait()
            out = np.stack([self._slots[q] for q in range(self.world)])
            self._barrier.wait()                      # nobody overwrites a slot before everybody has read it
            return out

    @settings(max_examples=40, deadline=None)
    @given(n=st.integers(1, 23), world=st.integers(2, 7), rows=st.integers(1, 4))
    def check(n, world, rows):
        blocks = {g: np.full((rows, 3), float(g)) + np.arange(3) for g in range(n)}
        slots, barrier, results, errors = [None] * world, threading.Barrier(world), [None] * world, []

        def run(rank):
            try:
                local = {g: blocks[g] for g in gdist.shard(n, rank, world)}
                results[rank] = gdist.gather_chains(local, n, ThreadBackend(rank, world, slots, barrier))
            except Exception as exc:       # noqa: BLE001
                errors.append(exc)
                barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=30)
        assert not errors, errors
        for r in range(world):
            assert len(results[r]) == n
            for g in range(n):
                assert np.array_equal(results[r][g], blocks[g])

    check()


def test_local_backend_roundtrip():
    b = gdist.LocalBackend()
    chains = gdist.run_grid([0.1, 0.2, 0.3], lambda p, g: np.full((4, 2), p + g), b)
    assert len(chains) == 3 and np.all(chains[2] == 2.3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    from golemflavor_amd import configs as Cf
    from golemflavor_amd import dist as gd
    from golemflavor_amd import mcmc as mcmc_utils
    from golemflavor_amd.descriptor import compile_model
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = gd.GlooBackend()
    # 1) broadcast of the packed descriptors ("fixed physics constants") from rank 0
    descs = []
    if rank == 0:
        descs = [compile_model(Cf.texture_paramset(6), "PRIOR_ONLY", scale_fixed=-40.0 - k) for k in range(5)]
    got = gd.broadcast_descriptors(descs, b)
    assert len(got) == 5 and [d.scale_fixed for d in got] == [-40.0 - k for k in range(5)] and got[3].ndim == 7
    # 2) independent chains: grid point g -> rank g mod world, seeded per grid point
    points = [(-1.0 + 0.5 * g, 0.3 + 0.1 * g) for g in range(5)]

    def run_chain(point, g):
        mu, sig = point

        class F:
            vectorized = True

            def __call__(self, th):
                return -0.5 * np.sum(((np.atleast_2d(th) - mu) / sig) ** 2, axis=1)
        s = mcmc_utils.EnsembleSampler(8, 2, F(), seed=100 + g)
        s.run_mcmc(np.random.default_rng(g).normal(mu, sig, size=(8, 2)), 30)
        return s.flatchain

    chains = gd.run_grid(points, run_chain, b)
    np.save(os.path.join(tmpdir, "chains_rank%d.npy" % rank), np.stack(chains))
    b.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "chains_rank0.npy")
    b = np.load(tmp_path / "chains_rank1.npy")
    assert a.shape == (5, 8 * 30, 2) and np.array_equal(a, b)          # every rank holds every chain
    # identical to a single-process run of the same seeded chains
    from golemflavor_amd import mcmc as mcmc_utils
    for g in range(5):
        mu, sig = -1.0 + 0.5 * g, 0.3 + 0.1 * g

        class F:
            vectorized = True

            def __call__(self, th):
                return -0.5 * np.sum(((np.atleast_2d(th) - mu) / sig) ** 2, axis=1)
        s = mcmc_utils.EnsembleSampler(8, 2, F(), seed=100 + g)
        s.run_mcmc(np.random.default_rng(g).normal(mu, sig, size=(8, 2)), 30)
        assert np.array_equal(a[g], s.flatchain)


_SOCKET_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from golemflavor_amd import dist as gdist
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
b = gdist.SocketBackend(rank, world, addr="127.0.0.1", port=port, token="test-job", timeout=60)
out = {"rank": rank, "port": b.port}
out["bcast"] = b.broadcast_bytes(b"physics constants" if rank == 0 else b"", 0).decode()
out["bcast_from_last"] = b.broadcast_bytes(b"x%d" % rank, world - 1).decode()
out["gather"] = b.allgather(np.arange(3, dtype=np.float64) + 10 * rank).tolist()
out["max"] = b.allreduce_max([float(rank), 5.0 - rank]).tolist()
b.barrier()
# the descriptor broadcast and the chain gather of the scan, on this control plane
from golemflavor_amd._lib import GfModelDesc
d = GfModelDesc(); d.ndim = 7 if rank == 0 else 0; d.smearing = 0.02 if rank == 0 else 0.0
descs = gdist.broadcast_descriptors([d, d] if rank == 0 else [], b)
out["descs"] = [(x.ndim, x.smearing) for x in descs]
n = 5
local = {g: np.full((2, 3), float(g)) for g in gdist.shard(n, rank, world)}
chains = gdist.gather_chains(local, n, b)
out["chains"] = [float(c[0, 0]) for c in chains]
b.close()
print(json.dumps(out))
"""


@pytest.mark.parametrize("world", [2, 3])
def test_socket_rendezvous_world_size_n(world, tmp_path):
    """The stdlib control plane of bench.py / scan.py (no torch in the process): `world` spawned processes meet on a TCP
    port, broadcast, all-gather, barrier, max-reduce, and run the scan's descriptor broadcast and chain gather on it.
    The port handed over is deliberately occupied: rank 0 moves to the next free one and the others find it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_SOCKET_CHILD)
    blocker = socket.socket()
    blocker.bind(("127.0.0.1", 0))
    blocker.listen(1)
    port = blocker.getsockname()[1]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se
        outs.append(json.loads(so.strip().splitlines()[-1]))
    blocker.close()
    for r, o in enumerate(outs):
        assert o["rank"] == r and o["port"] != port and o["port"] == outs[0]["port"]
        assert o["bcast"] == "physics constants" and o["bcast_from_last"] == "x%d" % (world - 1)
        assert o["gather"] == [[10.0 * q + k for k in range(3)] for q in range(world)]
        assert o["max"] == [world - 1.0, 5.0]
        assert o["descs"] == [[7, 0.02], [7, 0.02]]
        assert o["chains"] == [0.0, 1.0, 2.0, 3.0, 4.0]
