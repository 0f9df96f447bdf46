"""Multi-GPU orchestration: independent chains (grid points) sharded over one process per GPU.

The reference's only "distributed" layer is HTCondor: one job per grid point, results as .npy
files on a shared filesystem (submitter/mc_texture_dag.py:57-71, submitter/sens_dag.py:75-95).
Here the same partitioning runs inside one node: grid point g belongs to rank g mod world, the
data path has no collective, and only two exchanges exist -- a broadcast of the packed model
descriptors at start and a gather of the chain blocks at the end.

Backends
  RcclBackend : RCCL over xGMI through the library's own C ABI (gf_comm_*); device-side
                all-gather.  The 128-byte RCCL unique id is shipped out of band (a torch.distributed
                store / gloo broadcast, or any user callable).
  GlooBackend : torch.distributed on CPU tensors -- what the world_size-2 tests run on, and the
                control plane (barrier, max-over-ranks) of bench.py.
  LocalBackend: world size 1.
"""
import ctypes as C

import numpy as np

from . import _lib


def shard(n_items, rank, world):
    """Grid point g -> rank g mod world; returns this rank's points in order."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, n_items, world))


def slots_per_rank(n_items, world):
    return (n_items + world - 1) // world


class LocalBackend:
    rank, world = 0, 1

    def broadcast_bytes(self, buf, root=0):
        return buf

    def allgather(self, arr):
        return np.asarray(arr)[None, ...]

    def barrier(self):
        pass

    def close(self):
        pass


class GlooBackend:
    """torch.distributed (gloo, CPU).  The process group must already be initialised."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def broadcast_bytes(self, buf, root=0):
        t = self._torch.frombuffer(bytearray(buf), dtype=self._torch.uint8).clone()
        self._dist.broadcast(t, src=root)
        return bytes(t.numpy().tobytes())

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        t = self._torch.from_numpy(a.copy())
        outs = [self._torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])

    def barrier(self):
        self._dist.barrier()

    def close(self):
        pass


class RcclBackend:
    """RCCL over xGMI via gf_comm_* (one communicator per process, bound to `device`).

    `exchange_id(id_bytes_or_None) -> id_bytes` ships rank 0's unique id to every rank; with
    torch.distributed initialised (gloo) the default uses broadcast_object_list."""

    def __init__(self, rank, world, device, exchange_id=None):
        self._L = _lib.lib()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        uid = None
        if self.rank == 0:
            buf = (C.c_uint8 * _lib.GF_COMM_ID_BYTES)()
            _lib.check(self._L.gf_comm_unique_id(buf), "gf_comm_unique_id")
            uid = bytes(buf)
        if exchange_id is None:
            exchange_id = self._exchange_via_torch
        uid = exchange_id(uid) if self.world > 1 else uid
        idb = (C.c_uint8 * _lib.GF_COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        _lib.check(self._L.gf_comm_create(idb, self.rank, self.world, self.device, C.byref(h)), "gf_comm_create")
        self._h = h

    @staticmethod
    def _exchange_via_torch(uid):
        import torch.distributed as dist
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def broadcast_bytes(self, buf, root=0):
        raw = (C.c_uint8 * len(buf)).from_buffer_copy(bytes(buf))
        _lib.check(self._L.gf_comm_broadcast(self._h, raw, len(buf), int(root)), "gf_comm_broadcast")
        return bytes(raw)

    def allgather_device(self, d_send, d_recv, bytes_per_rank):
        _lib.check(self._L.gf_comm_allgather(self._h, d_send, d_recv, int(bytes_per_rank)), "gf_comm_allgather")

    def allgather(self, arr, model):
        """Host array in, (world, ...) host array out, staged through `model`'s device buffers."""
        a = np.ascontiguousarray(arr)
        d_send = model.alloc(a.nbytes).upload(a)
        d_recv = model.alloc(a.nbytes * self.world)
        self.allgather_device(d_send.ptr, d_recv.ptr, a.nbytes)
        out = d_recv.download((self.world,) + a.shape, dtype=a.dtype)
        d_send.free()
        d_recv.free()
        return out

    def barrier(self):
        _lib.check(self._L.gf_comm_barrier(self._h), "gf_comm_barrier")

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.gf_comm_destroy(self._h)
            self._h = None


def open_rccl(rank, world, device, timeout=120.0):
    """An RcclBackend whose bootstrap cannot hang the job: the communicator is created in a helper thread
    with a bounded wait, and (world > 1) all ranks agree over torch.distributed whether everybody got one.
    Returns (backend or None, error string or None)."""
    import threading
    box = {}

    def _setup():
        try:
            box["b"] = RcclBackend(rank, world, device)
        except Exception as exc:           # noqa: BLE001
            box["err"] = "%s: %s" % (type(exc).__name__, exc)

    th = threading.Thread(target=_setup, daemon=True)
    th.start()
    th.join(timeout=timeout)
    err = "timeout: RCCL communicator setup did not finish" if th.is_alive() else box.get("err")
    if world > 1:
        import torch.distributed as tdist
        errs = [None] * world
        tdist.all_gather_object(errs, err)
        err = next((e for e in errs if e is not None), None)
    if err is not None:
        b = box.get("b")
        if b is not None:
            b.close()
        return None, err
    return box["b"], None


def broadcast_descriptors(descs, backend, root=0):
    """Rank `root`'s packed gf_model_desc list -> every rank (the 'fixed physics constants')."""
    from ._lib import GfModelDesc
    size = C.sizeof(GfModelDesc)
    n = backend.broadcast_bytes(np.int64(len(descs) if backend.rank == root else 0).tobytes(), root)
    n = int(np.frombuffer(n, dtype=np.int64)[0])
    blob = b"".join(bytes(memoryview(d)) for d in descs) if backend.rank == root else bytes(size * n)
    blob = backend.broadcast_bytes(blob, root)
    return [GfModelDesc.from_buffer_copy(blob[i * size:(i + 1) * size]) for i in range(n)]


def gather_chains(local, n_points, backend, allgather=None):
    """All-gather per-grid-point chain blocks.

    local : {grid index: ndarray}, all blocks of one common shape and dtype float64.
    Returns the list of n_points arrays in grid order (on every rank).
    """
    mine = shard(n_points, backend.rank, backend.world)
    if sorted(local) != mine:
        raise ValueError("rank %d holds points %s, expected %s" % (backend.rank, sorted(local), mine))
    shapes = {np.asarray(v).shape for v in local.values()}
    if len(shapes) > 1:
        raise ValueError("chain blocks must share one shape, got %s" % shapes)
    if backend.world == 1 and allgather is None:
        return [np.asarray(local[g], dtype=np.float64) for g in range(n_points)]      # nothing to exchange, no copies
    # every rank needs the block shape even if it owns no point: agree on it through a tiny gather
    shp = np.zeros(8, dtype=np.int64)
    if shapes:
        s = shapes.pop()
        shp[0] = len(s)
        shp[1:1 + len(s)] = s
    ag = allgather or backend.allgather
    all_shp = ag(shp)
    ref = all_shp[np.argmax(all_shp[:, 0] > 0)] if np.any(all_shp[:, 0] > 0) else shp
    block_shape = tuple(int(x) for x in ref[1:1 + int(ref[0])])
    slots = slots_per_rank(n_points, backend.world)
    send = np.zeros((slots,) + block_shape, dtype=np.float64)
    for slot, g in enumerate(mine):
        send[slot] = local[g]
    allb = ag(send)                                    # (world, slots, ...)
    return [np.array(allb[g % backend.world, g // backend.world]) for g in range(n_points)]


def run_grid(points, run_chain, backend):
    """Run `run_chain(point, grid_index)` for this rank's shard and gather all chains."""
    mine = shard(len(points), backend.rank, backend.world)
    local = {g: np.asarray(run_chain(points[g], g), dtype=np.float64) for g in mine}
    return gather_chains(local, len(points), backend)
