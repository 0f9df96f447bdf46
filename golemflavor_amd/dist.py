"""Multi-GPU orchestration: independent chains (grid points) sharded over one process per GPU.

The reference's only "distributed" layer is HTCondor: one job per grid point, results as .npy
files on a shared filesystem (submitter/mc_texture_dag.py:57-71, submitter/sens_dag.py:75-95).
Here the same partitioning runs inside one node: grid point g belongs to rank (g + g div world) mod world (`owner`), the
data path has no collective, and only two exchanges exist -- a broadcast of the packed model
descriptors at start and a gather of the chain blocks at the end.

Backends
  RcclBackend  : RCCL over xGMI through the library's own C ABI (gf_comm_*); device-side all-gather.  The
                 128-byte RCCL unique id is shipped out of band by the control plane below.
  SocketBackend: the control plane of bench.py and scan.py -- rendezvous, barrier, max-over-ranks, the unique id --
                 over plain TCP sockets (standard library only; rank 0 listens, the others connect).  No PyTorch in
                 the process: which librccl a process maps depends on what was imported first (torch bundles its
                 own under the same soname), so the product's processes import nothing that links a ROCm runtime
                 before libgolemhip.so.
  GlooBackend  : torch.distributed on CPU tensors, for callers that already live inside a torch process group (and
                 the world_size-2 gloo test).
  LocalBackend : world size 1.
"""
import ctypes as C
import hashlib
import os
import socket
import struct
import time

import numpy as np

from . import _lib


def owner(g, world):
    """The rank that runs grid point g: (g + g div world) mod world -- round robin with the start moved on by one in every block
    of `world` points.  Plain g mod world (rounds 1-3) hands a rank every world-th point, and the grids of the scans are
    products whose fastest axis has `world`-friendly lengths: on 8 ranks ALL 32 top-scale points of the C5 grid (8 scales
    fastest) -- the chains that sit in the reference's failing region and cost several times the others (profiles/r04/
    chain_census.txt) -- went to rank 7.  Skewed, every rank gets 4 of them.  Each block of `world` consecutive points still
    gives every rank exactly one, so a point's slot on its rank stays g div world."""
    return (g + g // world) % world


def shard(n_items, rank, world):
    """This rank's grid points, in order (independent chains, no data-path collective): the points g with
    `owner(g, world) == rank`, i.e. one per block of `world` points."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    out = []
    for b in range((n_items + world - 1) // world):
        g = b * world + ((rank - b) % world)
        if g < n_items:
            out.append(g)
    return out


def slots_per_rank(n_items, world):
    return (n_items + world - 1) // world


class LocalBackend:
    rank, world = 0, 1

    def broadcast_bytes(self, buf, root=0):
        return buf

    def allgather(self, arr):
        return np.asarray(arr)[None, ...]

    def allgather_bytes(self, payload):
        return [bytes(payload)]

    def gather_bytes(self, payload, root=0):
        return [bytes(payload)]

    def allreduce_max(self, values):
        return np.asarray(values, dtype=np.float64).reshape(-1)

    def barrier(self):
        pass

    def close(self):
        pass


class HostSegment:
    """One block of host memory mapped by every rank of a node: the destination of a multi-rank scan.

    The reference's N jobs each write their own chain file to one place (golemflavor/mcmc.py:108-126,
    submitter/mc_texture_dag.py:57-71): N writers, no funnel.  Here every rank reads its own chains back from its own GPU over
    its own PCIe link straight into its part of this segment, and rank 0 -- which maps the same pages -- has the whole grid
    without a byte crossing xGMI, a socket or a second PCIe link.

    Backing: a file under /dev/shm when that mount has room, else an anonymous memfd the other ranks open through
    /proc/<pid>/fd (no size limit but memory).  The name is gone from the filesystem as soon as every rank has mapped it, so
    nothing outlives the processes.  A COLLECTIVE constructor: every rank of `control` calls it with the same size; on any
    rank's failure all of them get `.error` set and `.buffer` None, in step."""

    def __init__(self, control, nbytes, root=0, directory=None):
        import mmap
        import secrets
        self.nbytes, self.buffer, self.error, self.kind = int(nbytes), None, None, None
        self._mm = None
        self._alloc_thread = None
        rank, err, fd, path = control.rank, "", -1, ""
        if rank == root:
            try:
                d = directory or os.environ.get("GF_SEGMENT_DIR") or "/dev/shm"
                st = os.statvfs(d) if os.path.isdir(d) else None
                if st is not None and st.f_bavail * st.f_frsize > self.nbytes + (64 << 20) and os.access(d, os.W_OK):
                    path = os.path.join(d, "gf_segment_%d_%s" % (os.getpid(), secrets.token_hex(6)))
                    fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                    self.kind = "file under %s" % d
                else:
                    fd = os.memfd_create("gf_segment")
                    path = "/proc/%d/fd/%d" % (os.getpid(), fd)
                    self.kind = "memfd"
                os.ftruncate(fd, max(self.nbytes, 1))
                # The segment's pages are 4 KiB shmem pages (transparent huge pages are off for shmem on the MI355X boxes) and have to
                # be allocated and zeroed before anybody can write them: from the read-back's copy threads, fault by fault, that ran
                # at 1.8 GB/s per rank; posix_fallocate from ONE thread does 17 GB/s, more threads only contend (tools/
                # shm_fault_probe.py, profiles/r04/shm_pages.txt).  So the root allocates the file in the background, from here on --
                # behind the sampling for a scan that has one; a writer that gets ahead of it just takes the slow path for a while.
                if self.nbytes >= (64 << 20) and not os.environ.get("GF_SEGMENT_NO_FALLOCATE"):
                    import threading
                    fd2 = os.dup(fd)

                    def _allocate(fd2=fd2, n=self.nbytes):
                        try:
                            step = 1 << 30
                            for off in range(0, n, step):
                                os.posix_fallocate(fd2, off, min(step, n - off))
                        except OSError:
                            pass
                        finally:
                            os.close(fd2)

                    self._alloc_thread = threading.Thread(target=_allocate, daemon=True)
                    self._alloc_thread.start()
            except Exception as exc:       # noqa: BLE001
                err = "%s: %s" % (type(exc).__name__, exc)
        meta = control.broadcast_bytes(("%s\n%s\n%s" % (err, path, self.kind or "")).encode() if rank == root else b"", root).decode()
        err, path, kind = meta.split("\n", 2)
        self.kind = kind or None
        if not err and rank != root:
            try:
                fd = os.open(path, os.O_RDWR)
            except Exception as exc:       # noqa: BLE001
                err = "rank %d: %s: %s" % (rank, type(exc).__name__, exc)
        if not err and fd >= 0:
            try:
                self._mm = mmap.mmap(fd, max(self.nbytes, 1), mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
            except Exception as exc:       # noqa: BLE001
                err = err or "rank %d: mmap: %s: %s" % (rank, type(exc).__name__, exc)
        errs = control.allgather_bytes(err.encode())       # everybody has mapped it (or said why not): the name can go
        if rank == root and path and not path.startswith("/proc/"):
            try:
                os.unlink(path)
            except OSError:
                pass
        if fd >= 0:
            os.close(fd)
        err = next((e.decode() for e in errs if e), "")
        if err:
            self.error = err
            self.close()
        else:
            self.buffer = self._mm

    def array(self, shape, dtype=np.float64, offset=0):
        """A writable view of the segment (no copy); it keeps the mapping alive."""
        n = int(np.prod(shape, dtype=np.int64))
        return np.frombuffer(self._mm, dtype=dtype, count=n, offset=int(offset)).reshape(shape)

    def wait_allocated(self):
        """(root) Join the background allocation of the segment's pages; returns at once elsewhere."""
        t, self._alloc_thread = self._alloc_thread, None
        if t is not None:
            t.join()

    def register(self):
        """Register this rank's mapping of the WHOLE segment with the HIP runtime (`gf_host_register`, ABI 5), so that the DMA
        engines write this rank's read-backs straight into it -- 57 GB/s instead of 28-47 through the pinned ring and the host's
        copy threads (profiles/r04/host_register.txt).  Local, not collective (a rank whose registration fails keeps the ring:
        every read-back finds out by itself whether its destination is registered); call it once the pages exist
        (`wait_allocated` on the root + a barrier), on a rank that has selected its device.  Pinning 4 KiB shared-memory pages runs
        at ~11 GB/s: for a segment that outlives one scan.  Returns (registered, seconds)."""
        import time as _time
        if self._mm is None or self.error is not None or getattr(self, "_registered", None) is not None:
            return bool(getattr(self, "_registered", None)), 0.0
        from . import _lib
        t0 = _time.perf_counter()
        try:
            a = np.frombuffer(self._mm, dtype=np.uint8)
            _lib.check(_lib.lib().gf_host_register(a.ctypes.data, a.nbytes), "gf_host_register")
            self._registered = int(a.ctypes.data)
        except Exception as exc:           # noqa: BLE001  (no library, no GPU, RLIMIT_MEMLOCK ...): the ring stays
            self._registered = 0
            self.register_error = "%s: %s" % (type(exc).__name__, exc)
        return bool(self._registered), _time.perf_counter() - t0

    def close(self):
        """Unmap (a rank that holds views keeps the pages until they are gone: the mapping is closed by the last reference)."""
        self.wait_allocated()
        if getattr(self, "_registered", None):
            try:
                from . import _lib
                _lib.lib().gf_host_unregister(self._registered)
            except Exception:              # noqa: BLE001
                pass
            self._registered = None
        mm, self._mm, self.buffer = self._mm, None, None
        if mm is not None:
            try:
                mm.close()
            except BufferError:
                pass                       # views exist: numpy holds the mapping, the pages go with the last of them


class ControlPlaneTimeout(TimeoutError):
    """A rank of the job stopped answering on the control plane (it is alive enough to keep its socket open)."""


class SocketBackend:
    """Control plane over TCP, standard library only.  Star topology: rank 0 listens on (addr, port) and relays.

    addr / port default to MASTER_ADDR / GF_RDZV_PORT, else MASTER_PORT + 1 (torch.distributed.run keeps MASTER_PORT
    itself for its own store); if that port is taken rank 0 walks up to 31 ports further and the clients follow,
    recognising the right listener by a handshake that carries the job's token (TORCHELASTIC_RUN_ID by default).
    Collectives: broadcast_bytes, allgather_bytes, allgather (numpy), barrier, allreduce_max."""

    MAGIC = b"GFRDZV1\0"
    PORT_SPAN = 32
    # A collective whose peer has stopped answering fails instead of hanging: seconds a single send / receive may take.  The
    # default stays well under the 600 s the driver grants a bench run, so that a silent peer turns into a reported error
    # (socket.timeout on the waiting rank, ConnectionError on the others when it leaves) and not into a killed job;
    # GF_CONTROL_TIMEOUT or the `op_timeout` argument override it.
    OP_TIMEOUT = 240.0

    def __init__(self, rank, world, addr=None, port=None, token=None, timeout=120.0, op_timeout=None):
        self.rank, self.world = int(rank), int(world)
        if op_timeout is None:
            op_timeout = float(os.environ.get("GF_CONTROL_TIMEOUT", 0) or self.OP_TIMEOUT)
        self.OP_TIMEOUT = float(op_timeout)
        if not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside world of %d" % (self.rank, self.world))
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("GF_RDZV_PORT", 0)) or int(os.environ.get("MASTER_PORT", "29500")) + 1
        tok = token if token is not None else os.environ.get("TORCHELASTIC_RUN_ID", "") + ":" + os.environ.get("MASTER_PORT", "")
        self._token = hashlib.sha256(("%s|%d" % (tok, self.world)).encode()).digest()
        self._peers = {}          # rank 0: rank -> socket; others: {0: socket}
        self._listener = None
        deadline = time.monotonic() + float(timeout)
        if self.world == 1:
            return
        if self.rank == 0:
            self._serve(int(port), deadline)
        else:
            self._connect(int(port), deadline)

    # -- wire helpers ---------------------------------------------------------------------
    @staticmethod
    def _send(sock, payload):
        sock.sendall(struct.pack("<Q", len(payload)) + payload)

    @staticmethod
    def _recv_exact(sock, n):
        buf = bytearray()
        while len(buf) < n:
            try:
                chunk = sock.recv(min(n - len(buf), 1 << 20))
            except socket.timeout:
                raise ControlPlaneTimeout("control plane: a peer did not answer within %.0f s (GF_CONTROL_TIMEOUT)"
                                          % (sock.gettimeout() or 0.0)) from None
            if not chunk:
                raise ConnectionError("peer closed the rendezvous connection")
            buf += chunk
        return bytes(buf)

    @classmethod
    def _recv(cls, sock):
        (n,) = struct.unpack("<Q", cls._recv_exact(sock, 8))
        return cls._recv_exact(sock, n)

    def _serve(self, port, deadline):
        last = None
        for p in range(port, port + self.PORT_SPAN):
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                ls.bind((self.addr, p))
                ls.listen(self.world)
                self._listener, self.port = ls, p
                break
            except OSError as exc:
                last = exc
                ls.close()
        if self._listener is None:
            raise OSError("rank 0 could not bind a rendezvous port in %d..%d: %s" % (port, port + self.PORT_SPAN - 1, last))
        while len(self._peers) < self.world - 1:
            left = deadline - time.monotonic()
            if left <= 0:
                raise TimeoutError("rendezvous: %d of %d ranks connected" % (len(self._peers) + 1, self.world))
            self._listener.settimeout(left)
            try:
                conn, _ = self._listener.accept()
            except socket.timeout:
                continue
            conn.settimeout(10.0)
            try:
                hello = self._recv_exact(conn, len(self.MAGIC) + 8 + 32)
                magic, (w, r), tok = hello[:8], struct.unpack("<ii", hello[8:16]), hello[16:]
                if magic != self.MAGIC or w != self.world or tok != self._token or not 0 < r < self.world or r in self._peers:
                    conn.close()
                    continue
                conn.sendall(b"OK")
            except (OSError, ConnectionError, struct.error):
                conn.close()
                continue
            conn.settimeout(self.OP_TIMEOUT)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._peers[r] = conn

    def _connect(self, port, deadline):
        hello = self.MAGIC + struct.pack("<ii", self.world, self.rank) + self._token
        while time.monotonic() < deadline:
            for p in range(port, port + self.PORT_SPAN):
                try:
                    sk = socket.create_connection((self.addr, p), timeout=2.0)
                except OSError:
                    continue
                try:
                    sk.settimeout(5.0)
                    sk.sendall(hello)
                    if self._recv_exact(sk, 2) == b"OK":
                        sk.settimeout(self.OP_TIMEOUT)
                        sk.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._peers[0], self.port = sk, p
                        return
                except (OSError, ConnectionError):
                    pass
                sk.close()
            time.sleep(0.05)
        raise TimeoutError("rank %d could not reach the rendezvous at %s:%d.." % (self.rank, self.addr, port))

    # -- collectives ------------------------------------------------------------------------
    def allgather_bytes(self, payload):
        """Every rank's payload, in rank order, on every rank."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv(self._peers[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for r in range(1, self.world):
                self._send(self._peers[r], blob)
            return parts
        self._send(self._peers[0], payload)
        blob = self._recv(self._peers[0])
        parts, o = [], 0
        while o < len(blob):
            (n,) = struct.unpack_from("<Q", blob, o)
            parts.append(blob[o + 8:o + 8 + n])
            o += 8 + n
        return parts

    def broadcast_bytes(self, buf, root=0):
        if self.world == 1:
            return bytes(buf)
        return self.allgather_bytes(bytes(buf) if self.rank == root else b"")[root]

    def gather_bytes(self, payload, root=0):
        """Every rank's payload, in rank order, on `root` only (None elsewhere): nothing is relayed back."""
        payload = bytes(payload) if not isinstance(payload, (bytes, bytearray, memoryview)) else payload
        if self.world == 1:
            return [bytes(payload)]
        if root != 0:                       # star topology: rank 0 is the hub
            parts = self.allgather_bytes(payload)
            return parts if self.rank == root else None
        if self.rank == 0:
            return [bytes(payload)] + [self._recv(self._peers[r]) for r in range(1, self.world)]
        self._peers[0].sendall(struct.pack("<Q", len(payload)))
        self._peers[0].sendall(payload)
        return None

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        parts = self.allgather_bytes(a.tobytes())
        return np.stack([np.frombuffer(x, dtype=a.dtype).reshape(a.shape) for x in parts])

    def barrier(self):
        self.allgather_bytes(b"")

    def allreduce_max(self, values):
        v = np.asarray(values, dtype=np.float64).reshape(-1)
        return self.allgather(v).max(axis=0)

    def close(self):
        for sk in self._peers.values():
            try:
                sk.close()
            except OSError:
                pass
        self._peers = {}
        if self._listener is not None:
            self._listener.close()
            self._listener = None


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

    def allgather_bytes(self, payload):
        outs = [None] * self.world
        self._dist.all_gather_object(outs, bytes(payload))
        return outs

    def gather_bytes(self, payload, root=0):
        outs = [None] * self.world if self.rank == root else None
        self._dist.gather_object(bytes(payload), outs, dst=root)
        return outs

    def allreduce_max(self, values):
        return self.allgather(np.asarray(values, dtype=np.float64).reshape(-1)).max(axis=0)

    def barrier(self):
        self._dist.barrier()

    def close(self):
        pass


class RcclBackend:
    """RCCL over xGMI via gf_comm_* (one communicator per process, bound to `device`).

    `uid`: rank 0's 128-byte RCCL unique id, already shipped to this rank (`exchange_unique_id`).  For callers that hold
    no id yet, `control` does the exchange here: the backend that ships it (SocketBackend / GlooBackend), or a callable
    `exchange_id(id_bytes_or_None) -> id_bytes`."""

    kind = "rccl"

    def __init__(self, rank, world, device, control=None, uid=None):
        self._L = _lib.lib()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        if uid is None:
            if self.rank == 0:
                uid = make_unique_id()
            if self.world > 1:
                if control is None:
                    raise ValueError("RcclBackend needs a control backend to ship the unique id")
                uid = control(uid) if callable(control) else control.broadcast_bytes(uid or b"", 0)
        if uid is None or len(uid) != _lib.GF_COMM_ID_BYTES:
            raise ValueError("RCCL unique id must be %d bytes, got %s" % (_lib.GF_COMM_ID_BYTES, None if uid is None else len(uid)))
        idb = (C.c_uint8 * _lib.GF_COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        _lib.check(self._L.gf_comm_create(idb, self.rank, self.world, self.device, C.byref(h)), "gf_comm_create")
        self._h = h

    def broadcast_bytes(self, buf, root=0):
        raw = (C.c_uint8 * len(buf)).from_buffer_copy(bytes(buf))
        _lib.check(self._L.gf_comm_broadcast(self._h, raw, len(buf), int(root)), "gf_comm_broadcast")
        return bytes(raw)

    def allgather_device(self, d_send, d_recv, bytes_per_rank):
        _lib.check(self._L.gf_comm_allgather(self._h, d_send, d_recv, int(bytes_per_rank)), "gf_comm_allgather")

    def gather_device(self, d_send, d_recv_on_root, bytes_per_rank, root=0):
        """Rank r's block -> d_recv_on_root + r * bytes_per_rank on `root` (pass None elsewhere): only the root holds
        world x the block (the reference: N jobs saving N files to one place, golemflavor/mcmc.py:108-126)."""
        _lib.check(self._L.gf_comm_gather(self._h, d_send, d_recv_on_root, int(bytes_per_rank), int(root)), "gf_comm_gather")

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

    def info(self):
        """(nranks, rank, device) as the COMMUNICATOR reports them (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): what
        RCCL itself saw, not what this object was told."""
        n, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.check(self._L.gf_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)), "gf_comm_info")
        return n.value, r.value, d.value

    def nranks(self):
        return self.info()[0]

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.gf_comm_destroy(self._h)
            self._h = None


class IpcBackend:
    """The device gather without RCCL, for ranks on ONE node: blocks are shared between the processes as hipIpc handles
    (gf_ipc_export / gf_ipc_gather; 64 bytes each over `control`), the root copies them device to device.  Same
    `gather_device` face as RcclBackend, so `scan.DeviceGather` takes either.  The fallback when the RCCL communicator
    cannot be set up (`open_device_gather`), and the only inter-process DEVICE path a one-GPU box can exercise (RCCL refuses
    two ranks on one device): tests/test_gpu_mcmc.py runs a two-rank scan over it.

    Every rank takes both collectives of a gather whatever happened locally: a rank whose export failed ships an EMPTY
    handle, the root treats that as an error and its report reaches everybody (`GolemHipError(GF_ERR_COMM)` on all ranks
    alike, so that the caller can fall back to the host gather in step)."""

    kind = "hipIpc"

    def __init__(self, rank, world, device, control):
        self._L = _lib.lib()
        self.rank, self.world, self.device, self.control = int(rank), int(world), int(device), control
        self.init_seconds = 0.0

    def _export(self, d_send):
        h = (C.c_uint8 * _lib.GF_IPC_HANDLE_BYTES)()
        try:
            _lib.check(self._L.gf_ipc_export(d_send, h), "gf_ipc_export")
        except Exception as exc:           # noqa: BLE001  -- the peers are on their way into the gather: stay in step
            return b"", "%s: %s" % (type(exc).__name__, exc)
        return bytes(h), ""

    def gather_device(self, d_send, d_recv_on_root, bytes_per_rank, root=0):
        handle, local_err = self._export(d_send)
        blobs = self.control.gather_bytes(handle, root)
        err = b""
        if self.rank == root:
            try:
                bad = [r for r, x in enumerate(blobs) if len(x) != _lib.GF_IPC_HANDLE_BYTES]
                if bad:
                    raise RuntimeError("no hipIpc handle from rank(s) %s%s" % (bad, (": " + local_err) if local_err else ""))
                allh = (C.c_uint8 * (_lib.GF_IPC_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(blobs))
                _lib.check(self._L.gf_ipc_gather(self.device, allh, self.world, self.rank, d_send, d_recv_on_root, int(bytes_per_rank)),
                           "gf_ipc_gather")
            except Exception as exc:       # noqa: BLE001  -- the senders are waiting: tell them before raising
                err = ("%s: %s" % (type(exc).__name__, exc)).encode()
        # the senders keep their blocks until the root is done; the root's outcome reaches everybody
        err = self.control.broadcast_bytes(err, root)
        if err:
            raise _lib.GolemHipError(_lib.GF_ERR_COMM, "ipc gather failed on rank %d: %s%s"
                                     % (root, err.decode(), (" (this rank: %s)" % local_err) if local_err else ""))

    def probe(self):
        """A 16-byte export / open / copy round trip through `gather_device`, agreed on by all ranks: None, or the error.
        `open_device_gather` runs it before handing the backend out, so that ranks whose hostnames merely coincide (pods on
        different machines) or whose GPUs cannot reach each other find out at set-up time, not after the chains are sampled."""
        err = ""
        d_send = d_recv = None
        try:
            d_send, d_recv = C.c_void_p(), C.c_void_p()
            _lib.check(self._L.gf_device_malloc(self.device, 256, C.byref(d_send)), "gf_device_malloc")
            if self.rank == 0:
                _lib.check(self._L.gf_device_malloc(self.device, 256 * self.world, C.byref(d_recv)), "gf_device_malloc")
        except Exception as exc:           # noqa: BLE001
            err = "%s: %s" % (type(exc).__name__, exc)
            d_send = d_send if d_send is not None and d_send.value else None
        try:
            self.gather_device(d_send if d_send is not None else C.c_void_p(), d_recv if self.rank == 0 else None, 16, 0)
        except Exception as exc:           # noqa: BLE001
            err = err or "%s: %s" % (type(exc).__name__, exc)
        for d in (d_send, d_recv):
            if d is not None and d.value:
                self._L.gf_device_release(self.device, d)
        errs = self.control.allgather_bytes(err.encode())
        return next((e.decode() for e in errs if e), None)

    def barrier(self):
        self.control.barrier()

    def nranks(self):
        return self.world

    def close(self):
        pass


def node_identity():
    """What two processes must share to share a node: the kernel's boot id (one per running kernel -- containers and pods of one
    machine see the same, two machines never do) next to the hostname (which alone says little: pods on different machines
    are often named alike, and differ on one machine)."""
    import socket as _socket
    boot = ""
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            boot = f.read().strip()
    except OSError:
        pass
    return (boot or "no-boot-id:" + _socket.gethostname()).encode()


def same_node(control):
    """True when every rank of `control` runs on this machine (hipIpc handles and shared host segments mean nothing elsewhere).
    A collective: every rank calls it."""
    ids = control.allgather_bytes(node_identity())
    return all(n == ids[0] for n in ids)


def open_device_gather(rank, world, device, control, timeout=120.0, make_id=None, backend_factory=None, ipc_factory=None):
    """The device-to-device gather of a scan: RCCL if its communicator comes up (`open_rccl`), else -- the ranks being on one
    node -- hipIpc.  Returns (backend or None, rccl error or None, stuck): the RCCL error is reported whatever the fallback
    did; `backend.kind` says which path the chains will take.

    What follows an RCCL failure is decided from state EVERY rank holds: `open_rccl` agrees on the error text, but `stuck` (the
    helper thread still inside ncclCommInitRank) is local -- in the usual failure one rank's init errors out while its peers
    sit in the bootstrap until the timeout.  So the ranks first exchange their `stuck` flags; if anybody is stuck nobody tries
    hipIpc (a process with a thread inside RCCL's bootstrap is not one to open peers' memory in), otherwise all of them take
    `same_node` and the probe together.  The control plane sees the same collectives in the same order on every rank."""
    b, err, stuck = open_rccl(rank, world, device, control, timeout=timeout, make_id=make_id or make_unique_id,
                              backend_factory=backend_factory)
    if b is not None or world < 2:
        return b, err, stuck
    flags = control.allgather_bytes(b"1" if stuck else b"0")
    if any(f == b"1" for f in flags):
        return None, err, stuck
    if not same_node(control):
        return None, err, False
    ipc = (ipc_factory or IpcBackend)(rank, world, device, control)
    perr = ipc.probe()
    if perr is not None:
        return None, "%s; hipIpc probe: %s" % (err, perr), False
    return ipc, err, False


def rccl_library_info():
    """'<ncclGetVersion code> <path of the librccl this process mapped>' (goes into the JSON lines)."""
    buf = C.create_string_buffer(600)
    _lib.check(_lib.lib().gf_comm_library_info(buf, 600), "gf_comm_library_info")
    return buf.value.decode()


def make_unique_id():
    """Rank 0's RCCL unique id (128 bytes)."""
    buf = (C.c_uint8 * _lib.GF_COMM_ID_BYTES)()
    _lib.check(_lib.lib().gf_comm_unique_id(buf), "gf_comm_unique_id")
    return bytes(buf)


def exchange_unique_id(rank, world, control, make_id=make_unique_id, timeout=30.0):
    """Rank 0's unique id on every rank, or (None, error text) on EVERY rank when rank 0 could not make one.

    Runs in the caller's (main) thread and every rank always takes part, with the same two collectives in the same order
    whatever happened on rank 0: the control plane stays in step.  Rank 0 sends the 128-byte id, or an EMPTY payload meaning
    "no id" followed by its error text; a payload of any other length is refused on all ranks alike.  `make_id` itself is
    bounded by `timeout` (a helper thread), so a hung ncclGetUniqueId cannot keep rank 0 out of the exchange."""
    import threading
    uid, err = b"", ""
    if rank == 0:
        box = {}

        def _make():
            try:
                box["uid"] = make_id()
            except Exception as exc:           # noqa: BLE001
                box["err"] = "%s: %s" % (type(exc).__name__, exc)

        th = threading.Thread(target=_make, daemon=True)
        th.start()
        th.join(timeout=timeout)
        if th.is_alive():
            err = "timeout: RCCL unique id not available after %.0f s" % timeout
        elif "err" in box:
            err = box["err"]
        else:
            uid = bytes(box.get("uid") or b"")
            if len(uid) != _lib.GF_COMM_ID_BYTES:
                uid, err = b"", "unique id of %d bytes, expected %d" % (len(uid), _lib.GF_COMM_ID_BYTES)
    if world > 1:
        uid = control.broadcast_bytes(uid, 0)
        err = control.broadcast_bytes(err.encode(), 0).decode()
    if len(uid) != _lib.GF_COMM_ID_BYTES:
        return None, err or "rank 0 sent %d bytes instead of a %d-byte unique id" % (len(uid), _lib.GF_COMM_ID_BYTES)
    return uid, None


def open_rccl(rank, world, device, control, timeout=120.0, make_id=make_unique_id, backend_factory=None):
    """An RcclBackend whose bootstrap cannot hang the job or desynchronise the control plane.

    1. `exchange_unique_id` in THIS thread: every rank takes part whatever rank 0's outcome (an empty payload means "no id":
       all ranks then skip RCCL together).
    2. only `gf_comm_create` (ncclCommInitRank) runs in a helper thread with a bounded wait -- it touches no socket.
    3. (world > 1) all ranks agree over `control` whether everybody got a communicator.
    Returns (backend or None, error string or None, stuck) -- `stuck`: the helper thread never came back (the caller
    must leave with os._exit after flushing its output: the thread would block interpreter shutdown).
    `make_id`, `backend_factory(rank, world, device, uid)`: injection points of the CPU tests."""
    import threading
    t0 = time.perf_counter()
    uid, err = exchange_unique_id(rank, world, control, make_id=make_id, timeout=min(timeout, 30.0))
    if uid is None:
        return None, err, False
    box = {}
    factory = backend_factory or (lambda r, w, d, u: RcclBackend(r, w, d, uid=u))

    def _setup():
        try:
            box["b"] = factory(rank, world, device, uid)
        except Exception as exc:           # noqa: BLE001
            box["err"] = "%s: %s" % (type(exc).__name__, exc)

    th = threading.Thread(target=_setup, daemon=True)
    th.start()
    th.join(timeout=timeout)
    stuck = th.is_alive()
    err = "timeout: RCCL communicator setup did not finish in %.0f s" % timeout if stuck else box.get("err")
    if world > 1:
        # every rank reports, the stuck one too (its helper thread sits inside ncclCommInitRank and owns no socket):
        # nobody is left waiting in this exchange
        errs = control.allgather_bytes((err or "").encode())
        err = next((e.decode() for e in errs if e), None)
    if err is not None:
        b = box.get("b")
        if b is not None:
            b.close()
        return None, err, stuck
    b = box["b"]
    try:
        b.init_seconds = time.perf_counter() - t0
    except Exception:                      # noqa: BLE001
        pass
    return b, None, False


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
    return [np.array(allb[owner(g, backend.world), g // backend.world]) for g in range(n_points)]


def gather_chains_to_root(local, n_points, backend, root=0):
    """The same blocks on `root` only (None elsewhere): each rank sends its own points once, nothing is relayed back -- the
    host-side stand-in for `gf_comm_gather` when no RCCL communicator could be set up."""
    mine = shard(n_points, backend.rank, backend.world)
    if sorted(local) != mine:
        raise ValueError("rank %d holds points %s, expected %s" % (backend.rank, sorted(local), mine))
    if backend.world == 1:
        return [np.asarray(local[g], dtype=np.float64) for g in range(n_points)]
    blocks = [np.ascontiguousarray(local[g], dtype=np.float64) for g in mine]
    shapes = {b.shape for b in blocks}
    if len(shapes) > 1:
        raise ValueError("chain blocks must share one shape, got %s" % shapes)
    shp = np.zeros(8, dtype=np.int64)
    if shapes:
        s0 = next(iter(shapes))
        shp[0] = len(s0)
        shp[1:1 + len(s0)] = s0
    payload = shp.tobytes() + b"".join(b.tobytes() for b in blocks)
    parts = backend.gather_bytes(payload, root)
    if backend.rank != root:
        return None
    out = [None] * n_points
    for r, blob in enumerate(parts):
        hdr = np.frombuffer(blob[:64], dtype=np.int64)
        shape = tuple(int(x) for x in hdr[1:1 + int(hdr[0])])
        pts = shard(n_points, r, backend.world)
        if not pts:
            continue
        arr = np.frombuffer(blob, dtype=np.float64, offset=64).reshape((len(pts),) + shape)
        for slot, g in enumerate(pts):
            out[g] = arr[slot]
    return out


def run_grid(points, run_chain, backend):
    """Run `run_chain(point, grid_index)` for this rank's shard and gather all chains."""
    mine = shard(len(points), backend.rank, backend.world)
    local = {g: np.asarray(run_chain(points[g], g), dtype=np.float64) for g in mine}
    return gather_chains(local, len(points), backend)
