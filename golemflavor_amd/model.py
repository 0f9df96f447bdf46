"""`Model`: one compiled posterior bound to one GPU (a `gf_model` handle) + device buffers.

All numerical work happens in libgolemhip.so; this module only marshals numpy arrays across
the C ABI and owns handle lifetimes.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GF_LAYOUT_AOS, GF_LAYOUT_SOA, check  # noqa: F401


PREPARE_MIN_BYTES = 32 << 20
PREPARE_MAX_BYTES = 16 << 20       # from here on gf_memcpy_d2h brings its own pinned staging; its host threads map the pages as they fill them


def empty_for_download(shape, dtype=np.float64, copier_maps_pages=False):
    """np.empty whose pages are already mapped when it is big: a device-to-host copy into untouched memory runs at
    page-fault speed (11-20 GB/s on the MI355X box), into touched memory at PCIe speed (48-56 GB/s);
    gf_host_prepare touches the pages from several threads (2 GiB in 17 ms).  `copier_maps_pages`: the array goes to
    gf_memcpy_d2h, which does this itself for large copies."""
    out = np.empty(shape, dtype=dtype)
    if out.nbytes >= PREPARE_MIN_BYTES and not (copier_maps_pages and out.nbytes >= PREPARE_MAX_BYTES):
        check(_lib.lib().gf_host_prepare(out.ctypes.data_as(C.c_void_p), out.nbytes), "gf_host_prepare")
    return out


def empty_hugepages(shape, dtype=np.float64):
    """np.empty on anonymous memory the kernel may back with 2 MiB pages (mmap + madvise(MADV_HUGEPAGE), 2 MiB-aligned): a
    large device-to-host copy into FRESH memory spends much of its host time in page faults -- 4 KiB at a time, 3 million of
    them for a 12.6 GB chain; with transparent huge pages in `madvise` mode (the MI355X boxes') the same range takes 6 000.
    Falls back to plain np.empty where mmap.madvise / MADV_HUGEPAGE are missing.  The array keeps its mapping alive."""
    import mmap
    n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    huge = 2 << 20
    if n < 4 * huge or not hasattr(mmap, "MADV_HUGEPAGE"):
        return np.empty(shape, dtype=dtype)
    mm = mmap.mmap(-1, n + huge, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE)
    base = np.frombuffer(mm, dtype=np.uint8)
    off = (-base.ctypes.data) % huge
    try:
        mm.madvise(mmap.MADV_HUGEPAGE, 0, n + huge)
    except (OSError, ValueError):
        pass
    return base[off:off + n].view(dtype).reshape(shape)


def anon_huge_bytes():
    """AnonHugePages of this process (bytes), from /proc/self/smaps_rollup; -1 where that is not readable."""
    try:
        with open("/proc/self/smaps_rollup") as f:
            for line in f:
                if line.startswith("AnonHugePages:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return -1


class Prefaulted:
    """A result array allocated AHEAD of the device work that fills it, its pages mapped by a background thread
    (gf_host_prepare releases the GIL) while the GPU is busy with that work -- a grid scan knows the size of its result
    once the burn-in is enqueued, and the host has nothing else to do until the chain is ready.  The mapping does not change
    the content (madvise(MADV_POPULATE_WRITE)); `get()` / `finish()` join the thread -- do that before launching or allocating
    anything: page-faulting threads hold those calls up."""

    def __init__(self, shape, dtype=np.float64, threads=0):
        import threading
        self.array = np.empty(shape, dtype=dtype)
        self._t = None
        if self.array.nbytes >= PREPARE_MIN_BYTES:
            L = _lib.lib()
            # (the closure keeps the array alive for as long as the thread touches it, whatever happens to this object)
            self._t = threading.Thread(target=lambda a=self.array: L.gf_host_prepare_n(a.ctypes.data_as(C.c_void_p), a.nbytes, int(threads)), daemon=True)
            self._t.start()

    def finish(self):
        if self._t is not None:
            self._t.join()
            self._t = None

    def get(self):
        self.finish()
        return self.array


def _as_theta(theta, ndim):
    th = np.ascontiguousarray(theta, dtype=np.float64)
    if th.ndim == 1:
        th = th.reshape(1, -1)
    if th.ndim != 2 or th.shape[1] != ndim:
        # same exception type the reference raises on a length mismatch (llh.py:67-71)
        raise AssertionError(
            "Length of MCMC scan is not the same as the input params\n"
            "theta shape={0}\nndim={1}".format(np.shape(theta), ndim))
    return th


class DeviceBuffer:
    """A hipMalloc'd block owned by a Model (freed with it or explicitly)."""

    def __init__(self, model, nbytes):
        self.model, self.nbytes = model, int(nbytes)
        p = C.c_void_p()
        check(model._L.gf_device_alloc(model._h, self.nbytes, C.byref(p)), "gf_device_alloc")
        self.ptr = p
        model._buffers.append(self)

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(self.model._L.gf_memcpy_h2d(self.model._h, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes), "h2d")
        return self

    def download(self, shape, dtype=np.float64, offset_bytes=0, out=None):
        if out is None:
            out = empty_for_download(shape, dtype, copier_maps_pages=True)
        assert out.shape == tuple(shape) and out.dtype == np.dtype(dtype) and out.flags.c_contiguous
        assert out.nbytes + offset_bytes <= self.nbytes
        src = C.c_void_p(self.ptr.value + offset_bytes)
        check(self.model._L.gf_memcpy_d2h(self.model._h, out.ctypes.data_as(C.c_void_p), src, out.nbytes), "d2h")
        return out

    def at(self, offset_bytes):
        return C.c_void_p(self.ptr.value + int(offset_bytes))

    def free(self):
        if self.ptr is not None and self.model._h is not None:
            check(self.model._L.gf_device_free(self.model._h, self.ptr), "gf_device_free")
            self.ptr = None
            if self in self.model._buffers:
                self.model._buffers.remove(self)


class Event:
    def __init__(self, model):
        self.model = model
        p = C.c_void_p()
        check(model._L.gf_event_create(C.byref(p)), "gf_event_create")
        self.ptr = p

    def record(self):
        check(self.model._L.gf_event_record(self.model._h, self.ptr), "gf_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(self.model._L.gf_event_elapsed_ms(self.ptr, stop.ptr, C.byref(ms)), "gf_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self.ptr is not None:
                self.model._L.gf_event_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class Model:
    """Device-resident posterior.  `desc` comes from descriptor.compile_model."""

    def __init__(self, desc, device=0):
        self._L = _lib.lib()               # raises GolemHipUnavailable if the .so is missing
        self.desc = desc
        self.ndim = int(desc.ndim)
        self.mode = int(desc.mode)
        self.device = int(device)
        self._buffers = []
        h = C.c_void_p()
        check(self._L.gf_model_create(C.byref(desc), self.device, C.byref(h)), "gf_model_create")
        self._h = h

    # -- host-buffer hot path --------------------------------------------------------
    def lnprob(self, theta, want_fr=False, want_status=True):
        """theta (n, ndim) -> lnprob (n,) [, fr (n,3)] [, status (n,) int32]."""
        th = _as_theta(theta, self.ndim)
        n = th.shape[0]
        out = np.empty(n, dtype=np.float64)
        fr = np.empty((n, 3), dtype=np.float64) if want_fr else None
        st = np.empty(n, dtype=np.int32) if want_status else None
        check(self._L.gf_lnprob_batch(
            self._h, th.ctypes.data_as(_lib._dp), n, out.ctypes.data_as(_lib._dp),
            fr.ctypes.data_as(_lib._dp) if want_fr else None,
            st.ctypes.data_as(_lib._ip) if want_status else None), "gf_lnprob_batch")
        res = (out,)
        if want_fr:
            res += (fr,)
        if want_status:
            res += (st,)
        return res if len(res) > 1 else out

    def lnprob_cube(self, cube, cols, base, want_status=True):
        """MultiNest-style batch (mn.py:26-45): cube (n, nscan) in the unit cube -> lnprob (n,) [, status]; column
        cols[k] is mapped onto the model's own box on the device, the other columns take base[col]."""
        u = np.ascontiguousarray(np.atleast_2d(np.asarray(cube, dtype=np.float64)))
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        base = np.ascontiguousarray(base, dtype=np.float64)
        if u.shape[1] != len(cols) or base.shape != (self.ndim,):
            raise AssertionError("Length of MultiNest scan paramset is not the same as the input params")
        n = u.shape[0]
        out = np.empty(n, dtype=np.float64)
        st = np.empty(n, dtype=np.int32) if want_status else None
        check(self._L.gf_lnprob_cube_batch(self._h, u.ctypes.data_as(_lib._dp), n, len(cols), cols.ctypes.data_as(_lib._ip),
                                           base.ctypes.data_as(_lib._dp), out.ctypes.data_as(_lib._dp), None,
                                           st.ctypes.data_as(_lib._ip) if want_status else None), "gf_lnprob_cube_batch")
        return (out, st) if want_status else out

    def propagate(self, theta, want_status=True):
        """theta (n, ndim) -> measured composition (n, 3) [, status]."""
        th = _as_theta(theta, self.ndim)
        n = th.shape[0]
        fr = np.empty((n, 3), dtype=np.float64)
        st = np.empty(n, dtype=np.int32) if want_status else None
        check(self._L.gf_propagate_batch(
            self._h, th.ctypes.data_as(_lib._dp), n, fr.ctypes.data_as(_lib._dp),
            st.ctypes.data_as(_lib._ip) if want_status else None), "gf_propagate_batch")
        return (fr, st) if want_status else fr

    def haar_draw(self, seed, n, first_draw=0, want_angles=False):
        fr = np.empty((n, 3), dtype=np.float64)
        ang = np.empty((n, 4), dtype=np.float64) if want_angles else None
        check(self._L.gf_haar_draw(self._h, int(seed), int(first_draw), int(n),
                                   ang.ctypes.data_as(_lib._dp) if want_angles else None,
                                   fr.ctypes.data_as(_lib._dp)), "gf_haar_draw")
        return (fr, ang) if want_angles else fr

    def flavor_histogram(self, fr, nbins):
        """np.histogramdd(fr, bins=(nbins,)*3, range=((0,1),)*3)[0] on the GPU (golemflavor/plot.py:365-370)."""
        f = np.ascontiguousarray(fr, dtype=np.float64).reshape(-1, 3)
        counts = np.zeros((nbins, nbins, nbins), dtype=np.uint64)
        check(self._L.gf_flavor_histogram(self._h, f.ctypes.data_as(_lib._dp), f.shape[0], int(nbins),
                                          counts.ctypes.data_as(C.POINTER(C.c_uint64))), "gf_flavor_histogram")
        return counts

    # -- device-resident path ----------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def lnprob_device(self, d_theta, n, d_lnprob, d_fr=None, d_status=None, layout=GF_LAYOUT_AOS):
        check(self._L.gf_lnprob_batch_device(self._h, d_theta, int(layout), int(n), d_lnprob, d_fr, d_status),
              "gf_lnprob_batch_device")

    def propagate_device(self, d_theta, n, d_fr, d_status=None, layout=GF_LAYOUT_AOS):
        check(self._L.gf_propagate_batch_device(self._h, d_theta, int(layout), int(n), d_fr, d_status),
              "gf_propagate_batch_device")

    def haar_draw_device(self, seed, first_draw, n, d_angles, d_fr):
        check(self._L.gf_haar_draw_device(self._h, int(seed), int(first_draw), int(n), d_angles, d_fr),
              "gf_haar_draw_device")

    def sync(self):
        check(self._L.gf_model_sync(self._h), "gf_model_sync")

    def event(self):
        return Event(self)

    def close(self):
        if getattr(self, "_h", None) is not None:
            for b in list(self._buffers):
                b.free()
            self._L.gf_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
