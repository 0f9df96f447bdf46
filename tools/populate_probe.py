"""Mapping the pages of a fresh 12.6 GB array in a process whose allocator has already taken and returned arrays of that size (the
state bench.py's C5 scan finds): touching one byte per page from 16 threads (gf_host_prepare) against madvise(MADV_POPULATE_WRITE)
from 1 / 4 / 16 threads."""
import ctypes as C, os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import _lib
L = _lib.lib()
libc = C.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
n = 12_582_912_000
for warm in range(2):                      # fragment: allocate, touch and free twice
    a = np.empty(n // 8); L.gf_host_prepare(a.ctypes.data_as(C.c_void_p), a.nbytes); del a
def populate(a, nt):
    addr, nb = a.ctypes.data, a.nbytes
    lo = (addr + 4095) & ~4095; hi = (addr + nb) & ~4095
    per = ((hi - lo) // nt) & ~4095
    errs = []
    def work(k):
        s = lo + k * per; e = hi if k == nt - 1 else s + per
        rc = libc.madvise(C.c_void_p(s), e - s, 23)
        if rc != 0: errs.append(C.get_errno())
    th = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    return errs
for label, fn in (("touch, 16 threads", lambda a: L.gf_host_prepare(a.ctypes.data_as(C.c_void_p), a.nbytes)),
                  ("MADV_POPULATE_WRITE, 1 thread", lambda a: populate(a, 1)), ("MADV_POPULATE_WRITE, 4 threads", lambda a: populate(a, 4)),
                  ("MADV_POPULATE_WRITE, 16 threads", lambda a: populate(a, 16)), ("touch, 16 threads", lambda a: L.gf_host_prepare(a.ctypes.data_as(C.c_void_p), a.nbytes))):
    a = np.empty(n // 8)
    t0 = time.perf_counter(); r = fn(a); dt = time.perf_counter() - t0
    print("%-34s %.3f s = %.1f GB/s %s" % (label, dt, n / dt / 1e9, r if r else ""), flush=True)
    t0 = time.perf_counter(); del a; print("    (freeing it: %.3f s)" % (time.perf_counter() - t0))
