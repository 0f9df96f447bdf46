#!/usr/bin/env python3
"""Host side of a chain read-back: how to get fresh host memory ready for a 50 GB/s D2H (GPU box)."""
import ctypes as C, mmap, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golemflavor_amd import configs as Cf
from golemflavor_amd.descriptor import compile_model
from golemflavor_amd.model import Model
libc = C.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
n = 2 << 30
print('THP:', open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), 'cpus', os.cpu_count())
with Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY")) as m:
    d = m.alloc(n)
    def d2h(a):
        t = time.perf_counter(); d.download_into(a) if hasattr(d, 'download_into') else m._L.gf_memcpy_d2h(m._h, a.ctypes.data_as(C.c_void_p), d.ptr, a.nbytes); return a.nbytes / (time.perf_counter() - t) / 1e9
    a = np.empty(n, dtype=np.uint8); print('fresh np.empty D2H: %.1f GB/s' % d2h(a)); print('again (touched): %.1f GB/s' % d2h(a)); del a
    for T in (1, 4, 8, 16):
        a = np.empty(n, dtype=np.uint8)
        t = time.perf_counter()
        step = n // (T * 4)
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(lambda i: a[i:i + step].fill(0), range(0, n, step)))
        tp = time.perf_counter() - t
        print('prefault fill with %2d threads: %.3f s (%.1f GB/s); then D2H %.1f GB/s' % (T, tp, n / tp / 1e9, d2h(a)))
        del a
    for T in (1, 8):
        a = np.empty(n + (2 << 20), dtype=np.uint8)
        addr = (a.ctypes.data + (2 << 20) - 1) & ~((2 << 20) - 1)
        rc = libc.madvise(addr, n, 14)   # MADV_HUGEPAGE
        b = a[addr - a.ctypes.data: addr - a.ctypes.data + n]
        t = time.perf_counter(); step = n // (T * 4)
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(lambda i: b[i:i + step].fill(0), range(0, n, step)))
        tp = time.perf_counter() - t
        print('MADV_HUGEPAGE rc %d prefault %2d threads: %.3f s (%.1f GB/s); then D2H %.1f GB/s' % (rc, T, tp, n / tp / 1e9, d2h(b)))
        del a, b
    mm = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, 'MAP_POPULATE', 0x8000))
    t = time.perf_counter(); a = np.frombuffer(mm, dtype=np.uint8); print('MAP_POPULATE mmap created earlier; D2H %.1f GB/s' % d2h(a))
