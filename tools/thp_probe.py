import os, sys, time, ctypes as C, mmap
sys.path.insert(0, os.getcwd())
import numpy as np
from golemflavor_amd import _lib
L = _lib.lib()
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("enabled", "defrag", "shmem_enabled"):
    try: print("THP", f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except Exception as e: print("THP", f, e)
libc = C.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
nbytes = 1887436800
for mode in ("plain", "hugepage", "plain", "hugepage"):
    a = np.empty(nbytes, dtype=np.uint8)
    addr = a.ctypes.data
    if mode == "hugepage":
        lo = (addr + (1 << 21) - 1) & ~((1 << 21) - 1)
        rc = libc.madvise(C.c_void_p(lo), (addr + nbytes - lo) & ~((1 << 21) - 1), 14)      # MADV_HUGEPAGE
    t0 = time.perf_counter()
    L.gf_host_prepare(C.c_void_p(addr), nbytes)
    dt = time.perf_counter() - t0
    print("%-9s gf_host_prepare of %.2f GB: %.1f ms = %.1f GB/s" % (mode, nbytes / 1e9, 1e3 * dt, nbytes / dt / 1e9), end="")
    if mode == "hugepage": print("  (madvise rc %d)" % rc, end="")
    print()
    del a
print(open("/proc/meminfo").read().split("AnonHugePages")[1].split("\n")[0])
