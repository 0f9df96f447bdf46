"""How fast can the pages of a SHARED host segment (tmpfs under /dev/shm, or a memfd) be brought into existence on this box?
The multi-rank scan's destination is such a segment (dist.HostSegment); its pages are 4 KiB shmem pages (shmem_enabled = never
here), and filling fresh ones from the read-back's copy threads ran at 1.8 GB/s per rank (gpurun_out/r4_e).  Tried here, on
4 GiB each: touching from T threads (gf_host_prepare_n with GF_PREPARE_FORCE_POKE), madvise(MADV_POPULATE_WRITE) from T threads,
posix_fallocate from T threads, and anonymous memory (huge pages) for comparison.   python tools/shm_fault_probe.py"""
import ctypes as C, json, mmap, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golemflavor_amd import _lib
L = _lib.lib()
N = 4 << 30

def seg(kind):
    if kind == "shm":
        path = "/dev/shm/gf_probe_%d" % os.getpid()
        fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
        os.unlink(path)
    else:
        fd = os.memfd_create("gf_probe")
    os.ftruncate(fd, N)
    mm = mmap.mmap(fd, N, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
    return fd, mm

def addr(mm):
    return np.frombuffer(mm, dtype=np.uint8).ctypes.data

for kind in ("shm", "memfd"):
    for how, threads in (("populate", 1), ("populate", 8), ("populate", 32), ("fallocate", 1), ("fallocate", 8), ("fallocate", 32)):
        fd, mm = seg(kind)
        t0 = time.perf_counter()
        if how == "populate":
            L.gf_host_prepare_n(C.c_void_p(addr(mm)), N, threads)
        else:
            per = N // threads
            th = [threading.Thread(target=os.posix_fallocate, args=(fd, i * per, per)) for i in range(threads)]
            [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        # then a first write pass over it (what the copy threads do) -- pages exist now
        a = np.frombuffer(mm, dtype=np.float64)
        t1 = time.perf_counter(); L.gf_host_prepare_n(C.c_void_p(addr(mm)), N, 8); dt2 = time.perf_counter() - t1
        print(json.dumps({"backing": kind, "how": how, "threads": threads, "GBps": round(N / dt / 1e9, 2), "seconds": round(dt, 3),
                          "second_pass_GBps": round(N / dt2 / 1e9, 1)}), flush=True)
        del a
        mm.close(); os.close(fd)
a = np.empty(N // 8)
t0 = time.perf_counter(); L.gf_host_prepare_n(a.ctypes.data_as(C.c_void_p), N, 8); dt = time.perf_counter() - t0
print(json.dumps({"backing": "anonymous (numpy, huge pages)", "how": "populate", "threads": 8, "GBps": round(N / dt / 1e9, 2)}))
