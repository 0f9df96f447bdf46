"""What would a REGISTERED result arena buy the big read-backs?  hipHostRegister of a hugepage-backed numpy array (untouched / touched),
the device-to-host rate straight into it (one hipMemcpyAsync, no staging ring, no host threads), and into the same array unregistered
through the library's pinned ring for comparison.  python tools/host_register_probe.py [GB]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from golemflavor_amd import _lib
from golemflavor_amd.model import empty_hugepages

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gb * 1e9) // 8
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
L = _lib.lib()
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), n * 8) == 0
assert hip.hipMemset(d, 1, n * 8) == 0
hip.hipDeviceSynchronize()
for touched in (False, True):
    a = empty_hugepages((n,))
    if touched:
        t0 = time.perf_counter(); a[::512] = 0.0; t_touch = time.perf_counter() - t0
    else:
        t_touch = 0.0
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(a.ctypes.data, n * 8, 0)
    t_reg = time.perf_counter() - t0
    rates = []
    for rep in range(3):
        t0 = time.perf_counter()
        rc2 = hip.hipMemcpy(a.ctypes.data, d, n * 8, 2)
        hip.hipDeviceSynchronize()
        rates.append(n * 8 / (time.perf_counter() - t0) / 1e9)
    ok = bool(a[0] == a[-1] and a.view(np.uint8)[0] == 1)
    t0 = time.perf_counter(); hip.hipHostUnregister(a.ctypes.data); t_unreg = time.perf_counter() - t0
    print(json.dumps({"GB": gb, "touched_first": touched, "touch_s": round(t_touch, 3), "hipHostRegister_rc": rc, "register_s": round(t_reg, 3),
                      "register_GBps": round(gb / t_reg, 1), "d2h_direct_GBps": [round(r, 1) for r in rates], "memcpy_rc": rc2, "ok": ok,
                      "unregister_s": round(t_unreg, 3)}), flush=True)
    del a
# the same bytes into a fresh, unregistered array through a plain hipMemcpy (the runtime's own staging)
a = empty_hugepages((n,))
rates = []
for rep in range(3):
    t0 = time.perf_counter(); hip.hipMemcpy(a.ctypes.data, d, n * 8, 2); hip.hipDeviceSynchronize()
    rates.append(n * 8 / (time.perf_counter() - t0) / 1e9)
print(json.dumps({"GB": gb, "unregistered_plain_hipMemcpy_GBps": [round(r, 1) for r in rates]}), flush=True)

# the C5 scan's geometry as a pitched copy straight into the registered array: 256 rows (chains) of 16 steps x 512 walkers x 12 columns
hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
rows, width = 256, 16 * 512 * 12 * 8
nblk = min(62, int(n * 8 // (rows * width)))
dp = nblk * width                                  # a chain's stored steps are contiguous on the host: pitch = all its blocks
a = empty_hugepages((rows * dp // 8,))
a[::512] = 0.0
for registered in (True, False):
    if registered:
        hip.hipHostRegister(a.ctypes.data, a.nbytes, 0)
    t0 = time.perf_counter()
    for b in range(nblk):
        hip.hipMemcpy2D(a.ctypes.data + b * width, dp, d.value + b * rows * width, width, width, rows, 2)
    hip.hipDeviceSynchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"pitched_blocks": nblk, "rows": rows, "row_bytes": width, "registered": registered, "GBps": round(nblk * rows * width / dt / 1e9, 1)}), flush=True)
    if registered:
        hip.hipHostUnregister(a.ctypes.data)
del a
# a shared-memory file mapping (dist.HostSegment's kind of memory): allocate, map, register, copy
import mmap
fd = os.open("/dev/shm/gf_hostreg_probe", os.O_CREAT | os.O_RDWR, 0o600)
os.unlink("/dev/shm/gf_hostreg_probe")
t0 = time.perf_counter(); os.posix_fallocate(fd, 0, n * 8); t_fa = time.perf_counter() - t0
mm = mmap.mmap(fd, n * 8)
s = np.frombuffer(mm, dtype=np.float64)
t0 = time.perf_counter(); rc = hip.hipHostRegister(s.ctypes.data, n * 8, 0); t_reg = time.perf_counter() - t0
rates = []
for rep in range(3):
    t0 = time.perf_counter(); rc2 = hip.hipMemcpy(s.ctypes.data, d, n * 8, 2); hip.hipDeviceSynchronize()
    rates.append(n * 8 / (time.perf_counter() - t0) / 1e9)
print(json.dumps({"shm_GB": gb, "fallocate_s": round(t_fa, 3), "hipHostRegister_rc": rc, "register_s": round(t_reg, 3), "d2h_direct_GBps": [round(r, 1) for r in rates],
                  "memcpy_rc": rc2, "ok": bool(s.view(np.uint8)[-1] == 1)}), flush=True)
hip.hipHostUnregister(s.ctypes.data)
