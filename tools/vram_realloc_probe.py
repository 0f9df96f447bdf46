"""Does a device buffer that was freed and allocated again read back slower than a first allocation?  (The scans' second and later
results crossed PCIe at 30 GB/s where the first took 56: tools/c4_post_alone.py, arena_scan_ab.py.)  hipMalloc / hipMemset / DMA into
registered host memory / hipFree, in a loop; then the same with ONE allocation kept and reused."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from golemflavor_amd.model import empty_hugepages

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 9.4
n = int(gb * 1e9)
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
a = empty_hugepages((n // 8,))
a[::512] = 0.0
assert hip.hipHostRegister(a.ctypes.data, n, 0) == 0


def rate(d):
    out = []
    for rep in range(2):
        t0 = time.perf_counter(); hip.hipMemcpy(a.ctypes.data, d, n, 2); hip.hipDeviceSynchronize()
        out.append(round(n / (time.perf_counter() - t0) / 1e9, 1))
    return out


for it in range(4):
    d = C.c_void_p()
    t0 = time.perf_counter(); assert hip.hipMalloc(C.byref(d), n) == 0; t_m = time.perf_counter() - t0
    hip.hipMemset(d, it + 1, n); hip.hipDeviceSynchronize()
    r = rate(d)
    t0 = time.perf_counter(); hip.hipFree(d); t_f = time.perf_counter() - t0
    print(json.dumps({"allocation": it, "hipMalloc_s": round(t_m, 4), "d2h_GBps": r, "hipFree_s": round(t_f, 4)}), flush=True)
# two buffers alive at once, as the scans have them (chain + rows), then freed and allocated again
for it in range(3):
    d1, d2 = C.c_void_p(), C.c_void_p()
    hip.hipMalloc(C.byref(d1), n // 3); hip.hipMalloc(C.byref(d2), n)
    hip.hipMemset(d2, 7, n); hip.hipDeviceSynchronize()
    r = rate(d2)
    hip.hipFree(d2); hip.hipFree(d1)
    print(json.dumps({"two buffers, round": it, "d2h_GBps": r}), flush=True)
# one allocation kept and reused: does it stay fast?  (it is the 8th allocation of this process: is IT fast at all?)
d = C.c_void_p(); hip.hipMalloc(C.byref(d), n)
for it in range(3):
    hip.hipMemset(d, it + 1, n); hip.hipDeviceSynchronize()
    print(json.dumps({"one allocation reused, use": it, "d2h_GBps": rate(d)}), flush=True)
hip.hipFree(d)
# a LARGER one after the frees (cannot be served from the same freed block)
big = int(n * 1.6)
d = C.c_void_p(); hip.hipMalloc(C.byref(d), big); hip.hipMemset(d, 3, big); hip.hipDeviceSynchronize()
print(json.dumps({"a larger allocation after the frees": rate(d)}), flush=True)
d_keep = d
# while that one is held: a fresh 9.4 GB
d = C.c_void_p(); hip.hipMalloc(C.byref(d), n); hip.hipMemset(d, 3, n); hip.hipDeviceSynchronize()
print(json.dumps({"a fresh allocation while the larger one is held": rate(d)}), flush=True)
hip.hipFree(d); hip.hipFree(d_keep)
# does a kernel see the difference?  device-to-device copy rate of a re-allocated buffer
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
d1, d2 = C.c_void_p(), C.c_void_p(); hip.hipMalloc(C.byref(d1), n); hip.hipMalloc(C.byref(d2), n)
hip.hipMemset(d1, 1, n); hip.hipDeviceSynchronize()
t0 = time.perf_counter(); hip.hipMemcpy(d2, d1, n, 3); hip.hipDeviceSynchronize(); dt = time.perf_counter() - t0
print(json.dumps({"device-to-device copy of re-allocated buffers, GB/s (read + write)": round(2 * n / dt / 1e9, 1)}), flush=True)
