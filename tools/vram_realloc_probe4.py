"""Is the 30 GB/s state of a re-allocated buffer's first read-backs a matter of TIME (something the driver does after a free -- wiping
the freed VRAM -- that shares the DMA engine) rather than of the memory?  free -> sleep s -> malloc -> copy, for several s."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from golemflavor_amd.model import empty_hugepages

n = int(9.4e9) // 4096 * 4096
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
a = empty_hugepages((n // 8,))
a[::512] = 0.0
assert hip.hipHostRegister(a.ctypes.data, n, 0) == 0
st = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(st), 1)


def rate(d):
    t0 = time.perf_counter(); hip.hipMemcpyAsync(a.ctypes.data, d, n, 2, st); hip.hipStreamSynchronize(st)
    return round(n / (time.perf_counter() - t0) / 1e9, 1)


d = C.c_void_p(); hip.hipMalloc(C.byref(d), n); hip.hipMemset(d, 1, n); hip.hipDeviceSynchronize()
print(json.dumps({"first allocation": [rate(d), rate(d)]}), flush=True)
for sleep_after_free, sleep_after_malloc, memset in ((0, 0, True), (1.0, 0, True), (0, 1.0, True), (0, 0, False), (0, 0, True), (2.0, 0, True)):
    hip.hipFree(d)
    time.sleep(sleep_after_free)
    d = C.c_void_p(); hip.hipMalloc(C.byref(d), n)
    if memset:
        hip.hipMemset(d, 2, n); hip.hipDeviceSynchronize()
    time.sleep(sleep_after_malloc)
    print(json.dumps({"sleep after free": sleep_after_free, "sleep after malloc (+memset)": sleep_after_malloc, "memset": memset,
                      "copies": [rate(d), rate(d), rate(d)]}), flush=True)
