"""What puts a device allocation's read-back into the 30 GB/s state?  Cycles of hipMalloc / hipMemset / copy to registered host memory /
hipFree, the copy issued (a) as hipMemcpyAsync on a created non-blocking stream, (b) as a synchronous hipMemcpy, (c) both."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from golemflavor_amd.model import empty_hugepages

mode = sys.argv[1] if len(sys.argv) > 1 else "async"
n = int(9.4e9) // 4096 * 4096
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
a = empty_hugepages((n // 8,))
a[::512] = 0.0
assert hip.hipHostRegister(a.ctypes.data, n, 0) == 0
st = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(st), 1)


def async_rate(d):
    t0 = time.perf_counter(); hip.hipMemcpyAsync(a.ctypes.data, d, n, 2, st); hip.hipStreamSynchronize(st)
    return round(n / (time.perf_counter() - t0) / 1e9, 1)


def sync_rate(d):
    t0 = time.perf_counter(); hip.hipMemcpy(a.ctypes.data, d, n, 2); hip.hipDeviceSynchronize()
    return round(n / (time.perf_counter() - t0) / 1e9, 1)


for it in range(5):
    d = C.c_void_p(); hip.hipMalloc(C.byref(d), n); hip.hipMemset(d, it + 1, n); hip.hipDeviceSynchronize()
    rec = {"mode": mode, "allocation": it}
    if mode in ("async", "both"):
        rec["async_on_stream"] = [async_rate(d), async_rate(d)]
    if mode in ("sync", "both"):
        rec["sync_null_stream"] = [sync_rate(d), sync_rate(d)]
    if mode == "both":
        rec["async_again"] = [async_rate(d)]
    hip.hipFree(d)
    print(json.dumps(rec), flush=True)
