"""Can a small persistent staging buffer make the runtime's DMA fast from ANY source allocation?  Sources in the slow state (freed and
allocated again: tools/vram_realloc_probe.py) are copied device-to-device into a 256 MB staging buffer and leave from there.
Prints the DMA rate from several staging allocations themselves, then the rate of the whole two-hop pipeline."""
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
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
a = empty_hugepages((n // 8,))
a[::512] = 0.0
assert hip.hipHostRegister(a.ctypes.data, n, 0) == 0
st = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(st), 1)


def d2h_rate(d, nbytes, reps=3):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); hip.hipMemcpyAsync(a.ctypes.data, d, nbytes, 2, st); hip.hipStreamSynchronize(st)
        out.append(round(nbytes / (time.perf_counter() - t0) / 1e9, 1))
    return out


# put the process into the state of a second scan: a large allocation, freed, and another
d = C.c_void_p(); hip.hipMalloc(C.byref(d), n); hip.hipMemset(d, 1, n); hip.hipDeviceSynchronize()
print(json.dumps({"first 9.4 GB allocation": d2h_rate(d, n, 2)}), flush=True)
hip.hipFree(d)
src = C.c_void_p(); hip.hipMalloc(C.byref(src), n); hip.hipMemset(src, 2, n); hip.hipDeviceSynchronize()
print(json.dumps({"second 9.4 GB allocation (the source below)": d2h_rate(src, n, 2)}), flush=True)
S = 256 << 20
stagings = []
for k in range(6):
    s = C.c_void_p(); hip.hipMalloc(C.byref(s), S); hip.hipMemset(s, 3, S); hip.hipDeviceSynchronize()
    stagings.append(s)
    print(json.dumps({"staging allocation": k, "256 MB d2h GB/s": d2h_rate(s, S, 4), "64 MB": d2h_rate(s, 64 << 20, 4)}), flush=True)
for k, s in enumerate(stagings[:3]):
    t0 = time.perf_counter()
    for off in range(0, n, S):
        ln = min(S, n - off)
        hip.hipMemcpyAsync(s, src.value + off, ln, 3, st)
        hip.hipMemcpyAsync(a.ctypes.data + off, s, ln, 2, st)
    hip.hipStreamSynchronize(st)
    dt = time.perf_counter() - t0
    print(json.dumps({"two hops through staging": k, "GB/s": round(n / dt / 1e9, 1), "ok": bool(a.view(np.uint8)[0] == 2 and a.view(np.uint8)[n - 1] == 2)}), flush=True)
