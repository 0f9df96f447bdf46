"""When a stream's device-to-host copies are in the 30 GB/s state (tools/vram_realloc_probe2.py), what brings them back?  (a) a NEW stream,
(b) a tiny synchronous copy on the null stream first, (c) a tiny copy on the same stream first, (d) just trying again."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from golemflavor_amd.model import empty_hugepages

n = int(4e9) // 4096 * 4096
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


def rate(d, s, nbytes=n):
    t0 = time.perf_counter(); hip.hipMemcpyAsync(a.ctypes.data, d, nbytes, 2, s); hip.hipStreamSynchronize(s)
    return round(nbytes / (time.perf_counter() - t0) / 1e9, 1)


for it in range(8):
    d = C.c_void_p(); hip.hipMalloc(C.byref(d), n); hip.hipMemset(d, it + 1, n); hip.hipDeviceSynchronize()
    rec = {"allocation": it, "stream": [rate(d, st), rate(d, st)]}
    if rec["stream"][1] < 45 and it >= 2:
        cure = ["new stream", "tiny null-stream copy first", "tiny same-stream copy first", "again", "64 MB pieces", "new stream"][(it - 2) % 6]
        rec["cure"] = cure
        if cure == "new stream":
            s2 = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s2), 1)
            rec["after"] = [rate(d, s2), rate(d, s2), rate(d, st)]
        elif cure == "tiny null-stream copy first":
            hip.hipMemcpy(a.ctypes.data, d, 4096, 2)
            rec["after"] = [rate(d, st), rate(d, st)]
        elif cure == "tiny same-stream copy first":
            hip.hipMemcpyAsync(a.ctypes.data, d, 4096, 2, st); hip.hipStreamSynchronize(st)
            rec["after"] = [rate(d, st), rate(d, st)]
        elif cure == "64 MB pieces":
            t0 = time.perf_counter()
            for off in range(0, n, 64 << 20):
                hip.hipMemcpyAsync(a.ctypes.data + off, d.value + off, min(64 << 20, n - off), 2, st)
            hip.hipStreamSynchronize(st)
            rec["after"] = [round(n / (time.perf_counter() - t0) / 1e9, 1)]
        else:
            rec["after"] = [rate(d, st), rate(d, st)]
    hip.hipFree(d)
    print(json.dumps(rec), flush=True)
