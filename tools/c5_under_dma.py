"""What slows the C5 sampler while its chain is read back?  The sampler alone (chain kept on the device) against the same run while another
thread keeps a DMA going on a stream of its own: device -> registered host (linear, and pitched like the scan's blocks), host -> device,
and device -> device.  python tools/c5_under_dma.py"""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench  # noqa: E402
from golemflavor_amd import scan, mcmc as mcmc_utils  # noqa: E402
from golemflavor_amd.model import empty_hugepages  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
n = 4 << 30
host = empty_hugepages((n // 8,)); host[::512] = 0.0
assert hip.hipHostRegister(host.ctypes.data, n, 0) == 0
d1, d2 = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(d1), n); hip.hipMalloc(C.byref(d2), n)
st = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(st), 1)

pts = scan.sens_grid()
jobs = [scan._SensPoint(p, g, nwalkers=512, device=0) for g, p in enumerate(pts)]
s = mcmc_utils.DeviceEnsembleSampler(512, 12, [j.f for j in jobs], seed=25, stream_ids=list(range(len(jobs))))
s.on_nonunitary = "-inf"
os.environ["GF_SAMPLER_CHAIN"] = "0"
s.run_mcmc(np.stack([j.p0 for j in jobs]), 100, storechain=False)
stop = threading.Event()
moved = [0]


def pump(kind):
    piece = 256 << 20
    while not stop.is_set():
        for off in range(0, n, piece):
            if kind == "d2h":
                hip.hipMemcpyAsync(host.ctypes.data + off, d1.value + off, piece, 2, st)
            elif kind == "h2d":
                hip.hipMemcpyAsync(d1.value + off, host.ctypes.data + off, piece, 1, st)
            elif kind == "d2d":
                hip.hipMemcpyAsync(d2.value + off, d1.value + off, piece, 3, st)
            elif kind == "d2h_pitched":           # 256 rows of 786 432 B, 48 MB apart on the host side
                hip.hipMemcpy2DAsync(host.ctypes.data, 16 << 20, d1.value + (off % (2 << 30)), 786432, 786432, 256, 2, st)
            hip.hipStreamSynchronize(st)
            moved[0] += piece if kind != "d2h_pitched" else 256 * 786432
            if stop.is_set():
                break


for kind in (None, "d2h", "d2h_pitched", "h2d", "d2d", None):
    stop.clear(); moved[0] = 0
    th = None
    if kind:
        th = threading.Thread(target=pump, args=(kind,)); th.start(); time.sleep(0.05)
    t0 = time.perf_counter()
    s.run_mcmc(None, 400, storechain=False)
    dt = time.perf_counter() - t0
    m = moved[0]
    stop.set()
    if th:
        th.join()
    print(json.dumps({"beside": kind or "nothing", "us_per_half_step": round(1e6 * dt / 800, 1), "dma_GBps": round(m / dt / 1e9, 1) if kind else None}), flush=True)
