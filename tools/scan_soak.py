"""Thirty C4 and thirty C5 scans (100 + 200 steps) in one process into the registered arena: every result's finite fraction, the spread of
the times, and what the device cache holds at the end (it must not grow with the number of scans)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench  # noqa: E402
from golemflavor_amd import scan, _lib  # noqa: E402

L = _lib.lib()
L.gf_devcache_stats.restype = None
L.gf_devcache_stats.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong)]


def cache():
    idle, live, reuses = C.c_size_t(), C.c_size_t(), C.c_ulonglong()
    L.gf_devcache_stats(0, C.byref(idle), C.byref(live), C.byref(reuses))
    return {"idle_GB": round(idle.value / 1e9, 2), "live_GB": round(live.value / 1e9, 2), "reuses": int(reuses.value)}


arena = scan.ResultArena(2516582400)
scan.set_result_arena(arena)
times = {"C4": [], "C5": []}
for rep in range(30):
    for cfg in ("C4", "C5"):
        r = bench.extra_scan(0, cfg, 100, 200)
        assert r["finite_fraction"] > 0.9, r
        times[cfg].append(r["seconds"])
    if rep in (0, 1, 9, 29):
        print(json.dumps({"after scans": 2 * (rep + 1), "device cache": cache()}), flush=True)
for cfg, t in times.items():
    t = np.array(t[1:])
    print(json.dumps({"scan": cfg + " 100+200", "n": len(t), "median_s": round(float(np.median(t)), 4), "min_s": round(float(t.min()), 4), "max_s": round(float(t.max()), 4)}), flush=True)
