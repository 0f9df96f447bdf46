"""Destination memory of the big read-backs, with the pages REALLY mapped (round 4; the round-3 "hugepage" figures were taken with a
gf_host_prepare that mapped nothing).  One process, one box: device-to-host copies of the C4 / C5 result sizes through the library's
pinned ring (gf_memcpy_d2h) into
   fresh     np.empty, never touched              huge      mmap + madvise(MADV_HUGEPAGE), never touched
   reused    the same array a second time         pinned    the ring's own slots only (no host copy): what the link delivers
with AnonHugePages before / after, and then the scans themselves (bench.extra_scan: c4_scan_ref / c5_scan_ref) with and without huge
pages and with a process-lifetime arena.      python tools/readback_ab.py [--quick]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from golemflavor_amd import _lib  # noqa: E402
from golemflavor_amd import configs as Cf  # noqa: E402
from golemflavor_amd.descriptor import compile_model  # noqa: E402
from golemflavor_amd.model import Model, anon_huge_bytes, empty_hugepages  # noqa: E402

quick = "--quick" in sys.argv
L = _lib.lib()
m = Model(compile_model(Cf.unitary_paramset(), "PRIOR_ONLY", source_ratio=(1, 2, 0)), device=0)


def d2h(dst, d_src):
    t0 = time.perf_counter()
    _lib.check(L.gf_memcpy_d2h(m._h, dst.ctypes.data_as(C.c_void_p), d_src.ptr, dst.nbytes), "d2h")
    return time.perf_counter() - t0


for nbytes in ((1_887_436_800,) if quick else (1_887_436_800, 9_437_184_000, 12_582_912_000)):
    d = m.alloc(nbytes)
    n = nbytes // 8
    for kind in ("fresh", "huge", "fresh", "huge"):
        h0 = anon_huge_bytes()
        a = np.empty(n) if kind == "fresh" else empty_hugepages((n,))
        t1 = d2h(a, d)
        h1 = anon_huge_bytes()
        t2 = d2h(a, d)                                   # the same array again: every page is mapped
        print(json.dumps({"bytes": nbytes, "destination": kind, "first_copy_s": round(t1, 4), "first_GBps": round(nbytes / t1 / 1e9, 1),
                          "second_copy_s": round(t2, 4), "reused_GBps": round(nbytes / t2 / 1e9, 1),
                          "AnonHugePages_before": h0, "AnonHugePages_after": h1}), flush=True)
        del a
    d.free()

# what the link itself delivers in this run: device -> pinned slot, no host copy behind it (hipMemcpy into a pinned destination)
fn = getattr(L, "gf_internal_pinned_d2h_rate", None)
if fn is not None:
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_double)]
    rate = C.c_double(0.0)
    if fn(0, 2 << 30, C.byref(rate)) == 0:
        print(json.dumps({"pinned_to_pinned_GBps": round(rate.value, 1), "bytes": 2 << 30}), flush=True)
m.close()

# the scans themselves
for env, label in (({}, "huge pages (default)"), ({"GF_SCAN_NO_HUGEPAGES": "1"}, "plain np.empty"), ({"GF_SCAN_ARENA": "1"}, "arena, first use"),
                   ({"GF_SCAN_ARENA": "1"}, "arena, reused")):
    for k in ("GF_SCAN_NO_HUGEPAGES", "GF_SCAN_ARENA"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for cfg in ("C4", "C5"):
        h0 = anon_huge_bytes()
        r = bench.extra_scan(0, cfg, 100 if quick else bench.REF_BURNIN, 200 if quick else bench.REF_NSTEPS)
        print(json.dumps({"scan": cfg, "destination": label, "seconds": round(r["seconds"], 4), "sampling_s": round(r["sampling_s"], 4),
                          "d2h_s": round(r["d2h_s"], 4), "bytes": r["chain_bytes_to_host"], "AnonHugePages_before": h0,
                          "AnonHugePages_after": anon_huge_bytes()}), flush=True)
