"""bench.py's sequence of scans in one process, with the registered arena: c4_scan, c5_scan, c5_sampler, c4_scan_ref, c5_scan_ref, then the
two reference-length scans again -- why is the FIRST c5_scan_ref of a process slow?"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from golemflavor_amd import scan, _lib  # noqa: E402

arena = scan.ResultArena(12582912000)
scan.set_result_arena(arena)
L = _lib.lib()


def show(tag, r):
    print(json.dumps({"scan": tag, "seconds": round(r["seconds"], 4), "setup_s": round(r["setup_s"], 4), "sampling_s": round(r["sampling_s"], 4),
                      "d2h_s": round(r["d2h_s"], 4), "host": (r.get("nonunitary_proposals") or {}).get("host_thread_times")}), flush=True)


seq = [("c4_scan", "C4", 100, 200), ("c5_scan", "C5", 100, 200), ("c5_sampler", None, 0, 0), ("c4_scan_ref", "C4", 200, 1000), ("c5_scan_ref", "C5", 200, 1000),
       ("c4_scan_ref again", "C4", 200, 1000), ("c5_scan_ref again", "C5", 200, 1000), ("c5_scan_ref a third time", "C5", 200, 1000)]
if len(sys.argv) > 1 and sys.argv[1] == "c5first":
    seq = [("c5_scan_ref", "C5", 200, 1000), ("c5_scan_ref again", "C5", 200, 1000)]
for tag, cfg, burn, n in seq:
    if cfg is None:
        r = bench.extra_c5_sampler(0)
        print(json.dumps({"c5_sampler us per half-step": round(r["us_per_half_step_stored_run"], 1)}), flush=True)
        continue
    show(tag, bench.extra_scan(0, cfg, burn, n))
