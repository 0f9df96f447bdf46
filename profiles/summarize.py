#!/usr/bin/env python3
"""Condense a rocprofv3 output tree (profiles/run_profile.sh) into a short text summary + traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in rows("trace/**/*kernel_stats.csv"):
    print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"),
                                                         r.get("AverageNs"), r.get("Percentage")))
dur = defaultdict(list)
meta = {}
for r in rows("trace/**/*kernel_trace.csv"):
    name = r["Kernel_Name"]
    dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                        "Workgroup_Size_X", "Grid_Size_X")}
print("== per-kernel durations from the trace ==")
for k, v in dur.items():
    v = sorted(v)
    print("%-90s n=%d avg_us=%.2f median_us=%.2f min_us=%.2f %s" % (k[:90], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3,
                                                                v[0] / 1e3, meta[k]))
print("== PMC counters (per dispatch average) ==")
acc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for r in rows(sub + "/**/*counter_collection.csv"):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in acc.items():
    print(k[:100])
    for c, v in sorted(cs.items()):
        avg = sum(v) / len(v)
        print("   %-24s n=%d avg=%.6g" % (c, len(v), avg))
        summary.setdefault(k, {})[c] = avg
# HBM traffic: FETCH_SIZE/WRITE_SIZE are in KiB... (rocprofv3 derived metric: bytes/1024); on gfx950 FETCH_SIZE
# reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> double it
# (MI355X_MICROARCH.md, section HBM).
for k, cs in summary.items():
    if "lnprob" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fetch = cs["FETCH_SIZE"] * 1024.0 * 2.0
        write = cs["WRITE_SIZE"] * 1024.0
        print("traffic(%s): fetch %.4g B (x2-corrected) + write %.4g B = %.4g B per launch" % (k[:40], fetch, write, fetch + write))
        n = None
        try:
            n = json.loads(open(os.path.join(out, "bench_fetch.json")).read().strip().splitlines()[-1])["config"]["evals_per_step_per_gpu"]
        except Exception:
            pass
        json.dump({"kernel": k, "n": n, "fetch_bytes_corrected": fetch, "write_bytes": write,
                   "traffic_bytes_per_launch": fetch + write, "raw_FETCH_SIZE_KiB": cs["FETCH_SIZE"],
                   "raw_WRITE_SIZE_KiB": cs["WRITE_SIZE"],
                   "note": "FETCH_SIZE x1024 x2 (gfx950 wide-load correction) + WRITE_SIZE x1024; separate --pmc passes"},
                  open(os.path.join(out, "traffic.json"), "w"), indent=1)
