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

# BSM kernel: VALU wave-instructions per walker from SQ_INSTS_VALU of the largest dispatches (tools/bench_bsm.py runs
# n = 4 194 304 walkers: 65 536 wave-tiles) -> profiles/bsm_instr.json, which bench.py turns into a fraction of the fp64
# issue rate.  One file per run directory; copy it to profiles/ to make it the committed constant.
big = defaultdict(lambda: defaultdict(list))
for r in rows("pmc_sq/**/*counter_collection.csv"):
    if "k_bsm<" in r["Kernel_Name"] and r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_WAVES"):
        big[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
# effective clock of those dispatches: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / wall time of the same dispatch
# (MI355X_MICROARCH.md, "DVFS give-back": the chip lowers its clock under load)
clk = defaultdict(list)
for r in rows("pmc_sq2/**/*counter_collection.csv"):
    if "k_bsm<" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and int(r["Grid_Size"]) == 524288:
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if ns > 0:
            clk[r["Kernel_Name"]].append(float(r["Counter_Value"]) / 8.0 / ns)          # cycles per ns = GHz
instr = {}
N_BIG = 4 * 1024 * 1024
for (k, grid), cs in big.items():
    if grid != 524288 or "SQ_INSTS_VALU" not in cs:
        continue
    import re
    m = re.search(r"k_bsm<(\d+), (true|false), (\d), 1>", k)
    if not m:
        continue
    valu = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"])
    # status variants run in pieces: a dispatch covers N_BIG walkers only when SQ_INSTS_VALU says so; use waves x tiles
    key = "%s_%s" % (m.group(1), "no_status" if m.group(3) == "0" else "with_status")
    if m.group(3) == "0":
        instr[key] = {"kernel": k[:70], "valu_wave_instr_per_launch": valu, "walkers_per_launch": N_BIG,
                      "valu_wave_instr_per_walker": valu / (N_BIG / 64.0), "valu_instr_per_bin_incl_prologue": valu / (N_BIG / 64.0) / 20.0}
        if clk.get(k):
            v = sorted(clk[k])
            instr[key]["effective_clock_ghz"] = v[len(v) // 2]
if instr:
    instr["note"] = ("rocprofv3 --pmc SQ_INSTS_VALU, tools/bench_bsm.py, dispatches of 4 194 304 walkers (grid 524288), 20 energy bins; "
                     "effective_clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch wall time (median over the dispatches)")
    json.dump(instr, open(os.path.join(out, "bsm_instr.json"), "w"), indent=1)
    print("bsm_instr:", json.dumps(instr))
