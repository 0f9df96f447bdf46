#!/usr/bin/env python3
"""Total GPU time per kernel (and grid) of a rocprofv3 --kernel-trace CSV.  usage: trace_totals.py <dir or csv> [t0_fraction]
(t0_fraction: ignore dispatches that start in the first part of the trace's time span, e.g. 0.5 = second half only)"""
import collections, csv, glob, os, sys
from trace_by_grid import short

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if os.path.isdir(path):
    path = max(glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(path)))
t_lo = min(int(r["Start_Timestamp"]) for r in rows); t_hi = max(int(r["End_Timestamp"]) for r in rows)
cut = t_lo + frac * (t_hi - t_lo)
agg = collections.defaultdict(lambda: [0, 0])
first = last = None
for r in rows:
    if int(r["Start_Timestamp"]) < cut:
        continue
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]))
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    first = int(r["Start_Timestamp"]) if first is None else min(first, int(r["Start_Timestamp"]))
    last = int(r["End_Timestamp"]) if last is None else max(last, int(r["End_Timestamp"]))
tot = sum(v[1] for v in agg.values())
print("span %.1f ms, kernels busy %.1f ms" % ((last - first) / 1e6, tot / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-36s grid %9d calls %6d total %9.2f ms" % (k[0], k[1], v[0], v[1] / 1e6))
