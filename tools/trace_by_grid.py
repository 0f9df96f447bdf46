#!/usr/bin/env python3
"""Per-dispatch durations of a rocprofv3 --kernel-trace CSV grouped by (kernel, grid size): the aggregate --stats table
mixes the launches of different batch sizes.  usage: trace_by_grid.py <dir or kernel_trace.csv>"""
import collections, csv, glob, os, statistics, sys


def short(name):
    for tag in ("k_bsm<", "k_lnprob_sm", "k_stretch", "k_haar"):
        if tag in name:
            i = name.find(tag)
            j = name.find(">", i)
            return name[i:j + 1] if j > 0 else name[i:i + 40]
    i = name.find("k_")
    return name[i:].split("(")[0] if i >= 0 else name[:40]


def main(path):
    if os.path.isdir(path):
        files = glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True)
        path = max(files, key=os.path.getmtime)          # a merged gpurun_out/ keeps older runs: the newest
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), r.get("Scratch_Size", "?"), r.get("VGPR_Count", "?"))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("%-34s %10s %8s %5s %6s %10s %10s" % ("kernel", "grid", "scratch", "vgpr", "calls", "median_us", "min_us"))
    for k, v in sorted(agg.items()):
        print("%-34s %10d %8s %5s %6d %10.1f %10.1f" % (k[0], k[1], k[2], k[3], len(v), statistics.median(v) / 1e3, min(v) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
