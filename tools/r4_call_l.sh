#!/bin/bash
# round 4, GPU call L: what slows the C5 sampler while its chain is read back by DMA -- kernel durations or the gaps between kernels?
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r4_l; mkdir -p $O
export ROC_AQL_QUEUE_SIZE=131072
rocprofv3 --kernel-trace --stats --output-format csv -d $O/streamed -- python3 tools/arena_scan_ab.py pitched > $O/streamed.txt 2> $O/streamed.err
GF_SCAN_NO_STREAMED_CHAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/after -- python3 tools/arena_scan_ab.py after_the_run > $O/after.txt 2> $O/after.err
for v in streamed after; do echo "== $v"; tail -3 $O/$v.txt; python3 - <<PY
import csv, glob
f = glob.glob("$O/$v/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_stretch" in r["Name"]:
        print(r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
