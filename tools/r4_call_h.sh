#!/bin/bash
# round 4, GPU call H: kernel trace of the C5 sampler (grid kernels) after the shorter x87 primitives
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r4_h; mkdir -p $O
export GF_SAMPLER_CHAIN=0 ROC_AQL_QUEUE_SIZE=131072
rocprofv3 --kernel-trace --stats --output-format csv -d $O/grid -- python3 tools/c5_chain_census.py 100 200 > $O/grid.txt 2> $O/grid.err; echo "grid trace rc $?"
cat $O/grid/*/*_kernel_stats.csv | cut -c1-60,200-400 | head -12
python3 profiles/summarize.py $O/grid 2>/dev/null | grep -E "avg_us" | cut -c1-100,100-400 | head
