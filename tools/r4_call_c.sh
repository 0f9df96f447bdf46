#!/bin/bash
# round 4, GPU call C: more of the probe's matrix, and the default bench command under rocprofv3 with a ring that does not wrap
O=gpurun_out/r4_c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { tag=$1; shift; timeout -k 10 120 rocprofv3 --kernel-trace -d $O/p_$tag -- tools/graph_wrap_probe "$@" > $O/probe_$tag.out 2> $O/probe_$tag.err; echo "$tag ($*) rc $? replays logged $(grep -c '^replay' $O/probe_$tag.err)" | tee -a $O/probe.txt; rm -rf $O/p_$tag; }
run event 65 300 event
run copy8 65 300 copy 8
run k64 64 300
run k128 128 150
run k65_long 65 600 bound 8
export ROC_AQL_QUEUE_SIZE=131072
run q131072 65 600
echo "== the default bench command under rocprofv3 --kernel-trace --stats, ROC_AQL_QUEUE_SIZE=131072"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -- python3 bench.py --no-cpu-baseline > $O/bench_trace_full.json 2> $O/trace_full.err; echo "trace_full rc $?" | tee -a $O/probe.txt
unset ROC_AQL_QUEUE_SIZE
tail -c 300 $O/trace_full.err
find $O/trace_full -name "*.csv" | head
