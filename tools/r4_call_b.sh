#!/bin/bash
# round 4, GPU call B: the whole -m gpu suite, the probe with a bounded run-ahead, the DEFAULT bench command under rocprofv3
O=gpurun_out/r4_b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== tests"
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log
tail -8 $O/tests.log
echo "== probe, at most 8 / 32 replays in flight, under rocprofv3"
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/pb8 -- tools/graph_wrap_probe 65 300 bound 8 > $O/probe_b8.out 2> $O/probe_b8.err; echo "65x300 bound 8 rc $?" | tee -a $O/probe.txt
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/pb32 -- tools/graph_wrap_probe 65 300 bound 32 > $O/probe_b32.out 2> $O/probe_b32.err; echo "65x300 bound 32 rc $?" | tee -a $O/probe.txt
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/pb64 -- tools/graph_wrap_probe 65 300 bound 64 > $O/probe_b64.out 2> $O/probe_b64.err; echo "65x300 bound 64 rc $?" | tee -a $O/probe.txt
for f in b8 b32 b64; do grep -c "^replay" $O/probe_$f.err | sed "s/^/replays logged $f: /" | tee -a $O/probe.txt; done
rm -rf $O/pb8 $O/pb32 $O/pb64
echo "== the default bench command under rocprofv3 --kernel-trace --stats (no environment switch)"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -- python3 bench.py --no-cpu-baseline > $O/bench_trace_full.json 2> $O/trace_full.err; echo "trace_full rc $?" | tee -a $O/probe.txt
tail -c 400 $O/trace_full.err
find $O/trace_full -name "*kernel_stats.csv" | head
