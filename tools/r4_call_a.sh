#!/bin/bash
# round 4, GPU call A: the -m gpu suite, the profiler / graph-wrap probe, a bench line
O=gpurun_out/r4_a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( df -h /dev/shm; nproc; python3 -c "import os;print('affinity', len(os.sched_getaffinity(0)))"; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/shmem_enabled; free -g ) > $O/box.txt 2>&1
echo "== tests"; 
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log
tail -5 $O/tests.log
echo "== probe without the profiler"
timeout -k 10 120 tools/graph_wrap_probe 65 300 > $O/probe_plain.out 2> $O/probe_plain.err; echo "plain rc $?" | tee -a $O/probe.txt
echo "== probe under rocprofv3"
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/p200 -- tools/graph_wrap_probe 65 200 > $O/probe_200.out 2> $O/probe_200.err; echo "65x200 rc $?" | tee -a $O/probe.txt
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/p300 -- tools/graph_wrap_probe 65 300 > $O/probe_300.out 2> $O/probe_300.err; echo "65x300 rc $?" | tee -a $O/probe.txt
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/p300s -- tools/graph_wrap_probe 65 300 sync > $O/probe_300s.out 2> $O/probe_300s.err; echo "65x300 sync rc $?" | tee -a $O/probe.txt
export ROC_AQL_QUEUE_SIZE=65536
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/p300q -- tools/graph_wrap_probe 65 300 > $O/probe_300q.out 2> $O/probe_300q.err; echo "65x300 ROC_AQL_QUEUE_SIZE=65536 rc $?" | tee -a $O/probe.txt
unset ROC_AQL_QUEUE_SIZE
for f in 200 300 300s 300q; do grep -c "^replay" $O/probe_$f.err | sed "s/^/replays logged $f: /" | tee -a $O/probe.txt; done
rm -rf $O/p200 $O/p300 $O/p300s $O/p300q
echo "== bench"
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 600 $O/bench.err
