#!/bin/bash
# Profile bench.py's kernel on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/run_profile.sh r01'
# Pass 1: kernel trace + stats.  Passes 2..4: PMC counters, each in its own run (never combined with
# trace domains other than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampler --no-extras"
# the trace pass runs bench.py's default step counts, so that its per-kernel average is the number bench.py prints
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-sampler --no-extras > $OUT/bench_trace.json 2> $OUT/trace.err
# ... and one trace of the whole DEFAULT command (sub-records included): every kernel of every BASELINE configuration, C5's chain read
# back while it is sampled, graph replays, no GF_* switch.  ROC_AQL_QUEUE_SIZE=131072 (a variable of the HIP runtime): the command
# dispatches ~19 000 kernels, HIP's default AQL ring holds 16 384 packets, and under rocprofv3's queue interception a hipGraphLaunch
# batch that straddles the wrap of the ring is read past its end (SIGSEGV in the profiling stack; stand-alone reproducer
# tools/graph_wrap_probe.hip, record profiles/r04/rocprof_graph_wrap.txt).  With an 8 MiB ring the run never wraps.
ROC_AQL_QUEUE_SIZE=131072 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace_full.json 2> $OUT/trace_full.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
find $OUT -name "*.csv" | head -40
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
