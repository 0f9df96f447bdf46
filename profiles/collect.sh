#!/bin/bash
# Copy the summaries of gpurun_out/prof_<tag> (run_profile.sh) and gpurun_out/prof_<tag>_bsm (run_profile_bsm.sh) into
# profiles/<tag>/ -- the tracked copies the DESIGN.md numbers cite.  Run here, after the gpurun call has merged gpurun_out/.
set -e
TAG=${1:-r02}
S=gpurun_out/prof_$TAG
B=gpurun_out/prof_${TAG}_bsm
D=profiles/$TAG
mkdir -p $D
cp $(ls -t $S/trace/*/*_kernel_stats.csv | head -1) $D/bench_kernel_stats.csv
cp $(ls -t $S/trace_full/*/*_kernel_stats.csv | head -1) $D/bench_full_kernel_stats.csv
cp $S/bench_trace.json $D/bench_under_rocprof.json
cp $S/bench_trace_full.json $D/bench_full_under_rocprof.json
cp $S/summary.txt $D/bench_summary.txt
cp $S/traffic.json $D/bench_traffic.json
cp $S/traffic.json profiles/traffic.json
python3 tools/trace_by_grid.py $S/trace_full > $D/bench_full_by_grid.txt
cp $(ls -t $B/trace/*/*_kernel_stats.csv | head -1) $D/bsm_kernel_stats.csv
cp $B/summary.txt $D/bsm_counters_summary.txt
cp $B/bsm_instr.json $D/bsm_instr.json
cp $B/bsm_instr.json profiles/bsm_instr.json
cp $B/bench_trace.log $D/bsm_kernel_rates.jsonl
python3 tools/trace_by_grid.py $B/trace > $D/bsm_by_grid.txt
ls -la $D
