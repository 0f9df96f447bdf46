#!/bin/bash
# rocprofv3 passes for the BSM kernel (tools/bench_bsm.py): kernel trace + SQ counters.
set -u
TAG=${1:-r01_bsm}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_bsm.py > $OUT/bench_trace.log 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/bench_bsm.py > $OUT/bench_sq.log 2> $OUT/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- python3 tools/bench_bsm.py > $OUT/bench_sq2.log 2> $OUT/sq2.err
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
