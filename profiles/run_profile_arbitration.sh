#!/bin/bash
# The x87 arbitration on a full queue: kernel trace + two PMC passes of tools/arb_full_queue.py.
#   gpurun -- 'bash profiles/run_profile_arbitration.sh r03'
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_${TAG}_arb
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/arb_full_queue.py > $OUT/run_trace.txt 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/arb_full_queue.py > $OUT/run_sq.txt 2> $OUT/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 tools/arb_full_queue.py > $OUT/run_sq2.txt 2> $OUT/sq2.err
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
