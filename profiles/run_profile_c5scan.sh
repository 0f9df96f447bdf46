set -u
OUT=gpurun_out/prof_r03_c5scan
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 -m golemflavor_amd.scan --config C5 > $OUT/run_trace.txt 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 -m golemflavor_amd.scan --config C5 > $OUT/run_sq.txt 2> $OUT/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 -m golemflavor_amd.scan --config C5 > $OUT/run_sq2.txt 2> $OUT/sq2.err
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
grep -A20 "k_stretch_multi<12" $OUT/summary.txt | head -60
