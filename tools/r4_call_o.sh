#!/bin/bash
# round 4, GPU call O: counters of the C5 sampler's two kernels (grid launch shape)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r4_o; mkdir -p $O
export GF_SAMPLER_CHAIN=0 ROC_AQL_QUEUE_SIZE=131072
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 tools/c5_chain_census.py 100 200 > $O/sq.txt 2> $O/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- python3 tools/c5_chain_census.py 100 200 > $O/sq2.txt 2> $O/sq2.err
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "sq2"):
    f = glob.glob("gpurun_out/r4_o/%s/*/*_counter_collection.csv" % d)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:45]
        if "k_stretch" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: "%.3g" % (sum(v) / len(v)) for c, v in cs.items()}, "n", len(next(iter(cs.values()))))
PY
