#!/bin/bash
# round 4, GPU call F: the shortened x87 primitives (round64 without selects, the 11-operation sum): the dependent-chain probe in both
# forms, the unitarity / sampler / bsm test files, the failing-region and C5 numbers of the bench
O=gpurun_out/r4_f
mkdir -p $O
timeout -k 10 120 tools/x87_int_probe_r3 > $O/probe_r3.txt 2>&1; echo "probe r3 rc $?"
timeout -k 10 120 tools/x87_int_probe_r4 > $O/probe_r4.txt 2>&1; echo "probe r4 rc $?"
tail -4 $O/probe_r3.txt $O/probe_r4.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"
tail -5 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_f/bench.json").read().strip().splitlines()[-1])
print("value %.3e frac %.3f" % (d["value"], d["roofline"]["frac"]))
print("failing region", d["c4_bulk"]["with_status_through_the_failing_region"])
for k in ("c4_scan", "c5_scan", "c4_scan_ref", "c5_scan_ref"):
    print(k, d[k]["seconds"], d[k].get("sampling_s"))
print("c5_sampler", d.get("c5_sampler"))
PY
