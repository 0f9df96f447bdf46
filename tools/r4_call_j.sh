#!/bin/bash
# round 4, GPU call J: registered result arena (ABI 5) -- the read-back test, the N > 1 bench line on one GPU, the bench line
O=gpurun_out/r4_j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_r3.py tests/test_gpu_bench_contract.py tests/test_gpu_mcmc.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"
tail -5 $O/pytest.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_j/bench.json").read().strip().splitlines()[-1])
print("value %.3e frac %.3f" % (d["value"], d["roofline"]["frac"]))
print("arena", d.get("result_arena"))
for k in ("c4_scan", "c5_scan", "c4_scan_ref", "c5_scan_ref"):
    s = d[k]
    print(k, round(s["seconds"], 4), "sampling", round(s.get("sampling_s", 0), 4), "d2h_s", round(s.get("d2h_s", 0), 4), s.get("destination"), "fresh:", s.get("seconds_into_fresh_memory"))
PY
