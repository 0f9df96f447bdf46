#!/bin/bash
# round 4, GPU call M: the binary with the one-division quotient and the direct square-root remainder: full -m gpu suite, extended fuzz,
# 8 M-walker tier check, the dependent-chain probe, the bench line
O=gpurun_out/r4_m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
GF_FUZZ_SEEDS=400 GF_FUZZ_SEEDS_BSM=250 GF_FUZZ_SEEDS_SAMPLER=100 GF_FUZZ_SEEDS_MULTI=40 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 $O/fuzz.log
timeout -k 10 600 python tools/tier_mismatch.py 1000000 > $O/tier.log 2>&1; echo "tier rc $?"; tail -1 $O/tier.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_m/bench.json").read().strip().splitlines()[-1])
print("value %.3e frac %.3f" % (d["value"], d["roofline"]["frac"]), "failing %.3e" % d["c4_bulk"]["with_status_through_the_failing_region"]["evals_per_s"],
      "c5_sampler us/half-step %.1f, s %.4f" % (d["c5_sampler"]["us_per_half_step_stored_run"], d["c5_sampler"]["seconds"]))
for k in ("c4_scan", "c5_scan", "c4_scan_ref", "c5_scan_ref"):
    print(k, round(d[k]["seconds"], 4), d[k].get("seconds_into_fresh_memory"))
PY
