#!/bin/bash
# round 4, GPU call I: extended fuzz + the 8 M-walker tier check on the binary with the shortened x87 primitives
O=gpurun_out/r4_i; mkdir -p $O
GF_FUZZ_SEEDS=400 GF_FUZZ_SEEDS_BSM=250 GF_FUZZ_SEEDS_SAMPLER=100 GF_FUZZ_SEEDS_MULTI=40 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > $O/fuzz.log 2>&1; echo "fuzz rc $?"
tail -3 $O/fuzz.log
timeout -k 10 600 python tools/tier_mismatch.py 1000000 > $O/tier.log 2>&1; echo "tier rc $?"
tail -5 $O/tier.log
