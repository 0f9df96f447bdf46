#!/bin/bash
# round 4, GPU call N: a longer fuzz on the round's last binary (the x87 primitives changed this round); seeds 250-260 of the BSM fuzz first
O=gpurun_out/r4_n; mkdir -p $O
GF_FUZZ_SEEDS_BSM=260 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "random_bsm_configurations and (250 or 251 or 252 or 253 or 254 or 255 or 256 or 257 or 258 or 259)" > $O/fuzz_254.log 2>&1; echo "seed 254 rc $?"; tail -2 $O/fuzz_254.log
GF_FUZZ_SEEDS=1600 GF_FUZZ_SEEDS_BSM=1000 GF_FUZZ_SEEDS_SAMPLER=300 GF_FUZZ_SEEDS_MULTI=120 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $O/fuzz.log 2>&1; echo "fuzz rc $?"; tail -4 $O/fuzz.log
