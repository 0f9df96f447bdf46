#!/bin/bash
O=gpurun_out/r4_d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_fuzz.py tests/test_gpu_mcmc.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log
tail -6 $O/tests.log
timeout -k 10 600 python tools/c5_sampler_ab.py 2 > $O/c5_ab.jsonl 2> $O/c5_ab.err; echo "ab rc $?"
cat $O/c5_ab.jsonl
tail -3 $O/c5_ab.err
