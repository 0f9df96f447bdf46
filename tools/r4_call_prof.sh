#!/bin/bash
# round 4: the profile passes behind profiles/r04 (run_profile.sh, run_profile_bsm.sh) + traces of the C5 sampler in both launch shapes
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
bash profiles/run_profile.sh r04 > gpurun_out/prof_r04.log 2>&1; echo "run_profile rc $?"
bash profiles/run_profile_bsm.sh r04_bsm > gpurun_out/prof_r04_bsm.log 2>&1; echo "run_profile_bsm rc $?"
O=gpurun_out/prof_r04_sampler; mkdir -p $O
export GF_SAMPLER_CHAIN=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -- python3 tools/c5_chain_census.py 100 200 > $O/chain.txt 2> $O/chain.err; echo "chain trace rc $?"
export GF_SAMPLER_CHAIN=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/grid -- python3 tools/c5_chain_census.py 100 200 > $O/grid.txt 2> $O/grid.err; echo "grid trace rc $?"
unset GF_SAMPLER_CHAIN
ls gpurun_out/prof_r04 gpurun_out/prof_r04_bsm $O
