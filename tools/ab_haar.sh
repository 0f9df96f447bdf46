#!/bin/bash
# A/B of libgolemhip.so variants on k_haar (C3): tools/ab_haar.sh base v1 ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  python tools/bench_haar.py 200 2>&1 | cut -c1-140 | sed "s/^/$v | /"
done; done
