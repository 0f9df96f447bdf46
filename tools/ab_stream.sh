#!/bin/bash
# tools/ab_stream.sh base v1 v2 ...: tools/bench_stream.py per variant, 3 interleaved reps
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  python tools/bench_stream.py 2>/dev/null
done; done
