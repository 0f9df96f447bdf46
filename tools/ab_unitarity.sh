#!/bin/bash
# A/B of libgolemhip.so variants on the arbitration path: tools/ab_unitarity.sh base v1 v2 ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  python tools/bench_unitarity.py 2>/dev/null | grep -E "band 2" | sed "s/^/$v /" | cut -c1-120
  python tools/scan_c4_twice.py 2>/dev/null | tail -1 | sed "s/^/$v C4 scan /"
done
