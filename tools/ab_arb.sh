#!/bin/bash
# A/B of libgolemhip.so variants (tools/build_variants.sh) on the arbitration path: tools/ab_arb.sh base v1 v2 ...
# per variant: the bulk arbitration (tools/arb_probe.py, first three workloads) and the C5 scan's sampling phase (the settle kernel)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  python tools/arb_probe.py 2>&1 | head -3 | cut -c1-150 | sed "s/^/$v | /"
  python tools/scan_c5_twice.py 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/$v | C5 scan /"
done; done
