#!/bin/bash
# A/B of libgolemhip.so variants (tools/build_variants.sh) on the BSM kernel: tools/ab_bsm.sh base v1 v2 ... (2 interleaved reps)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  python tools/bench_bsm.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d['n'] >= 131072: print('$v', d['case'], d['n'], 'status' if d['status'] else 'plain ', '%.4f ms %.3fe9/s'%(d['kernel_ms'], d['evals_per_s']/1e9))"
done; done
