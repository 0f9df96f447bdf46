#!/bin/bash
# A/B bench of kernel variants built by tools/build_variants.sh: tools/ab.sh base v1 v2 ...  (3 interleaved reps)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-sampler --steps 300 --warmup 10 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '%.4g evals/s'%d['value'], '%.1f GB/s'%d['roofline']['achieved'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])"
done; done
