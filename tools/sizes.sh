#!/bin/bash
cd $GRAFT_REPO_ROOT
export GOLEMHIP_LIB=$PWD/variants/nt.so
for e in 64 256 1024 4096; do
  python bench.py --no-cpu-baseline --steps 200 --warmup 20 --ensembles $e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ens $e', '%.4g evals/s'%d['value'], '%.1f GB/s'%d['roofline']['achieved'], 'kernel_us %.2f'%(1e3*d['roofline']['kernel_ms']), 'scaled_to_4096 %.1f us'%(1e3*d['roofline']['kernel_ms']*4096/$e))"
done
