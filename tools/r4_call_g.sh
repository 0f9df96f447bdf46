#!/bin/bash
# round 4, GPU call G: A/B of one kernel's occupancy through a variant library: $1 = variant name under variants/ (base = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r4_g; mkdir -p $O
V=${1:-uni3}
for v in base $V base $V; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 tools/arb_full_queue.py > $O/$v.txt 2> $O/$v.err
done
for d in $O/base $O/$V; do for f in $d/*/*_kernel_stats.csv; do echo "$f: $(grep -E "k_bsm_tier2|k_uni_resolve" $f | sed 's/.*)",//' | cut -d, -f1-3 | tr '\n' ' ')"; done; done
