#!/bin/bash
# round 4, GPU call G: k_uni_resolve compiled for three waves per SIMD (168 VGPRs, 352 B of scratch per lane) against two (256, 64 B)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r4_g; mkdir -p $O
for v in base uni3 base uni3; do
  if [ $v = base ]; then unset GOLEMHIP_LIB; else export GOLEMHIP_LIB=$PWD/variants/$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 tools/arb_full_queue.py > $O/$v.txt 2> $O/$v.err
  echo "$v: $(grep k_uni_resolve $O/$v/*/*_kernel_stats.csv | cut -d, -f2-4 | tr -d '"' | sed 's/.*GfUniQueue\*, unsigned int\*)//')"
done
