#!/bin/bash
# round 4, GPU call K: the registered-destination read-back with the device cache (no hipFree of large buffers -> no wipe on the DMA engine):
# the runtime's DMA (default) / the copy kernel / without the cache
O=gpurun_out/r4_k; mkdir -p $O
for v in "dma_cache:" "kernel_cache:GF_D2H_KERNEL=1" "dma_nocache:GF_DEVICE_CACHE_GB=0"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name ($envs)"
  env $envs timeout -k 10 200 python tools/c4_post_alone.py 2>&1 | tail -3
  env $envs timeout -k 10 200 python tools/arena_scan_ab.py pitched 2>&1 | tail -5
  env $envs GF_SCAN_NO_STREAMED_CHAIN=1 timeout -k 10 200 python tools/arena_scan_ab.py after_the_run 2>&1 | tail -3
done > $O/ab.txt 2>&1
cat $O/ab.txt
