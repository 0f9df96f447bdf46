#!/bin/bash
# Static resources of every kernel of libgolemhip.so as compiled (the code object's own metadata: registers, scratch, LDS):
#   tools/kernel_resources.sh > profiles/rNN/kernel_resources.txt
# No GPU needed.  scratch = .private_segment_fixed_size (bytes per lane), lds = .group_segment_fixed_size (static bytes per block).
set -e
cd "$(dirname "$0")/../golemflavor_amd/csrc"
T=$(mktemp -d)
printf "%-110s %6s %6s %8s %8s %7s\n" kernel vgpr sgpr scratch lds spills
for f in gf_kernels gf_bsm gf_unitarity gf_sampler gf_capi; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c $f.hip -o $T/$f.o -save-temps=obj 2>/dev/null
  S=$T/$f-hip-amdgcn-amd-amdhsa-gfx950.s
  [ -f $S ] || continue
  awk '/^  - \.agpr_count/{blk=1} /\.group_segment_fixed_size:/{lds=$2} /\.name:/{name=$2} /\.private_segment_fixed_size:/{scr=$2} /\.sgpr_count:/{sg=$2} /\.vgpr_count:/{vg=$2} /\.vgpr_spill_count:/{sp=$2; printf "%s %s %s %s %s %s\n", name, vg, sg, scr, lds, sp}' $S | while read n vg sg scr lds sp; do
    printf "%-110s %6s %6s %8s %8s %7s\n" "$(echo $n | c++filt | cut -c1-110)" $vg $sg $scr $lds $sp
  done
done
rm -rf $T
