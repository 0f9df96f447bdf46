#!/bin/bash
# tools/build_variants.sh "name:flags" ... -> /root/repo/variants/<name>.so (A/B experiments, not shipped)
set -e
cd "$(dirname "$0")/../golemflavor_amd/csrc"
mkdir -p ../../variants
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  for f in gf_kernels gf_bsm gf_unitarity gf_capi gf_comm gf_sampler gf_devcache; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $f.hip -o /tmp/var_$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$name.so /tmp/var_gf_kernels.o /tmp/var_gf_bsm.o /tmp/var_gf_unitarity.o /tmp/var_gf_capi.o /tmp/var_gf_comm.o /tmp/var_gf_sampler.o /tmp/var_gf_devcache.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo built $name
done
