#!/bin/bash
# usage: build_variants.sh name "flags" [name "flags" ...]  -> scratch/variants/librufus_<name>.so (rfx_msp.hip only differs)
set -e
cd /root/repo/rufus_amd/csrc
mkdir -p /root/repo/scratch/variants
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value $f -c ${SRC:-rfx_msp.hip} -o /tmp/var_$n.o
  objs=""; for o in rfx_kernels.o rfx_p2l.o rfx_msp.o rfx_overlap.o rfx_synth.o rfx_model.o rfx_text.o rfx_api.o rfx_host.o; do
    if [ "$o" = "$(basename ${SRC:-rfx_msp.hip} .hip).o" ]; then objs="$objs /tmp/var_$n.o"; else objs="$objs $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/librufus_$n.so $objs
  echo built $n
done
