#!/bin/bash
# measurement builds of dcn.hip with -DDCN_LDS_TAPS=n (taps whose input-gradient scatter uses LDS atomics)
set -e
cd "$(dirname "$0")/.."
C=hrnet-hand-pose-estimation_amd/csrc
for n in "$@"; do
  d=scratch/var_dcnt_$n; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I include -I $C -Wno-unused-result -DDCN_LDS_TAPS=$n -c $C/dcn.hip -o $d/dcn.o &
done
wait
for n in "$@"; do
  d=scratch/var_dcnt_$n
  objs=$(ls $C/build/*.o | grep -v "/dcn.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhrnet_hip.so $objs $d/dcn.o
done
echo built
