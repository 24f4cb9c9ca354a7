#!/bin/bash
# measurement builds of dcn.hip (-DDCN_ABLATE=1/2/4) linked with the shipped objects -> scratch/var_dcn_N/
set -e
cd "$(dirname "$0")/.."
C=hrnet-hand-pose-estimation_amd/csrc
for n in 1 2 4 7; do
  d=scratch/var_dcn_$n; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I include -I $C -Wno-unused-result -DDCN_ABLATE=$n -c $C/dcn.hip -o $d/dcn.o &
done
wait
for n in 1 2 4 7; do
  d=scratch/var_dcn_$n
  objs=$(ls $C/build/*.o | grep -v "/dcn.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhrnet_hip.so $objs $d/dcn.o
done
echo built
