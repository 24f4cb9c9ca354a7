#!/bin/bash
# experiment builds for DESIGN section 4 trap 4: conv_ring.hip compiled with -DHR_TRAP4=N (SLP on), linked with the
# shipped objects -> scratch/var_trap4_N/libhrnet_hip.so
set -e
cd "$(dirname "$0")/.."
C=hrnet-hand-pose-estimation_amd/csrc
for n in 0 1 2 3; do
  d=scratch/var_trap4_$n; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I include -I $C -Wno-unused-result -DHR_TRAP4=$n -c $C/conv_ring.hip -o $d/conv_ring.o &
done
wait
for n in 0 1 2 3; do
  d=scratch/var_trap4_$n
  objs=$(ls $C/build/*.o | grep -v conv_ring.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhrnet_hip.so $objs $d/conv_ring.o
  echo built $d/libhrnet_hip.so
done
