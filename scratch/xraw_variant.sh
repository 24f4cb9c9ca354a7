#!/bin/bash
# variant library: bwd_fused.hip without the raw-input LDS image (HR_FUSED_XRAW=0), other objects as shipped
set -e
cd "$(dirname "$0")/.."
C=hrnet-hand-pose-estimation_amd/csrc
d=scratch/var_xraw0; mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I include -I $C -Wno-unused-result -DHR_FUSED_XRAW=0 -c $C/bwd_fused.hip -o $d/bwd_fused.o
objs=$(ls $C/build/*.o | grep -v bwd_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhrnet_hip.so $objs $d/bwd_fused.o
echo built $d/libhrnet_hip.so
