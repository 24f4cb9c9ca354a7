#!/bin/bash
# fused backward: isolated timing, phase stamps (measurement build)
mkdir -p gpurun_out/r4
export HRNET_HIP_LIB=$GRAFT_REPO_ROOT/scratch/var_measure/libhrnet_hip.so
o=gpurun_out/r4/fused_micro.txt
STAMP=1 python scratch/fused_micro.py > $o 2>&1
HRNET_FUSED_CUS=256 STAMP=1 python scratch/fused_micro.py >> $o 2>&1
for a in 1 2 3 4 8; do HRNET_FUSED_ABLATE=$a python scratch/fused_micro.py >> $o 2>&1; done
grep -v amdgpu.ids $o
