#!/bin/bash
mkdir -p gpurun_out/r4
o=gpurun_out/r4/fused3.txt
python scratch/fused_micro.py > $o 2>&1
HRNET_FUSED_CUS=256 python scratch/fused_micro.py >> $o 2>&1
export HRNET_HIP_LIB=$GRAFT_REPO_ROOT/scratch/var_measure/libhrnet_hip.so
STAMP=1 python scratch/fused_micro.py >> $o 2>&1
grep -v amdgpu.ids $o | grep -v "^ *[2-3][0-9] \|^ *1[7-9] \|^ *[3-8] "
