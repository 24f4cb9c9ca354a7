#!/bin/bash
mkdir -p gpurun_out/r4
o=gpurun_out/r4/fused4.txt
for v in 0 1 2; do echo "variant $v" >> $o; HRNET_FUSED_VARIANT=$v python scratch/fused_micro.py >> $o 2>&1; HRNET_FUSED_VARIANT=$v HRNET_FUSED_CUS=256 python scratch/fused_micro.py >> $o 2>&1; done
grep -v amdgpu.ids $o
