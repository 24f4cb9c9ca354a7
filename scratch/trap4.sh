#!/bin/bash
# runs scratch/trap4_run.py against the four experiment builds (scratch/trap4_build.sh) and the shipped library
mkdir -p gpurun_out/r4
o=gpurun_out/r4/trap4.txt; : > $o
for n in 0 1 2 3; do
  HRNET_HIP_LIB=$GRAFT_REPO_ROOT/scratch/var_trap4_$n/libhrnet_hip.so timeout -k 5 200 python scratch/trap4_run.py ${1:-400} 2>&1 | grep -v amdgpu.ids >> $o
done
timeout -k 5 200 python scratch/trap4_run.py ${1:-400} 2>&1 | grep -v amdgpu.ids >> $o
cat $o
