#!/bin/bash
mkdir -p gpurun_out/r4
o=gpurun_out/r4/wg_$1.txt
python scratch/wgrad_r4.py > $o 2>&1
grep -v amdgpu.ids $o
timeout -k 10 500 python -m pytest tests/test_wgrad_atomic_gpu.py tests/test_kernels_walk_gpu.py tests/test_kernels_gpu.py -x -q -k "wgrad or weight" 2>&1 | tail -3
