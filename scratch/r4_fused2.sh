#!/bin/bash
mkdir -p gpurun_out/r4
o=gpurun_out/r4/fused2.txt
timeout -k 10 400 python -m pytest tests/test_bwd_fused_gpu.py -x -q > $o 2>&1
tail -5 $o
python scratch/fused_micro.py >> $o 2>&1
HRNET_FUSED_CUS=256 python scratch/fused_micro.py >> $o 2>&1
tail -6 $o | grep -v amdgpu.ids
bash scratch/r4_pmc_lds.sh
