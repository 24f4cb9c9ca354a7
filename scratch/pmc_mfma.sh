#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/pmc_m.log 2>&1
python scratch/pmc_mfma.py
