#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/pmc_w.log 2>&1
python scratch/pmc_traffic.py
