#!/bin/bash
# usage: scratch/rp.sh <tag> : rocprofv3 kernel trace of 5 bench steps, summary printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/rp_$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_$1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/rp_$1.log 2>&1
python scratch/rpsum.py gpurun_out/rp_$1
