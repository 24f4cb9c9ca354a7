#!/bin/bash
# kernel trace of the training step with the full per-kernel dump kept (scratch/cu_time.py reads it)
tag=${1:-base}
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/rp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_$tag -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/rp_$tag.log 2>&1
python scratch/trace_step.py gpurun_out/rp_$tag --dump > gpurun_out/r4/trace_$tag.txt
head -60 gpurun_out/r4/trace_$tag.txt
find gpurun_out/rp_$tag -name '*_kernel_trace.csv' -delete
