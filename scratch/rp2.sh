#!/bin/bash
# usage: scratch/rp2.sh <tag> [ENV=VAL ...] : rocprofv3 kernel trace of 3+2 bench steps -> gpurun_out/rp_<tag>, timeline of the last step
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/rp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_$tag -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/rp_$tag.log 2>&1
python scratch/trace_step.py gpurun_out/rp_$tag --dump > gpurun_out/trace_$tag.txt
head -70 gpurun_out/trace_$tag.txt
# keep only the small summaries (the traces are tens of MB)
find gpurun_out/rp_$tag -name '*_kernel_trace.csv' -delete
