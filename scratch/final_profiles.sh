#!/bin/bash
# round-2 measurement batch (one gpurun call): bench lines of every config, rocprofv3 kernel stats, PMC passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_default.log 2>&1 && tail -1 $O/bench_default.log > $O/r02_bench_final.json
echo "default done: $(python -c "import json;d=json.load(open('$O/r02_bench_final.json'));print(d['ms_per_step'], d['value'])")"
python bench.py --mode infer --dtype fp32 --no-cpu-baseline --no-roofline > $O/c2_fp32.log 2>&1; tail -1 $O/c2_fp32.log > $O/c2_fp32.json
python bench.py --mode infer --dtype bf16 --no-cpu-baseline --no-roofline > $O/c2_bf16.log 2>&1; tail -1 $O/c2_bf16.log > $O/c2_bf16.json
python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/c3_fp32.log 2>&1; tail -1 $O/c3_fp32.log > $O/c3_fp32.json
python bench.py --arch w48 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/c4_w48.log 2>&1; tail -1 $O/c4_w48.log > $O/c4_w48.json
python bench.py --mode dcn --no-cpu-baseline > $O/c5_dcn.log 2>&1; tail -1 $O/c5_dcn.log > $O/c5_dcn.json
echo "configs done"
bash scratch/rp.sh r02 > $O/rp_r02.sum 2>&1
echo "rocprof stats done"
bash scratch/pmc_traffic.sh > $O/pmc_traffic.out 2>&1
echo "traffic done"
bash scratch/pmc_mfma.sh > $O/pmc_mfma.out 2>&1
echo "mfma done"
