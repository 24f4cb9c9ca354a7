#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r03
mkdir -p $O
python bench.py --mode infer --dtype fp32 --no-cpu-baseline --no-roofline > $O/c2_fp32.log 2>&1; tail -1 $O/c2_fp32.log > $O/c2_fp32.json
python bench.py --mode infer --dtype bf16 --no-cpu-baseline --no-roofline > $O/c2_bf16.log 2>&1; tail -1 $O/c2_bf16.log > $O/c2_bf16.json
python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-selfcheck > $O/c3_fp32.log 2>&1; tail -1 $O/c3_fp32.log > $O/c3_fp32.json
python bench.py --arch w48 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-selfcheck > $O/c4_w48.log 2>&1; tail -1 $O/c4_w48.log > $O/c4_w48.json
python bench.py --mode dcn --no-cpu-baseline > $O/c5_dcn.log 2>&1; tail -1 $O/c5_dcn.log > $O/c5_dcn.json
echo done
