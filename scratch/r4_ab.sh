#!/bin/bash
# usage: r4_ab.sh "ENV=VAL ENV2=VAL" "..." : bench ms/step per environment (appends to gpurun_out/r4/ab.txt)
mkdir -p gpurun_out/r4
export HRNET_MEASURE=1    # the switches swept here are measurement switches (engine._knob, csrc hr_knob)
for e in "$@"; do
  env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-selfcheck > gpurun_out/r4/ab_last.json 2> gpurun_out/r4/ab_last.err
  r=$(tail -1 gpurun_out/r4/ab_last.json | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || (tail -3 gpurun_out/r4/ab_last.err | tr '\n' ' '))
  echo "$e -> $r ms/step" | tee -a gpurun_out/r4/ab.txt
done
