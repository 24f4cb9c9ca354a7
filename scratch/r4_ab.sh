#!/bin/bash
# usage: r4_ab.sh "ENV=VAL ENV2=VAL" "..." : bench ms/step per environment
mkdir -p gpurun_out/r4
for e in "$@"; do
  r=$(env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-selfcheck 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$e -> $r ms/step" | tee -a gpurun_out/r4/ab.txt
done
