#!/bin/bash
# same-box A/B of the other configurations: round-1 tree (scratch/r01) vs this tree
one() { (cd $1 && python bench.py "${@:2}" --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"); }
for cfg in "--mode infer --dtype bf16" "--mode infer --dtype fp32" "--arch w48 --batch 32 --steps 10 --warmup 3" "--dtype fp32 --steps 10 --warmup 3"; do
  echo "== $cfg"; echo -n "r01: "; one scratch/r01 $cfg; echo -n "now: "; one . $cfg
done
