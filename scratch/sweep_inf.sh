#!/bin/bash
run() { echo "== infer $1"; env $1 python bench.py --mode infer --dtype bf16 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for cfg in "$@"; do run "$cfg"; done
