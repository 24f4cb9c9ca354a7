#!/bin/bash
# A/B runs inside one box: name=env pairs
run() { echo "== $1"; env $1 python bench.py --steps 30 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for cfg in "$@"; do run "$cfg"; done
