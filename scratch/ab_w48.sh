#!/bin/bash
cd /root/repo
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --arch w48 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-selfcheck 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
