#!/bin/bash
# same-box A/B of environment settings inside the training step: scratch/ab_env.sh "A=1 B=2" "A=0" ...
cd /root/repo
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], 'loss', d.get('final_loss'), 'slab MB', d.get('wgrad_slab_mb_per_step'))"
done
