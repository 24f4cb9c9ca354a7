#!/bin/bash
# same-box A/B of the ring convolution inside the training step
cd /root/repo
for cfg in "HRNET_CONV_RING=0 HRNET_FUSE_SUM=1" "HRNET_CONV_RING=0 HRNET_FUSE_SUM=0" "HRNET_CONV_RING=1 HRNET_FUSE_SUM=0" "HRNET_CONV_RING=1 HRNET_FUSE_SUM=1"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
