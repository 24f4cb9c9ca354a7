#!/bin/bash
# SQ counters of the DCN launches (bench.py --mode dcn): where the single-pass backward spends its cycles
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  rm -rf gpurun_out/pmc_dcn_sq
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_dcn_sq -- python bench.py --mode dcn --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4/pmc_dcn_sq.log 2>&1
  python - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pmc_dcn_sq/*/*counter_collection.csv')
if not fs:
    print('no counter file'); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name']
    k = 'bwd_fused' if 'dcn_bwd_fused' in k else 'fwd' if 'dcn_fwd' in k else None
    if k is None: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, d in acc.items():
    print(k, {c: round(v / max(n[k][c], 1)) for c, v in d.items()})
PY
done
