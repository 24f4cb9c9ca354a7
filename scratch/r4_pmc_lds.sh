#!/bin/bash
# LDS bank-conflict counters of the fused backward launches (scratch/fused_micro.py)
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python scratch/fused_micro.py > gpurun_out/r4/pmc_lds.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_lds/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k, d in acc.items():
    if 'fused' in k:
        print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in d.items()})
PY
