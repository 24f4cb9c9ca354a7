"""MFMA-busy fraction per kernel class from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass of
bench.py: busy = MFMA-busy cycles summed over the SIMDs / (kernel cycles x 1024 SIMDs), kernel cycles =
GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs, MI355X_MICROARCH.md). -> gpurun_out/mfma_busy.json"""
import csv, glob, collections, json
from pmc_traffic import canon
f = glob.glob('gpurun_out/pmc_m/*/*counter_collection.csv')[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    per[(r['Dispatch_Id'], canon(r['Kernel_Name']))][r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (d, k), v in per.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
        a = agg[k]; a[0] += 1; a[1] += v['SQ_VALU_MFMA_BUSY_CYCLES']; a[2] += v['GRBM_GUI_ACTIVE'] / 8.0
out = {}
for k, (n, busy, cyc) in agg.items():
    if busy > 0:
        out[k] = {'launches_profiled': n, 'mfma_busy_frac': round(busy / (cyc * 1024.0), 4),
                  'avg_kernel_cycles': round(cyc / n)}
cls = collections.defaultdict(lambda: [0.0, 0.0])
for k, (n, busy, cyc) in agg.items():
    c = k.split('|')[0]
    cls[c][0] += busy; cls[c][1] += cyc
out['_by_class'] = {c: round(b / (cy * 1024.0), 4) for c, (b, cy) in cls.items() if b > 0}
json.dump(out, open('gpurun_out/mfma_busy.json', 'w'), indent=1)
print(out['_by_class'])
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get('launches_profiled', 0) if isinstance(kv[1], dict) and 'launches_profiled' in kv[1] else 0)[:10]:
    print(k, v)
