import csv, re, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
grp = collections.OrderedDict()
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name']); name = re.sub(r'\(.*', '', name)
    m = re.search(r'(conv_kernel|wgrad_kernel|wgrad_reduce|bn_finalize|bn_bwd_reduce|bn_bwd_finalize|grad_term|sum_terms|pack_table|bilinear|adam|colsum|im2col|copyBuffer)', name)
    key = m.group(1) if m else name[:40]
    g = grp.setdefault(key, [0, 0.0]); g[0] += int(r['Calls']); g[1] += float(r['TotalDurationNs'])
print('per step ms %.2f (5 steps)' % (tot/1e6/5))
for k, v in sorted(grp.items(), key=lambda kv: -kv[1][1]):
    print('%-18s %6d calls/step %8.3f ms/step %7.2f us avg' % (k, v[0]//5, v[1]/1e6/5, v[1]/v[0]/1e3))
print('--- top kernels')
for r in rows[:14]:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name']); name = re.sub(r'\(.*', '', name)[:80]
    print('%5d/step %8.3f ms/step %7.2f us  %s' % (int(r['Calls'])//5, float(r['TotalDurationNs'])/1e6/5, float(r['AverageNs'])/1e3, name))
