"""Analyse a rocprofv3 kernel trace of bench.py: GPU-busy union, concurrency, and per kernel class the
time during which it is the ONLY class running (exclusive) for the last step."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
def cls(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.search(r'(conv_kernel|wgrad_kernel|wgrad_reduce|bn_finalize|bn_bwd_reduce|bn_bwd_finalize|grad_term|sum_terms|pack_table|bilinear_cat_bwd|bilinear_cat|adam|colsum|im2col|fill_zero|nhwc|nchw|loss)', n)
    return m.group(1) if m else n[:30]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), cls(r['Kernel_Name']),
       int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))) for r in rows]
ev.sort()
# steps are delimited by adam kernels
adam = [e for e in ev if e[2] == 'adam']
if len(adam) >= 2:
    t0, t1 = adam[-2][1], adam[-1][1]
else:
    t0, t1 = ev[0][0], ev[-1][1]
step = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print('step wall %.2f ms, %d kernels, sum of kernel time %.2f ms' % ((t1 - t0) / 1e6, len(step), sum(e[1] - e[0] for e in step) / 1e6))
pts = []
for s, e, c, wg in step:
    pts.append((s, 1, c, wg)); pts.append((e, -1, c, wg))
pts.sort()
active = collections.Counter(); nact = 0; last = t0
busy = 0; excl = collections.Counter(); conc = collections.Counter(); smallonly = 0; wgs = 0
for t, d, c, wg in pts:
    dt = t - last
    if nact > 0:
        busy += dt
        conc[min(nact, 5)] += dt
        live = [k for k, v in active.items() if v > 0]
        if len(live) == 1:
            excl[live[0]] += dt
        if wgs < 256:
            smallonly += dt
    last = t
    active[c] += d; nact += d; wgs += d * wg
print('busy union %.2f ms, idle %.2f ms, time with < 256 workgroups in flight %.2f ms' % (busy / 1e6, (t1 - t0 - busy) / 1e6, smallonly / 1e6))
print('concurrency histogram (ms):', {k: round(v / 1e6, 2) for k, v in sorted(conc.items())})
print('exclusive time per class (ms):')
for k, v in excl.most_common(14):
    tot = sum(e[1] - e[0] for e in step if e[2] == k)
    n = sum(1 for e in step if e[2] == k)
    print('  %-18s excl %6.2f   total %6.2f   n %4d' % (k, v / 1e6, tot / 1e6, n))
