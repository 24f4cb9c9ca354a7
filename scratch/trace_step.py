"""Timeline of the LAST training step in a rocprofv3 kernel trace of bench.py (scratch/rp2.sh): wall time, union of
busy time, concurrency histogram, per queue busy time / gaps, and per kernel class: launches, summed duration, mean.
usage: python scratch/trace_step.py DIR [--dump]"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)
    n = n.replace('_ZN12_GLOBAL__N_1', '')
    return n[:70]
ev = []
for r in rows:
    wg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), wg, r.get('Queue_Id', '0'), int(r.get('LDS_Block_Size', 0) or 0), int(r.get('VGPR_Count', 0) or 0)))
ev.sort()
adam = [e for e in ev if 'adam' in e[2]]
t0, t1 = adam[-2][1], adam[-1][1]
step = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print('step wall %.3f ms, %d kernels, sum of kernel time %.2f ms' % ((t1 - t0) / 1e6, len(step), sum(e[1] - e[0] for e in step) / 1e6))
pts = []
for e in step:
    pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
n = 0; last = t0; conc = collections.Counter()
for t, d in pts:
    conc[min(n, 6)] += t - last
    last = t; n += d
conc[0] += t1 - last
print('concurrency (ms with k kernels running):', {k: round(v / 1e6, 2) for k, v in sorted(conc.items())})
# per queue
byq = collections.defaultdict(list)
for e in step: byq[e[4]].append(e)
for q, L in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in L)
    gaps = [L[i + 1][0] - L[i][1] for i in range(len(L) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    print('queue %s: %4d kernels busy %.2f ms; gaps < 20 us: n %d mean %.2f us' % (q, len(L), busy / 1e6, len(small), (sum(small) / max(1, len(small))) / 1e3))
cls = collections.OrderedDict()
for e in step:
    c = cls.setdefault(e[2], [0, 0])
    c[0] += 1; c[1] += e[1] - e[0]
print('%-72s %5s %9s %8s' % ('kernel', 'n', 'ms', 'us avg'))
for k, v in sorted(cls.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-72s %5d %9.3f %8.2f' % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
if '--dump' in sys.argv:
    for e in step:
        print('%9.2f %8.2f q%s wg%5d lds%6d vgpr%4d %s' % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[4], e[3], e[5], e[6], e[2]))
