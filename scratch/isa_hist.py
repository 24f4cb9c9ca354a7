"""instruction histogram of one kernel in a hipcc -S dump: python isa_hist.py file.s substring"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^_ZN.*:', l) and sys.argv[2] in l]
i0 = start[0]
i1 = next(i for i in range(i0 + 1, len(lines)) if lines[i].startswith('\ts_endpgm'))
ins = [l.strip().split()[0] for l in lines[i0:i1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
c = collections.Counter(ins)
print(lines[i0].split(':')[0], len(ins))
grp = collections.Counter()
for k, v in c.items():
    g = 'mfma' if 'mfma' in k else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_')) else 'other'
    grp[g] += v
print(dict(grp))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print('  %-28s %d' % (k, v))
