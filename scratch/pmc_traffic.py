"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel HBM bytes per
launch (profiles/traffic_rNN.json, read back by bench.py for roofline.traffic)."""
import csv, glob, collections, json, re, sys


def canon(name):
    """canonical key 'conv|3|1|8|8|32|2|2|2' from rocprofv3's (sometimes mis-demangled) kernel names"""
    if 'bwd_fused_kernel' in name and 'dcn' not in name:
        if name.startswith('_ZN'):
            ints = re.findall(r'Li(\d+)E', name) + ['true' if 'Lb1E' in name else 'false']
        else:
            body = name[name.index('<') + 1:name.rindex('>')]
            ints = [q.strip() for q in body.split(',')][1:]
        return '|'.join(['bwd_fused'] + ints)
    if 'conv_ring_kernel' in name:
        if name.startswith('_ZN'):
            ints = re.findall(r'Li(\d+)E', name) + ['true' if 'Lb1E' in name else 'false']
        else:
            ints = [q.strip() for q in name[name.index('<') + 1:name.rindex('>')].split(',')]
        return '|'.join(['conv_ring'] + ints)
    if 'bwd_pw_kernel' in name:
        ints = re.findall(r'Li(\d+)E', name) if name.startswith('_ZN') else [q.strip() for q in name[name.index('<') + 1:name.rindex('>')].split(',')]
        return '|'.join(['bwd_pw'] + ints)
    kind = ('conv' if 'conv_kernel' in name else 'conv_bs' if 'conv_bs_kernel' in name else
            'conv_fwd' if 'conv_fwd_kernel' in name else 'conv_fwdb' if 'conv_fwdb_kernel' in name else 'conv_fwds' if 'conv_fwds_kernel' in name else 'conv_dg' if 'conv_dg_kernel' in name else
            'wgrad' if 'wgrad_kernel' in name else None)
    if kind is None:
        m = re.search(r'(\w+_kernel)', name)
        return m.group(1) if m else name[:40]
    if name.startswith('_ZN'):
        ints = re.findall(r'Li(\d+)E', name)
    else:
        body = name[name.index('<') + 1:name.rindex('>')]
        parts = [p.strip() for p in body.split(',')][2:]          # drop dtype garbage
        if parts[:2] == ['ELi', 'E']:
            ints = ['3', '1'] + parts[2:]
        elif parts[:2] == ['E', '1']:
            ints = ['1', '1'] + parts[2:]
        else:
            ints = parts
    return '|'.join([kind] + list(ints))


if __name__ == '__main__':
    out = {}
    for tag, cname in (('pmc_f', 'FETCH_SIZE'), ('pmc_w', 'WRITE_SIZE')):
        f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % tag)[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != cname:
                continue
            a = agg[canon(r['Kernel_Name'])]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
        for k, v in agg.items():
            out.setdefault(k, {})[cname] = [v[0], v[1] / v[0]]
    rows = []
    for k, v in out.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            n = v['FETCH_SIZE'][0]
            # MI355X_MICROARCH.md, HBM: counters are KiB; on gfx950 FETCH_SIZE reports exactly half of a wide
            # coalesced streaming read (16 B/lane) -> doubled; WRITE_SIZE is exact for 16-byte stores
            fetch = v['FETCH_SIZE'][1] * 1024 * 2
            write = v['WRITE_SIZE'][1] * 1024
            rows.append((n * (fetch + write), k, n, fetch, write))
    rows.sort(reverse=True)
    res = {k: {'launches_profiled': n, 'fetch_bytes_per_launch_x2_corrected': round(f), 'write_bytes_per_launch': round(w),
               'hbm_bytes_per_launch': round(f + w)} for _, k, n, f, w in rows}
    json.dump(res, open('gpurun_out/traffic.json', 'w'), indent=1)
    for _, k, n, f, w in rows[:12]:
        print('%5d launches  fetch %8.2f MB  write %8.2f MB per launch  %s' % (n, f / 1e6, w / 1e6, k))
