"""wall-clock per network phase UNDER the real multi-stream overlap: the recorded programs are run in segments cut at
the points where every lane has joined lane 0; events on lane 0 bracket each segment. Next to it the isolated sum of
the same ops (one op at a time)."""
import sys, os, re
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
from hipnet import _capi as C, synth
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
b = synth.rhd_batch(64, seed=1)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
for _ in range(3):
    hm, _ = model(x); (hm - gt).square().mean().backward()
net = model.hip()
with torch.no_grad():
    net.pack_weights(for_backward=True)
plan = net.plan(64, 256, 256, True, True)
hm = torch.empty((64, plan.nj, 64, 64), device='cuda'); inter = torch.empty((64, 32, 64, 64), device='cuda')
plan.fwd.set_ptr(plan.in_op, 0, x.data_ptr()); plan.fwd.set_ptr(plan.out_op, 1, hm.data_ptr()); plan.fwd.set_ptr(plan.inter_op, 1, inter.data_ptr())
g = torch.randn_like(hm) * 1e-3
net.prepare_grads()
plan.bwd.set_ptr(plan.gout_op, 0, g.data_ptr())
streams = plan._side_streams()

def phase_of(name):
    if name is None: return None
    m = re.match(r'(stage\d\.\d|layer1|transition\d|last_layer|conv\d)', name)
    return m.group(1) if m else name

for pname, prog in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    n = len(prog)
    kinds = [int(o.kind) for o in prog.ops]
    lanes = [int(o.i[C.LANE_SLOT]) for o in prog.ops]
    # cut points: after a cluster of event ops that ends with STREAM_WAITs on lane 0 (a join), and before a fork cluster
    cuts = [0]
    i = 0
    while i < n:
        if kinds[i] in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT):
            j = i
            while j < n and kinds[j] in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT): j += 1
            waits0 = [k for k in range(i, j) if kinds[k] == C.OP_STREAM_WAIT and lanes[k] == 0]
            if waits0: cuts.append(j)       # join: everything before is complete when lane 0 passes
            i = j
        else:
            i += 1
    cuts.append(n)
    cuts = sorted(set(cuts))
    for rep in range(2):
        evs = []
        torch.cuda.synchronize()
        for lo, hi in zip(cuts, cuts[1:]):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); prog.run(lo, hi, streams=streams); e1.record()
            evs.append((lo, hi, e0, e1))
        torch.cuda.synchronize()
    wall = [(lo, hi, e0.elapsed_time(e1)) for lo, hi, e0, e1 in evs]
    # isolated per-op times
    iso = []
    for idx in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); prog.run(idx, idx + 1); e1.record(); iso.append((e0, e1))
    torch.cuda.synchronize()
    iso = [a.elapsed_time(b) for a, b in iso]
    tot_w = tot_i = 0.0
    agg = {}
    for lo, hi, w in wall:
        tags = [prog.tags.get(k) for k in range(lo, hi) if prog.tags.get(k)]
        ph = phase_of(tags[0]) if tags else 'misc'
        a = agg.setdefault(ph, [0.0, 0.0, 0.0, 0])
        lane0 = sum(iso[k] for k in range(lo, hi) if lanes[k] == 0 and kinds[k] not in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT))
        a[0] += w; a[1] += sum(iso[lo:hi]); a[2] += lane0; a[3] += hi - lo
        tot_w += w; tot_i += sum(iso[lo:hi])
    print('== %s: wall under overlap %.2f ms, isolated sum %.2f ms, %d ops, %d segments' % (pname, tot_w, tot_i, n, len(wall)))
    for ph, (w, i_, l0, c) in agg.items():
        print('   %-12s wall %6.2f ms   isolated sum %6.2f   lane-0 isolated %6.2f   ops %4d' % (ph, w, i_, l0, c))
    # detail of the lane-0-only segments (first and last): one line per op
    names = {1: 'conv', 2: 'wgrad', 4: 'bn_fin', 5: 'sum', 6: 'grad_term', 7: 'bn_red', 8: 'bn_bfin', 9: 'cat', 10: 'cat_bwd', 11: 'im2col', 12: 'to_nchw', 13: 'to_nhwc', 15: 'bias_grad', 20: 'wred', 21: 'fused'}
    for (lo, hi, w) in (wall[0], wall[-1]) if pname == 'fwd' else (wall[0], wall[-1]):
        print('   -- segment ops [%d,%d) wall %.2f ms' % (lo, hi, w))
        cur = None
        for k in range(lo, hi):
            if prog.tags.get(k): cur = prog.tags[k]
            if kinds[k] in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT): continue
            op = prog.ops[k]
            shape = ''
            if kinds[k] == 1: shape = 'N%d %dx%d %d->%d k%d s%d' % (op.i[1], op.i[2], op.i[3], op.i[4], op.i[7], op.i[8], op.i[9])
            if kinds[k] == 2: shape = '%dx%d %d->%d k%d' % (op.i[2], op.i[3], op.i[4], op.i[7], op.i[8])
            if kinds[k] in (5, 6, 7): shape = '%dx%d C%d' % (op.i[2], op.i[3], op.i[4])
            print('      %4d %-10s %-28s %-24s %7.1f us' % (k, names.get(kinds[k], str(kinds[k])), cur or '', shape, iso[k] * 1e3))
    if pname == 'bwd':
        # one stage-4 and one stage-3 module: per-lane isolated chains and their op mix
        for (lo, hi, w) in (wall[3], wall[7]):
            print('   -- module segment [%d,%d) wall %.2f ms' % (lo, hi, w))
            per = {}
            for k in range(lo, hi):
                if kinds[k] in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT): continue
                d = per.setdefault(lanes[k], {})
                nm = names.get(kinds[k], str(kinds[k]))
                e = d.setdefault(nm, [0, 0.0]); e[0] += 1; e[1] += iso[k] * 1e3
            for ln in sorted(per):
                tot = sum(v[1] for v in per[ln].values()); cnt = sum(v[0] for v in per[ln].values())
                print('      lane %d: %3d ops, isolated %7.1f us: %s' % (ln, cnt, tot, {k: (v[0], round(v[1])) for k, v in sorted(per[ln].items(), key=lambda kv: -kv[1][1])}))
