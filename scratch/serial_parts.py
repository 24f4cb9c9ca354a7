"""isolated time of the lane-0-only parts of the recorded programs (before the first fork / after the last join)"""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
from hipnet import _capi as C, synth
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
b = synth.rhd_batch(64, seed=1)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
for _ in range(2):
    hm, _ = model(x); (hm - gt).square().mean().backward()
net = model.hip()
with torch.no_grad():
    net.pack_weights(for_backward=True)
plan = net.plan(64, 256, 256, True, True)
hm, inter = plan.run_forward(x)
g = torch.empty_like(hm); g.copy_(hm - gt)
net.prepare_grads()
plan.bwd.set_ptr(plan.gout_op, 0, g.data_ptr())
for pname, prog in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    kinds = [int(o.kind) for o in prog.ops]
    ev = [i for i, k in enumerate(kinds) if k in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT)]
    first, last = ev[0], ev[-1]
    times = []
    for idx in range(len(prog)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.run(idx, idx + 1); e1.record(); times.append((e0, e1))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in times]
    names = {1: 'conv', 2: 'wgrad', 4: 'bn_fin', 5: 'sum', 6: 'grad_term', 7: 'bn_red', 8: 'bn_bfin', 9: 'cat', 10: 'cat_bwd', 20: 'wred'}
    def part(lo, hi):
        d = {}
        for i in range(lo, hi):
            d[names.get(kinds[i], str(kinds[i]))] = d.get(names.get(kinds[i], str(kinds[i])), 0.0) + ms[i]
        return round(sum(d.values()), 2), {k: round(v, 2) for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:6]}
    print(pname, 'ops', len(prog), 'before first fork:', part(0, first))
    print(pname, 'after last join:', part(last + 1, len(prog)))
    print(pname, 'all:', part(0, len(prog))[0])
