import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import bench, torch
from hipnet import _capi as C
model, cfg, sd = bench.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
net = model.hip()
plan = net.plan(64, 256, 256, True, True)
tot = 0; act = 0; rows = {}
for op in plan.bwd.ops:
    if op.kind == C.OP_WGRAD:
        i = op.i
        ns, cout, cin, ks = i[11], i[7], i[4], i[8]
        b = ns * cout * cin * ks * ks * 4
        tot += b
        key = (cout, cin, ks, i[5], ns)
        r = rows.setdefault(key, [0, 0]); r[0] += 1; r[1] += b
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:12]:
    print('Cout %d Cin %d k%d Ho %d nsplit %d: %d convs, slab MB each %.1f, total MB %.0f' % (k[0], k[1], k[2], k[3], k[4], v[0], v[1]/v[0]/1e6, v[1]/1e6))
print('total slab GB per step (written once, read once): %.2f' % (tot/1e9))
