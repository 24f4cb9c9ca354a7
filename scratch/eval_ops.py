"""op-by-op timing of the eval forward program (config 2) of the tree given as argv[1]"""
import sys, os, collections
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'hrnet-hand-pose-estimation_amd', 'lib'))
import torch
import bench as B
from hipnet import _capi as C, synth
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
model, cfg, sd = B.build_model(dtype, 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().eval()
b = synth.rhd_batch(64, seed=1)
x = torch.from_numpy(b['imgs']).cuda()
with torch.no_grad():
    for _ in range(3):
        model(x)
net = model.hip()
plan = net.plan(64, 256, 256, False, False)
if isinstance(plan, list): plan = plan[0]
prog = plan.fwd
n = len(prog)
torch.cuda.synchronize()
for rep in range(2):
    evs = []
    for idx in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); prog.run(idx, idx + 1); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
iso = [a.elapsed_time(b) for a, b in evs]
agg = collections.defaultdict(lambda: [0, 0.0])
for idx, t in enumerate(iso):
    op = prog.ops[idx]
    k = int(op.kind)
    key = str(k)
    if k == 1:
        key = 'conv N%d %dx%d %d->%d k%d s%d' % (op.i[1], op.i[5], op.i[6], op.i[4], op.i[7], op.i[8], op.i[9])
    elif k == 5:
        key = 'sum %dx%d C%d n%d' % (op.i[2], op.i[3], op.i[4], op.i[5])
    a = agg[key]; a[0] += 1; a[1] += t
print('total isolated %.3f ms over %d ops' % (sum(iso), n))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print('%-40s %4d  %8.3f ms  %7.1f us' % (k, c, t, t / c * 1e3))
