import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch, torch.nn.functional as F
from test_model_gpu import make_model
from hipnet import synth
from oracle import hrnet_cpu as O
B, HW = int(sys.argv[1]), int(sys.argv[2])
model, _, sd = make_model('fp32', 3)
x = torch.from_numpy(synth.rhd_batch(B, seed=99, img_h=HW, img_w=HW)['imgs'])
rec = {}
orig = O._bn
def bn_rec(P, t, prefix):
    rec[prefix] = (t.detach().double().mean((0,2,3)), t.detach().double().var((0,2,3), unbiased=False), t.detach().float().var((0,2,3), unbiased=False).double())
    return orig(P, t, prefix)
O._bn = bn_rec
osd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
with torch.no_grad():
    O.hrnet_forward(osd, O.W32_EXTRA, x.double(), training=True)
model.train()
with torch.no_grad():
    pass
hm, _ = model(x.cuda())
net = model.hip()
rows = []
for name, b in net.bns.items():
    m64, v64, v32 = rec[name]
    inv64 = 1/torch.sqrt(v64 + 1e-5)
    e_inv = ((b.invstd.cpu().double() - inv64).abs() / inv64).max().item()
    e_mean = ((b.mean.cpu().double() - m64).abs() / (v64.sqrt() + 1e-12)).max().item()
    ratio = (m64.abs() / v64.sqrt()).max().item()
    rows.append((e_inv, e_mean, ratio, name))
for r in rows[:12] + sorted(rows, reverse=True)[:12]:
    print('invstd relerr %.2e  mean err/std %.2e  max|mean|/std %.1f  %s' % r)
print('median invstd err %.2e' % np.median([r[0] for r in rows]))
