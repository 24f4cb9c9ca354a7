import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch
import test_poseaggr_gpu as T
from hipnet import synth
from oracle import poseaggr_cpu as O
model, cfg, sd = T._model('fp32')
b = synth.rhd_batch(10, seed=3, img_h=128, img_w=128)
x = torch.from_numpy(b['imgs']).cuda()
with torch.no_grad():
    net = model.hip()
    logits, _, plan = net.forward(x, training=False, need_grad=False)
    print('logits', logits.abs().max().item(), torch.isfinite(logits).all().item())
    agg = model._aggregate(logits.contiguous(), net, plan)
    print('agg hip', agg.abs().max().item(), torch.isfinite(agg).all().item())
sd64 = {k: v.double() for k, v in sd.items() if k.startswith(('offset_feats', 'offsets', 'deform_conv'))}
lg = logits.double().cpu()
B = 2
ref = lg[2*B:3*B].repeat(5,1,1,1)
f = O.offset_feats(ref - lg, sd64)
print('oracle feats', f.abs().max().item(), torch.isfinite(f).all().item())
want = O.aggregate(lg, sd64)
print('agg oracle', want.abs().max().item(), torch.isfinite(want).all().item())
print('agg err', (agg.double().cpu() - want).abs().max().item())
