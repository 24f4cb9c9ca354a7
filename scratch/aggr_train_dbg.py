import os, sys
os.environ['HRNET_DETERMINISTIC'] = '1'
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch
import test_poseaggr_gpu as T
from hipnet import synth
from oracle import poseaggr_cpu as O
model, cfg, _ = T._model('fp32')
model.train()
b = synth.rhd_batch(5, seed=8, img_h=128, img_w=128)
x = torch.from_numpy(b['imgs']).cuda()
sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
heat, temp = model(x)
R = torch.randn(heat.shape, generator=torch.Generator().manual_seed(2))
(heat * R.cuda()).sum().backward()
head = {k: p for k, p in model.named_parameters() if k.startswith(('offset_feats', 'offsets', 'deform_conv'))}
model.load_state_dict(sd0, strict=True); model.invalidate_weights()
with torch.no_grad():
    logits, _, _ = model.hip().forward(x, training=True, need_grad=False)
sd64 = {k: v.double().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd0.items() if k.startswith(('offset_feats', 'offsets', 'deform_conv'))}
want = O.heatmaps(O.aggregate(logits.double().cpu(), sd64, training=True), 1.7)
(want * R.double()).sum().backward()
rows = []
for k, p in head.items():
    ref = sd64[k].grad; got = p.grad.double().cpu()
    rows.append((k, ((got - ref).norm() / max(ref.norm().item(), 1e-30)).item(), ref.norm().item(), got.norm().item()))
for k, e, rn, gn in rows:
    if k.startswith(('offsets', 'deform')) or k.startswith('offset_feats.19') or k.startswith('offset_feats.0.') or k.startswith('offset_feats.10.conv1'):
        print('%-40s rel err %.3e  |ref| %.3e |got| %.3e' % (k, e, rn, gn))
