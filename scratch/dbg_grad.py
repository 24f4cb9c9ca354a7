import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch
from test_model_gpu import make_model, oracle_state
from hipnet import synth
from oracle import hrnet_cpu as O
from core.loss import HeatmapLoss
B, HW = int(sys.argv[1]), int(sys.argv[2])
model, _, sd = make_model('fp32', 3)
b = synth.rhd_batch(B, seed=99, img_h=HW, img_w=HW)
x = torch.from_numpy(b['imgs']); gt = torch.from_numpy(b['heatmaps'])
def run_oracle(dtype):
    osd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pk = [k for k in osd if not k.endswith(('running_mean','running_var','num_batches_tracked'))]
    for k in pk: osd[k].requires_grad_(True)
    hm, inter, _ = O.hrnet_forward(osd, O.W32_EXTRA, x.to(dtype), training=True)
    loss = O.heatmap_loss(hm, gt.to(dtype)); loss.backward()
    return hm.detach(), {k: osd[k].grad.double() for k in pk}, loss.item()
hm32, g32, l32 = run_oracle(torch.float32)
hm64, g64, l64 = run_oracle(torch.float64)
model.train()
hm, inter = model(x.cuda()); loss = HeatmapLoss()(hm, gt.cuda()); loss.backward()
named = dict(model.named_parameters())
print('loss hip %.8f o32 %.8f o64 %.8f' % (loss.item(), l32, l64))
print('hm err hip-vs-64 %.3e  o32-vs-64 %.3e' % ((hm.cpu().double()-hm64).abs().max(), (hm32.double()-hm64).abs().max()))
rows = []
for k in g64:
    ref = g64[k]; sc = max(ref.abs().max().item(), 1e-12)
    eh = (named[k].grad.cpu().double() - ref).abs().max().item() / sc
    eo = (g32[k] - ref).abs().max().item() / sc
    rows.append((eh, eo, sc, k))
rows.sort(reverse=True)
for r in rows[:25]: print('hip %.2e  o32 %.2e  scale %.2e  %s' % r)
eh = np.array([r[0] for r in rows]); eo = np.array([r[1] for r in rows])
print('median hip %.2e o32 %.2e ; max hip %.2e o32 %.2e; n(hip>10*o32+1e-5)=%d' % (np.median(eh), np.median(eo), eh.max(), eo.max(), int((eh > 10*eo + 1e-5).sum())))
print('---- in network order (tail) ----')
order = [k for k in g64]
d = {r[3]: r for r in rows}
for k in order:
    if k.startswith(('stage4.2', 'last_layer', 'stage4.1.fuse')):
        r = d[k]
        print('hip %.2e  o32 %.2e  scale %.2e  %s' % r)
