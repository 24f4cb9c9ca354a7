import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
from hipnet import synth
from core.loss import HeatmapLoss
from hipnet.optim import FlatAdam
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
b = synth.rhd_batch(64, seed=1)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
crit = HeatmapLoss(); opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
def step(timing=None):
    t0 = time.perf_counter()
    opt.zero_grad()
    hm, _ = model(x)
    t1 = time.perf_counter()
    loss = crit(hm, gt)
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, t3 - t2))
for _ in range(5): step()
torch.cuda.synchronize()
T = []
t0 = time.perf_counter()
for _ in range(10):
    step(T)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
import numpy as np
T = np.array(T) * 1e3
print('host enqueue per step: fwd %.2f ms, loss+bwd %.2f ms, opt %.2f ms; host total %.2f ms/step, gpu-complete %.2f ms/step' % (T[:, 0].mean(), T[:, 1].mean(), T[:, 2].mean(), th / 10 * 1e3, tt / 10 * 1e3))
