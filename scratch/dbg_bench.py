import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch
import bench
from core.loss import HeatmapLoss
from hipnet import synth
from hipnet.optim import FlatAdam
B = int(sys.argv[1]); dtype = sys.argv[2]
def log(*a):
    print(*a, file=sys.stderr, flush=True)
t = time.time()
model, cfg, sd = bench.build_model(dtype, 'RHD_HRNet_w32_max_hmloss_v1.yaml'); log('model', time.time()-t)
model = model.cuda().train()
b = synth.rhd_batch(B, seed=1234); log('batch', time.time()-t)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
crit = HeatmapLoss(); opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
for i in range(4):
    t0 = time.time()
    opt.zero_grad(); hm, _ = model(x); torch.cuda.synchronize(); t1 = time.time()
    loss = crit(hm, gt); loss.backward(); torch.cuda.synchronize(); t2 = time.time()
    opt.step(); torch.cuda.synchronize(); t3 = time.time()
    log('step %d fwd %.1f ms  bwd %.1f ms  opt %.1f ms  loss %.4f' % (i, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, loss.item()))
log('mem GB', torch.cuda.max_memory_allocated()/1e9)
