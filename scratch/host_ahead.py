"""is the host ahead of the GPU? CPU wall-clock at the points of a step vs the GPU's completion time of the same points"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
from hipnet import synth
from hipnet.optim import FlatAdam
from core.loss import HeatmapLoss
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
b = synth.rhd_batch(64, seed=1)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
crit = HeatmapLoss(); opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
def step(rec=None):
    pts = []
    def mark(name):
        if rec is not None:
            e = torch.cuda.Event(enable_timing=True); e.record(); pts.append((name, time.perf_counter(), e))
    mark('start')
    opt.zero_grad(); hm, _ = model(x); mark('fwd enqueued')
    loss = crit(hm, gt); mark('loss enqueued')
    loss.backward(); mark('bwd enqueued')
    opt.step(); mark('opt enqueued')
    if rec is not None: rec.append(pts)
for _ in range(10): step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter(); e0 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(6): step(rec)
torch.cuda.synchronize()
print('6 steps: %.2f ms/step' % ((time.perf_counter() - t0) / 6 * 1e3))
for i, pts in enumerate(rec[2:5]):
    print('step', i + 2)
    for name, tc, e in pts:
        print('   %-14s cpu %8.3f ms   gpu reaches it at %8.3f ms   (gpu - cpu = %7.3f)' % (name, (tc - t0) * 1e3, e0.elapsed_time(e), e0.elapsed_time(e) - (tc - t0) * 1e3))
