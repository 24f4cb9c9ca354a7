"""does a FIFTH active stream (RCCL's, at N>1) hurt the 4-lane step? emulate it at N=1: a background torch stream that
runs a few element-wise kernels over 4-28 MB ranges of a buffer while the backward pass runs"""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
from hipnet import synth
from hipnet.optim import FlatAdam
from core.loss import HeatmapLoss
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
b = synth.rhd_batch(64, seed=1234)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
crit = HeatmapLoss(); opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
bg = torch.cuda.Stream()
buf = torch.zeros(30_000_000, device='cuda')
mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
def step():
    opt.zero_grad()
    hm, _ = model(x)
    loss = crit(hm, gt)
    if mode == 'during':
        # background work enqueued right before the backward pass: runs beside it
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(bg):
            bg.wait_event(ev)
            for k in range(8):
                buf[k * 1_000_000:(k + 1) * 1_000_000 + 3_000_000].mul_(1.0001)
    loss.backward()
    if mode == 'after':
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(bg):
            bg.wait_event(ev)
            for k in range(8):
                buf[k * 1_000_000:(k + 1) * 1_000_000 + 3_000_000].mul_(1.0001)
    torch.cuda.current_stream().wait_stream(bg)
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(mode, 'ms/step', (time.perf_counter() - t0) / 20 * 1e3)
