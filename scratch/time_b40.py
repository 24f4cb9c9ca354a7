"""where the time of tests/test_bench_path_gpu.py::test_fp32_training_step_b40_256_on_the_multi_tile_walk goes"""
import sys, time, os
t0 = time.time()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib'); sys.path.insert(0, '/root/repo/tests')
import torch
print('import torch %.1f s' % (time.time() - t0)); t0 = time.time()
torch.zeros(1).cuda(); torch.cuda.synchronize()
print('first cuda %.1f s' % (time.time() - t0)); t0 = time.time()
import test_bench_path_gpu as T
from hipnet import synth
from oracle import hrnet_cpu as O
_, sd = T._model('fp32', salt=0)
print('model build + load + cuda %.1f s' % (time.time() - t0)); t0 = time.time()
batch = synth.rhd_batch(40, seed=4321)
print('batch %.1f s' % (time.time() - t0)); t0 = time.time()
ref = T._oracle_step(sd, O.W32_EXTRA, batch, torch.float32)
print('oracle step (cpu threads %d) %.1f s' % (torch.get_num_threads(), time.time() - t0)); t0 = time.time()
model, _ = T._model('fp32', sd)
print('second model %.1f s' % (time.time() - t0)); t0 = time.time()
model.train(); model.zero_grad(set_to_none=True)
hm, inter = model(torch.from_numpy(batch['imgs']).cuda()); torch.cuda.synchronize()
print('first forward (plan build) %.1f s' % (time.time() - t0)); t0 = time.time()
from core.loss import HeatmapLoss
loss = HeatmapLoss()(hm, torch.from_numpy(batch['heatmaps']).cuda()); loss.backward(); torch.cuda.synchronize()
print('backward %.1f s' % (time.time() - t0)); t0 = time.time()
grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}
print('grads to cpu %.1f s' % (time.time() - t0)); t0 = time.time()
cos = T._cos(grads, ref['grads']); errs = T._per_tensor_err(grads, ref['grads'])
print('cos/err %.1f s' % (time.time() - t0)); t0 = time.time()
