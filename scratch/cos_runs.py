import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch
import test_model_gpu as T
from hipnet import synth
from oracle import hrnet_cpu as O
model, _, sd = T.make_model('fp32', 3)
b = synth.rhd_batch(4, seed=99, img_h=128, img_w=128)
r64 = T._run_oracle(sd, O.W32_EXTRA, b, torch.float64)
r32 = T._run_oracle(sd, O.W32_EXTRA, b, torch.float32)
d = [0, 0, 0]
for k, ref in r64['grads'].items():
    g = r32['grads'][k]
    d[0] += float((g * ref).sum()); d[1] += float((g * g).sum()); d[2] += float((ref * ref).sum())
print('oracle fp32 vs fp64 cos', d[0] / np.sqrt(d[1] * d[2]))
for mode in ('0', '0', '0', '1', '1'):
    os.environ['HRNET_DETERMINISTIC'] = mode
    model, _, sd = T.make_model('fp32', 3)
    hm, inter, loss = T._run_hip(model, b)
    e_hip, e_o32, cos = T._grad_errors(model, r64, r32)
    print('det', mode, 'cos', cos, 'med', np.median(e_hip), np.median(e_o32), 'p95', np.percentile(e_hip, 95), np.percentile(e_o32, 95), 'max', e_hip.max())
