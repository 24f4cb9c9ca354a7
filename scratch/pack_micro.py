import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import bench as B
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
net = model.hip()
for prog, name in ((net.pack_f, 'forward layouts'), (net.pack_d, 'input-gradient layouts')):
    for _ in range(3): prog.run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): prog.run()
    e1.record(); torch.cuda.synchronize()
    print(name, '%.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
