import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import hip_helpers as hh
from hipnet import _capi as C
dt = torch.bfloat16
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def case(N, H, Cc, ks):
    x = torch.randn(N, H, H, Cc, device='cuda').to(dt)
    w = torch.randn(Cc, Cc, ks, ks) * 0.05
    wp, cop, cip = hh.pack_weights(w, dt, mode=1)
    y = torch.zeros(N, H, H, Cc, device='cuda', dtype=dt)
    yr = torch.randn(N, H, H, Cc, device='cuda').to(dt); om = torch.randn(N, H, H, Cc, device='cuda').to(dt)
    sc = torch.rand(Cc, device='cuda') + 0.5; sh = torch.rand(Cc, device='cuda') - 0.5
    tiles = C.call('hrnet_conv_tiles_bwdstats', N, H, H, Cc, ks, 1)
    st = torch.zeros(tiles, 2, Cc, device='cuda')
    def plain(acc): C.call('hrnet_conv2d', 1, x.data_ptr(), wp.data_ptr(), None, None, None, y.data_ptr(), None, N, H, H, Cc, H, H, Cc, ks, 1, 0, 0, acc, C.stream_ptr())
    def bs(mode, acc): C.call('hrnet_conv2d_bwdstats', 1, x.data_ptr(), wp.data_ptr(), y.data_ptr(), st.data_ptr(), yr.data_ptr(), om.data_ptr() if mode == 2 else None,
                         sc.data_ptr() if mode == 1 else None, sh.data_ptr() if mode == 1 else None, N, H, H, Cc, H, H, Cc, ks, 1, 0, acc, C.stream_ptr())
    t0, t0a = bench(lambda: plain(0)), bench(lambda: plain(1))
    t1, t2, t2a = bench(lambda: bs(1, 0)), bench(lambda: bs(2, 0)), bench(lambda: bs(2, 1))
    print('N%d H%d C%d k%d: dgrad %.1f us, +acc %.1f | bwdstats relu-affine %.1f, sum-mask %.1f, sum-mask+acc %.1f' % (N, H, Cc, ks, t0, t0a, t1, t2, t2a), flush=True)
case(64, 64, 32, 3); case(64, 32, 64, 3); case(64, 16, 128, 3); case(64, 8, 256, 3); case(64, 64, 64, 3)
