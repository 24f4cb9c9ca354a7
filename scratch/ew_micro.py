import sys
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
for (N, H, Cc) in [(64, 64, 32), (64, 32, 64), (64, 16, 128), (64, 8, 256), (64, 64, 256), (64, 64, 480)]:
    d = 'cuda'
    g = torch.randn(N, H, H, Cc, device=d).to(dt); out = torch.randn(N, H, H, Cc, device=d).to(dt); y = torch.randn(N, H, H, Cc, device=d).to(dt)
    dst = torch.empty_like(g); dst2 = torch.zeros_like(g)
    sc = torch.rand(Cc, device=d) + 0.5; sh = torch.rand(Cc, device=d) - 0.5; coef = torch.rand(3 * Cc, device=d)
    blocks = C.call('hrnet_reduce_blocks', N, H, H, Cc)
    part = torch.empty(blocks, 2, Cc, device=d)
    mean = torch.zeros(Cc, device=d); inv = torch.ones(Cc, device=d); gam = torch.ones(Cc, device=d); dg = torch.zeros(Cc, device=d); db = torch.zeros(Cc, device=d)
    t_red = bench(lambda: C.call('hrnet_bn_bwd_reduce', 1, part.data_ptr(), g.data_ptr(), out.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), N, H, H, Cc, 0, 0, C.stream_ptr()))
    t_fin = bench(lambda: C.call('hrnet_bn_bwd_finalize', part.data_ptr(), blocks, Cc, float(N*H*H), gam.data_ptr(), mean.data_ptr(), inv.data_ptr(), dg.data_ptr(), db.data_ptr(), coef.data_ptr(), 0, C.stream_ptr()))
    t_app = bench(lambda: C.call('hrnet_grad_term2', 1, dst.data_ptr(), dst2.data_ptr(), g.data_ptr(), out.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), coef.data_ptr(), N, H, H, Cc, 0, 1, C.stream_ptr()))
    mb = g.numel() * 2 / 1e6
    ts = [g, out]; scs = [None, None]
    t_sum = bench(lambda: C.call('hrnet_sum_terms', 1, dst.data_ptr(), N, H, H, Cc, 2, hh.ptr_array([y, g]), hh.ptr_array([sc, None]), hh.ptr_array([sh, None]), hh.int_array([0, 0]), hh.int_array([0, 0]), 1, C.stream_ptr()))
    print('N%d H%d C%d (%.1f MB/tensor): reduce %.1f us (%.2f TB/s)  finalize %.1f us  apply2 %.1f us (%.2f TB/s)  sum2 %.1f us (%.2f TB/s)' % (
        N, H, Cc, mb, t_red, 3*mb/t_red/1e0/1e6*1e6/1e6, t_fin, t_app, 6*mb/t_app, t_sum, 3*mb/t_sum))
