import sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch, numpy as np
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
def case(N, H, Cin, Cout, ks, stride=1):
    x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
    w = torch.randn(Cout, Cin, ks, ks) * 0.05
    wp, cop, cip = hh.pack_weights(w, dt)
    sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.rand(Cin, device='cuda') - 0.5
    Ho = (H + 2*(ks//2) - ks)//stride + 1
    y = torch.empty(N, Ho, Ho, cop, device='cuda', dtype=dt)
    tiles = C.call('hrnet_conv_tiles', N, Ho, Ho, cop, ks, stride)
    st = torch.zeros(tiles, 2, cop, device='cuda')
    def run(aff, stats, relu=True):
        C.call('hrnet_conv2d', 1, x.data_ptr(), wp.data_ptr(), sc.data_ptr() if aff else None, sh.data_ptr() if aff else None, None,
               y.data_ptr(), st.data_ptr() if stats else None, N, H, H, Cin, Ho, Ho, cop, ks, stride, 0, 1 if (aff and relu) else 0, 0, C.stream_ptr())
    fl = 2.0*N*Ho*Ho*Cout*Cin*ks*ks
    t_full = bench(lambda: run(True, True)); t_nostat = bench(lambda: run(True, False)); t_noaff = bench(lambda: run(False, False))
    y2 = torch.empty_like(x)
    t_copy = bench(lambda: y2.copy_(x))
    print('N%d H%d Cin%d Cout%d k%d s%d: full %.1f us (%.0f TF)  nostats %.1f  noaffine+nostats %.1f  | copy x->x %.1f us (%.0f MB)' % (
        N, H, Cin, Cout, ks, stride, t_full, fl/t_full/1e6, t_nostat, t_noaff, t_copy, x.numel()*2/1e6))
import os
if os.environ.get('CASES') == 'deep':
    case(64, 16, 128, 128, 3); case(64, 8, 256, 256, 3); case(64, 16, 64, 128, 3); case(64, 32, 64, 64, 3)
    sys.exit(0)
case(64, 64, 32, 32, 3)
case(64, 32, 64, 64, 3)
case(64, 16, 128, 128, 3)
case(64, 8, 256, 256, 3)
case(64, 64, 256, 64, 1)
case(64, 64, 64, 256, 1)
case(64, 64, 480, 480, 1)
case(64, 64, 64, 64, 3)
case(64, 64, 256, 32, 3)
