import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import os, torch
import hip_helpers as hh
from hipnet import _capi as C
d = 'cuda'
N, H, W, Cin, Cout = 64, 64, 64, 480, 480
x = torch.randn(N, H, W, Cin, device=d).bfloat16(); w = torch.randn(Cout, Cin, device=d).bfloat16(); b = torch.randn(Cout, device=d)
y = torch.empty(N, H, W, Cout, device=d, dtype=torch.bfloat16); sums = torch.zeros(8, 2, Cout, device=d)
def fwd():
    C.call('hrnet_conv2d_bnref', 1, x.data_ptr(), w.data_ptr(), None, None, None, 0.0, 0.0, b.data_ptr(), y.data_ptr(), sums.data_ptr(), N, H, W, Cin, H, W, Cout, 1, 1, 0, C.stream_ptr())
def dg():
    C.call('hrnet_conv2d', 1, x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), None, N, H, W, Cin, H, W, Cout, 1, 1, 0, 0, 0, C.stream_ptr())
for name, f in (('forward+bias+stats', fwd), ('input gradient', dg)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('%s: %.1f us  %.0f TFLOP/s  %.2f TB/s' % (name, us, 2.0 * N * H * W * Cin * Cout / us / 1e6, (x.numel() + y.numel()) * 2 / us / 1e6))
