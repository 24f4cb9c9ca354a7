import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
import hip_helpers as hh
from hipnet import _capi as C
d = 'cuda'
P = 64 * 64 * 64
for Cin, Cout, aff, add, rows in ((64, 256, True, False, True), (256, 64, False, True, True), (64, 64, True, True, True)):
    dz = torch.randn(P, Cout, device=d).bfloat16(); y = torch.randn(P, Cout, device=d).bfloat16()
    x = torch.randn(P, Cin, device=d).bfloat16(); addend = torch.randn(P, Cin, device=d).bfloat16(); bsy = torch.randn(P, Cin, device=d).bfloat16()
    wT = torch.randn(Cin, Cout, device=d).bfloat16(); coef = torch.randn(3 * Cout, device=d); sc = torch.rand(Cin, device=d); sh = torch.rand(Cin, device=d)
    dx = torch.empty(P, Cin, device=d, dtype=torch.bfloat16)
    ns = C.call('hrnet_bwd_pw_splits', 1, P, Cin, Cout)
    slabs = torch.empty(ns, Cout, Cin, device=d); rw = torch.empty(ns, 2, Cin, device=d)
    def run():
        C.call('hrnet_conv1x1_bwd_fused', 1, dz.data_ptr(), y.data_ptr(), coef.data_ptr(), x.data_ptr(), sc.data_ptr() if aff else None,
               sh.data_ptr() if aff else None, 1 if aff else 0, wT.data_ptr(), dx.data_ptr(), addend.data_ptr() if add else None, 1,
               rw.data_ptr() if rows else None, bsy.data_ptr() if rows else None, slabs.data_ptr(), P, Cin, Cout, C.stream_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = P * 2 * (2 * Cout + Cin + Cin + (Cin if add else 0) + (Cin if rows else 0)) + ns * Cout * Cin * 4
    print('Cin %d Cout %d ns %d: %.1f us, %.0f MB -> %.2f TB/s' % (Cin, Cout, ns, us, nbytes / 1e6, nbytes / us / 1e6))
