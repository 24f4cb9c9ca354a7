"""isolated timing of the deferred weight-gradient launches of the training step: slab form against the float-atomic form,
several split counts (the atomic form through a one-op recorded program, as the step launches it)"""
import sys, ctypes
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch
from hipnet import _capi as C
dt = torch.bfloat16
def bench(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def case(N, H, Cin, Cout, ks, stride=1, splits=(None,)):
    x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
    Ho = (H + 2*(ks//2) - ks)//stride + 1
    dy = torch.randn(N, Ho, Ho, Cout, device='cuda').to(dt)
    sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.rand(Cin, device='cuda') - 0.5
    fl = 2.0*N*Ho*Ho*Cout*Cin*ks*ks
    out = []
    ns0 = C.call('hrnet_wgrad_splits', 1, N, Ho, Ho, Cout, Cin, ks, stride)
    for ns in splits:
        ns = ns or ns0
        slabs = torch.empty(ns, Cout, ks*ks, Cin, device='cuda')
        grad = torch.zeros(Cout, Cin, ks, ks, device='cuda')
        def w(): C.call('hrnet_conv2d_wgrad', 1, x.data_ptr(), dy.data_ptr(), sc.data_ptr(), sh.data_ptr(), slabs.data_ptr(), N, H, H, Cin, Ho, Ho, Cout, ks, stride, 1, ns, C.stream_ptr())
        op = C.HrOp(); op.kind = C.OP_WGRAD
        for k, v in enumerate((1, N, H, H, Cin, Ho, Ho, Cout, ks, stride, 1, ns, 1, Cout, Cin)): op.i[k] = v
        for k, t in enumerate((x, dy, sc, sh, grad)): op.p[k] = t.data_ptr()
        def wa(): C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
        tw, ta = bench(w), bench(wa)
        out.append('ns=%d%s: slab %.1f atomic %.1f us (%.0f TF)' % (ns, '*' if ns == ns0 else '', tw, ta, fl/ta/1e6))
    print('N%d H%d Cin%d Cout%d k%d s%d: ' % (N, H, Cin, Cout, ks, stride) + ' | '.join(out), flush=True)
case(64, 16, 128, 128, 3, splits=(None, 32, 16, 8))
case(64, 8, 256, 256, 3, splits=(None, 32, 16, 8))
case(64, 32, 32, 64, 3, stride=2, splits=(None, 32, 16))
case(64, 16, 64, 128, 3, stride=2, splits=(None, 32, 16))
case(64, 16, 128, 256, 3, stride=2, splits=(None, 32, 16))
case(64, 32, 64, 32, 1, splits=(None, 64))
case(64, 16, 128, 64, 1, splits=(None, 64, 32))
case(64, 64, 256, 64, 1, splits=(None,))
