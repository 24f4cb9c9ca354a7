"""isolated timing of hrnet_conv3x3_bwd_fused at the benchmark shapes (and with HRNET_FUSED_ABLATE phases removed)"""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib'); sys.path.insert(0, '/root/repo/tests')
import torch
from hipnet import _capi as C
d = 'cuda:0'
def run(N, H, W, Cc, iters=30):
    dt = torch.bfloat16
    g = torch.Generator(device=d).manual_seed(1)
    mk = lambda c: torch.randn(N, H, W, c, device=d, generator=g).to(dt)
    dz, y, x, add, bsy = mk(Cc), mk(Cc), mk(Cc), mk(Cc), mk(Cc)
    dx = torch.empty_like(x)
    wT = torch.randn(Cc * 9 * Cc, device=d, generator=g).to(dt)
    coef = torch.rand(3 * Cc, device=d)
    sc, sh = torch.rand(Cc, device=d) + 0.5, torch.rand(Cc, device=d) - 0.5
    ns = C.call('hrnet_bwd_fused_splits', 1, N, H, W, Cc, Cc)
    slabs = torch.empty(ns * Cc * 9 * Cc, device=d)
    rows = torch.empty(ns * 2 * Cc, device=d)
    def go():
        C.call('hrnet_conv3x3_bwd_fused', 1, dz.data_ptr(), y.data_ptr(), coef.data_ptr(), x.data_ptr(), sc.data_ptr(),
               sh.data_ptr(), 1, wT.data_ptr(), dx.data_ptr(), add.data_ptr(), 1, rows.data_ptr(), bsy.data_ptr(),
               slabs.data_ptr(), N, H, W, Cc, Cc, C.stream_ptr())
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    bytes_alg = N * H * W * Cc * 2 * (3 + 1 + 2)       # dz, y, x reads; dx write; addend + bs_y reads
    print('ablate=%s N=%d %dx%d C=%d ns=%d: %.1f us  (%.2f TB/s algorithmic, %.0f TFLOP/s)' % (
        os.environ.get('HRNET_FUSED_ABLATE', '0'), N, H, W, Cc, ns, us, bytes_alg / us / 1e6, 4.0 * N * H * W * Cc * Cc * 9 / us / 1e6))
def stamped(N, H, W, Cc):
    import numpy as np
    buf = torch.zeros(1024 * 32, dtype=torch.int64, device=d)
    os.environ['HRNET_FUSED_STAMP_PTR'] = hex(buf.data_ptr())
    run(N, H, W, Cc, iters=3)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(1024, 32)
    st = st[st[:, 0] != 0]
    nst = int((st[0] != 0).sum())
    dl = np.diff(st[:, :nst], axis=1)      # s_memtime ticks = shader cycles
    names = ['weights', 'coef tables + first loads issued', 'loop entry'] + ['store_tile (waits for the loads)', 'barrier', 'epilogue operands issued', 'dgrad', 'wgrad', 'epilogue', 'barrier2 + loop'] * 8
    print('workgroups', len(st), 'stamps', nst, 'lifetime median %.0f cycles' % float(np.median(st[:, nst - 1] - st[:, 0])))
    for k in range(nst - 1):
        nm = names[k] if k < len(names) else '?'
        print('  %2d %-34s %6.0f cycles' % (k, nm, float(np.median(dl[:, k]))))
if os.environ.get('STAMP'):
    stamped(64, 64, 64, 32)
    stamped(64, 32, 32, 64)
else:
    run(64, 64, 64, 32)
    run(64, 32, 32, 64)
