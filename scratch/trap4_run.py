"""Experiment for DESIGN section 4, trap 4 (wrong backward-statistics rows out of the LDS-ring input-gradient launches
when conv_ring.hip is compiled with SLP vectorisation): the same backward-statistics launch N times in one process,
rows compared element by element with the per-element majority value (a launch is deterministic: one row per pixel
walk, no atomics). Prints how many launches deviated and where (row, sum index, channel -> lane group, parity).
usage: HRNET_HIP_LIB=scratch/var_trap4_K/libhrnet_hip.so python scratch/trap4_run.py [launches]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'hrnet-hand-pose-estimation_amd', 'lib'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, torch
import hip_helpers as hh
from hipnet import _capi as C
from hipnet._capi import HrOp
DT = torch.bfloat16
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 400
print('library:', os.environ.get('HRNET_HIP_LIB', '(shipped)'))
for (N, H, W, Cc, masked) in ((64, 16, 16, 128, 'affine'), (64, 8, 8, 256, 'mask')):
    g = torch.Generator().manual_seed(1)
    q = lambda t: t.to(DT).float()
    dy = q(torch.randn(N, Cc, H, W, generator=g)); w = q(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(Cc * 9))
    bs_y = q(torch.randn(N, Cc, H, W, generator=g)); bs_m = q(torch.randn(N, Cc, H, W, generator=g))
    bsc, bsh = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    wp, _, _ = hh.pack_weights(w, DT, mode=1)
    dyd, byd, bmd = hh.nhwc(dy, DT), hh.nhwc(bs_y, DT), hh.nhwc(bs_m, DT)
    bscd, bshd = bsc.to(hh.DEV), bsh.to(hh.DEV)
    C.call('hrnet_conv_ring_enable', 1)
    assert C.call('hrnet_conv_ring_supported', 1, N, H, W, Cc, Cc) >= 3
    nrows = C.call('hrnet_conv_rows_bwdstats', 1, N, H, W, Cc, Cc, 3, 1)
    y = torch.zeros(N, H, W, Cc, dtype=DT, device=hh.DEV)
    allrows = torch.empty(n_launch, nrows, 2, Cc, dtype=torch.float32, device=hh.DEV)
    # a second stream keeps the chip busy with unrelated copies (the failures were seen inside the training step)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=hh.DEV)
    for it in range(n_launch):
        op = HrOp(); op.kind = C.OP_CONV
        for k, val in enumerate((1, N, H, W, Cc, H, W, Cc, 3, 1, 0, 0, 0, 0, 0)): op.i[k] = val
        rows = allrows[it]
        ptrs = [(0, dyd), (1, wp), (5, y), (6, rows), (7, byd)]
        ptrs += [(8, bmd)] if masked == 'mask' else [(9, bscd), (10, bshd)]
        for k, t in ptrs: op.p[k] = C.ptr(t)
        if it % 3 == 0:
            with torch.cuda.stream(side): junk.add_(1)
        C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
    torch.cuda.synchronize()
    r = allrows.cpu().numpy().view(np.uint32)
    maj = np.apply_along_axis(lambda v: np.bincount(np.unique(v, return_inverse=True)[1]).argmax(), 0, r) if False else None
    ref = np.median(r.astype(np.float64), axis=0)          # bit patterns: the majority value is the median when < half deviate
    bad = r.astype(np.float64) != ref[None]
    nbad_launch = int(bad.reshape(n_launch, -1).any(1).sum())
    print('case N{} {}x{} C{} {}: rows {}, launches with a deviating element: {} of {}; deviating elements {}'.format(
        N, H, W, Cc, masked, nrows, nbad_launch, n_launch, int(bad.sum())))
    if bad.any():
        idx = np.argwhere(bad)
        ch = idx[:, 3]
        print('   which sum (0: sum dz, 1: sum dz*y):', np.bincount(idx[:, 2], minlength=2).tolist())
        print('   channel & 7 histogram (position inside a lane\'s 8 channels):', np.bincount(ch & 7, minlength=8).tolist())
        print('   (channel >> 3) & 3 histogram (lane group lg):', np.bincount((ch >> 3) & 3, minlength=4).tolist())
        print('   first deviations (launch, row, which, channel, got, majority):')
        for a in idx[:8]:
            got = allrows[a[0], a[1], a[2], a[3]].item()
            print('     ', a.tolist(), got, np.array([ref[a[1], a[2], a[3]]]).astype(np.uint32).view(np.float32)[0])
