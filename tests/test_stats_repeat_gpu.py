"""Regression for DESIGN section 4, trap 4: the backward-statistics rows (sum dz, sum dz*y per workgroup - no atomics,
one row per pixel walk) of the launches that gather them must come out bit-identical when the SAME launch is repeated.
With hipcc 7.2's SLP-packed accumulate (`v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,0,1]`) the LDS-ring input-gradient
launch deviated in about every second launch (only sum dz*y, only even channels of lanes 48-63); the accumulate is a
scalar inline-asm FMA now (csrc/common.h hr_fma_acc). Every kernel family with such an epilogue is repeated here, with
a second stream keeping the chip busy. Reference ops: the reduction half of autograd's NativeBatchNormBackward behind
lib/models/pose_hrnet.py:41-57,78-98."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = torch.bfloat16
REPEAT = 160


def _h():
    import hip_helpers as hh
    return hh


def _C():
    from hipnet import _capi as C
    return C


def _q(t):
    return t.to(DT).float()


def _repeat(launch, outs, n=REPEAT):
    """run `launch(i)` n times (outs(i) -> tensors it wrote); returns how many runs differ from the majority, bit for bit"""
    side = torch.cuda.Stream()
    junk = torch.empty(32 << 20, dtype=torch.uint8, device='cuda:0')
    for i in range(n):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk.add_(1)
        launch(i)
    torch.cuda.synchronize()
    bad = 0
    detail = None
    for k in range(len(outs(0))):
        r = torch.stack([outs(i)[k].reshape(-1).view(torch.int32) for i in range(n)]).cpu().numpy()
        ref = np.median(r.astype(np.float64), axis=0)
        dev = (r.astype(np.float64) != ref[None])
        if dev.any() and detail is None:
            detail = (k, np.argwhere(dev)[:5].tolist())
        bad = max(bad, int(dev.any(1).sum()))
    return bad, detail


@pytest.mark.parametrize('case', [(64, 16, 16, 128, 'affine', 1), (64, 8, 8, 256, 'mask', 1), (16, 32, 32, 64, 'affine', 0)])
def test_input_gradient_launch_with_backward_statistics_is_repeatable(case):
    """hrnet_conv2d_bwdstats through the LDS ring (ring = 1: the launches that failed) and through the tile-walking body"""
    hh, C = _h(), _C()
    from hipnet._capi import HrOp
    N, H, W, Cc, masked, ring = case
    g = torch.Generator().manual_seed(1)
    dy = _q(torch.randn(N, Cc, H, W, generator=g))
    w = _q(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(Cc * 9))
    bs_y, bs_m = _q(torch.randn(N, Cc, H, W, generator=g)), _q(torch.randn(N, Cc, H, W, generator=g))
    bsc, bsh = (torch.rand(Cc, generator=g) + 0.5).to(hh.DEV), (torch.rand(Cc, generator=g) - 0.5).to(hh.DEV)
    wp, _, _ = hh.pack_weights(w, DT, mode=1)
    dyd, byd, bmd = hh.nhwc(dy, DT), hh.nhwc(bs_y, DT), hh.nhwc(bs_m, DT)
    prev = C.call('hrnet_conv_ring_enable', 1 if ring else 0)
    try:
        if ring:
            assert C.call('hrnet_conv_ring_supported', 1, N, H, W, Cc, Cc) >= 3
        nrows = C.call('hrnet_conv_rows_bwdstats', 1, N, H, W, Cc, Cc, 3, 1)
        y = torch.zeros(N, H, W, Cc, dtype=DT, device=hh.DEV)
        rows = torch.empty(REPEAT, nrows, 2, Cc, dtype=torch.float32, device=hh.DEV)

        def launch(i):
            op = HrOp()
            op.kind = C.OP_CONV
            for k, val in enumerate((1, N, H, W, Cc, H, W, Cc, 3, 1, 0, 0, 0, 0, 0)):
                op.i[k] = val
            ptrs = [(0, dyd), (1, wp), (5, y), (6, rows[i]), (7, byd)]
            ptrs += [(8, bmd)] if masked == 'mask' else [(9, bsc), (10, bsh)]
            for k, t in ptrs:
                op.p[k] = C.ptr(t)
            C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
        bad, detail = _repeat(launch, lambda i: (rows[i],))
    finally:
        C.call('hrnet_conv_ring_enable', 1 if prev != 0 else 0)
    assert bad == 0, (bad, detail)


@pytest.mark.parametrize('case', [(64, 64, 64, 32, 32), (64, 32, 32, 64, 64), (8, 32, 32, 64, 64)])
def test_fused_3x3_backward_is_repeatable(case):
    """hrnet_conv3x3_bwd_fused (slab form: nothing atomic): dx, the statistics rows and the slabs, launch after launch -
    with the statistics operand a tensor of its own and (bs_y == x) taken from the staged input image"""
    hh, C = _h(), _C()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(9 + Cin + N)
    x = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
    dz = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
    y = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
    bsy = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
    coef = torch.randn(3 * Cout, generator=g).to(hh.DEV) * 0.5
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(hh.DEV), (torch.rand(Cin, generator=g) - 0.5).to(hh.DEV)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    wT, _, _ = hh.pack_weights(w, DT, mode=1)
    ns = C.call('hrnet_bwd_fused_splits', 1, N, H, W, Cin, Cout)
    n = 60
    for own_x in (False, True):
        dx = torch.empty(n, N, H, W, Cin, dtype=DT, device=hh.DEV)
        rows = torch.empty(n, ns, 2, Cin, device=hh.DEV)
        slabs = torch.empty(2, ns, Cout, 9, Cin, device=hh.DEV)

        def launch(i):
            op = C.HrOp()
            op.kind = C.OP_BWD_FUSED
            for k, v in enumerate((1, N, H, W, Cin, Cout, 1, 1, 0, Cout, Cin)):
                op.i[k] = v
            for k, t in enumerate((dz, y, coef, x, sc, sh, wT, dx[i], None, rows[i], x if own_x else bsy, slabs[i % 2])):
                op.p[k] = C.ptr(t)
            C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
        bad, detail = _repeat(launch, lambda i: (rows[i],), n=n)
        assert bad == 0, (own_x, bad, detail)
        assert torch.equal(slabs[0], slabs[1])
        assert all(torch.equal(dx[i].view(torch.int16), dx[0].view(torch.int16)) for i in range(1, n))


def test_fused_3x3_backward_statistics_from_the_staged_input_equal_the_separate_operand():
    """bs_y == x: the launch reads the next BatchNorm's raw input out of its own LDS image instead of fetching the tensor
    a second time - rows, dx and slabs must be the bits a launch with a COPY of x as the statistics operand produces"""
    hh, C = _h(), _C()
    for (N, H, W, Cin, Cout, relu) in ((6, 32, 32, 32, 32, 1), (5, 20, 24, 64, 64, 1), (3, 16, 16, 32, 32, 0), (4, 24, 20, 128, 128, 1)):
        if not C.call('hrnet_bwd_fused_supported', 1, Cin, Cout):
            continue
        g = torch.Generator().manual_seed(3 + Cin + N)
        x = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
        dz = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
        y = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
        coef = torch.randn(3 * Cout, generator=g).to(hh.DEV) * 0.5
        sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(hh.DEV), (torch.rand(Cin, generator=g) - 0.5).to(hh.DEV)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
        wT, _, _ = hh.pack_weights(w, DT, mode=1)
        ns = C.call('hrnet_bwd_fused_splits', 1, N, H, W, Cin, Cout)
        outs = []
        for bs in (x, x.clone()):
            dx = torch.full((N, H, W, Cin), float('nan'), dtype=DT, device=hh.DEV)
            rows = torch.full((ns, 2, Cin), float('nan'), device=hh.DEV)
            slabs = torch.zeros(ns, Cout, 9, Cin, device=hh.DEV)
            op = C.HrOp()
            op.kind = C.OP_BWD_FUSED
            for k, v in enumerate((1, N, H, W, Cin, Cout, relu, 1, 0, Cout, Cin)):
                op.i[k] = v
            for k, t in enumerate((dz, y, coef, x, sc if relu else None, sh if relu else None, wT, dx, None, rows, bs, slabs)):
                op.p[k] = C.ptr(t)
            C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
            hh.sync()
            outs.append((dx, rows, slabs))
        (dx0, r0, s0), (dx1, r1, s1) = outs
        assert not torch.isnan(r0).any()
        assert torch.equal(r0, r1) and torch.equal(dx0.view(torch.int16), dx1.view(torch.int16)) and torch.equal(s0, s1), (Cin, Cout)


@pytest.mark.parametrize('case', [(16, 64, 64, 64, 256), (16, 64, 64, 256, 64)])
def test_fused_pointwise_backward_is_repeatable(case):
    hh, C = _h(), _C()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(5 + Cin)
    P = N * H * W
    x = torch.randn(P, Cin, generator=g).to(DT).to(hh.DEV)
    dz = torch.randn(P, Cout, generator=g).to(DT).to(hh.DEV)
    y = torch.randn(P, Cout, generator=g).to(DT).to(hh.DEV)
    bsy = torch.randn(P, Cin, generator=g).to(DT).to(hh.DEV)
    coef = torch.randn(3 * Cout, generator=g).to(hh.DEV) * 0.5
    affine = Cin <= 64
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).to(hh.DEV), (torch.rand(Cin, generator=g) - 0.5).to(hh.DEV)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    wT, _, _ = hh.pack_weights(w, DT, mode=1)
    ns = C.call('hrnet_bwd_pw_splits', 1, P, Cin, Cout)
    n = 60
    rows = torch.empty(n, ns, 2, Cin, device=hh.DEV)
    dx = torch.empty(2, P, Cin, dtype=DT, device=hh.DEV)
    slabs = torch.empty(2, ns, Cout, Cin, device=hh.DEV)

    def launch(i):
        C.call('hrnet_conv1x1_bwd_fused', 1, dz.data_ptr(), y.data_ptr(), coef.data_ptr(), x.data_ptr(),
               sc.data_ptr() if affine else None, sh.data_ptr() if affine else None, 1 if affine else 0, wT.data_ptr(),
               dx[i % 2].data_ptr(), None, 1, rows[i].data_ptr(), bsy.data_ptr(), slabs[i % 2].data_ptr(), P, Cin, Cout,
               C.stream_ptr())
    bad, detail = _repeat(launch, lambda i: (rows[i],), n=n)
    assert bad == 0, (bad, detail)
    assert torch.equal(dx[0].view(torch.int16), dx[1].view(torch.int16)) and torch.equal(slabs[0], slabs[1])
