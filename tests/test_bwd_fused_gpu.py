"""GPU parity of hrnet_conv3x3_bwd_fused (BatchNorm-backward apply + weight gradient + input gradient +
residual addend + ReLU mask + next BatchNorm's backward sums in one launch) against plain torch fp32 autograd
on the CPU (autograd of the BasicBlock body, reference lib/models/pose_hrnet.py:41-57).
fp32 device path <= 2e-4 relative, bf16 <= 3e-2 relative (bf16 operands, f32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 3e-2}


def _h():
    import hip_helpers as hh
    return hh


def _q(t, dtype):
    return t.to(dtype).float()


CASES = [
    # N, H, W, Cin, Cout, affine+relu on x, addend, mask_out, coef, rows
    (2, 16, 16, 32, 32, True, False, True, True, True),      # conv2 of a BasicBlock: x = relu(bn1(y1))
    (2, 16, 16, 32, 32, False, True, True, True, True),      # conv1: plain x, residual stream added, masked by x
    (3, 20, 24, 32, 32, True, True, True, True, True),       # partial tiles on both axes
    (1, 16, 16, 32, 32, False, False, False, False, False),  # bare: g = dz, no mask, no statistics
    (2, 32, 32, 64, 64, True, True, True, True, True),       # 64 channels: two input-channel blocks per walk
    (1, 36, 24, 48, 48, True, True, True, True, True),       # w48 widths: ragged channel blocks
    (2, 16, 16, 64, 32, True, False, True, True, True),      # Cin != Cout
    (40, 64, 64, 32, 32, True, True, True, True, True),      # benchmark-like: every workgroup walks 4 tiles
    (70, 32, 32, 64, 64, False, True, True, True, True),     # 64 channels, 4 tiles per walk
    (2, 16, 16, 128, 128, True, True, True, True, True),     # 128 channels (third branch): 8x16 tiles, four input-channel blocks
    (64, 16, 16, 128, 128, False, True, True, True, True),   # ... at the benchmark batch: every workgroup walks 4 tiles
    (2, 20, 12, 96, 96, True, True, True, True, True),       # w48's third branch: 96 channels in a 128-channel instantiation
    (2, 16, 16, 128, 64, True, False, True, True, True),     # Cin != Cout
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CASES)
def test_fused_backward_of_conv3x3_bn(dtype, case):
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, affine, use_add, mask_out, use_coef, use_rows = case
    dt = hh.dt_id(dtype)
    if not C.call('hrnet_bwd_fused_supported', dt, Cin, Cout):
        assert dtype == torch.float32 and Cout > 32       # the only shapes the fp32 instantiation leaves out
        pytest.skip('shape served by the unfused kernels in fp32')
    g = torch.Generator().manual_seed(5 + Cin + Cout + N)
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9), dtype)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    dz = _q(torch.randn(N, Cout, H, W, generator=g), dtype)
    dz = dz * (torch.rand(dz.shape, generator=g) < 0.6)            # a masked gradient: zeros where the ReLU was off
    y = _q(torch.randn(N, Cout, H, W, generator=g), dtype)
    addend = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    bsy = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.rand(Cin, generator=g) - 0.5
    cA, cB, cC = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2, torch.randn(Cout, generator=g) * 0.1
    # ---- reference ----
    a = x
    if affine:
        a = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    a = _q(a, dtype).requires_grad_(True)
    gy = dz
    if use_coef:
        gy = cA.view(1, -1, 1, 1) * dz + cB.view(1, -1, 1, 1) * y + cC.view(1, -1, 1, 1)
    gy = _q(gy, dtype)                                             # the kernel stages g in the compute dtype
    wr = w.clone().requires_grad_(True)
    F.conv2d(a, wr, None, padding=1).backward(gy)
    v = a.grad.clone()
    if use_add:
        v = v + addend
    if mask_out:
        v = v * (a.detach() > 0)
    want_rows = torch.stack([v.double().sum((0, 2, 3)), (v.double() * bsy.double()).sum((0, 2, 3))])
    # ---- device ----
    d = hh.DEV
    wT, _, _ = hh.pack_weights(w, dtype, mode=1)
    dzd, yd, xd, add_d, bsd = (hh.nhwc(t, dtype) for t in (dz, y, x, addend, bsy))
    coef = torch.cat([cA, cB, cC]).to(d)
    scd, shd = sc.to(d), sh.to(d)
    ns = C.call('hrnet_bwd_fused_splits', dt, N, H, W, Cin, Cout)
    tiles = N * ((H + 15) // 16) * ((W + 15) // 16)
    assert 1 <= ns <= 2 * tiles              # (a variant with 8x16 tiles has twice the tiles)
    if N >= 40 and Cout <= 64:
        assert tiles >= 2 * ns                                     # the multi-tile walk with register prefetch
    slabs = torch.full((ns, Cout, 9, Cin), float('nan'), device=d)
    rows = torch.full((ns, 2, Cin), float('nan'), device=d)
    dx = torch.full((N, H, W, Cin), float('nan'), dtype=dtype, device=d)
    C.call('hrnet_conv3x3_bwd_fused', dt, dzd.data_ptr(), yd.data_ptr(), coef.data_ptr() if use_coef else None,
           xd.data_ptr(), scd.data_ptr() if affine else None, shd.data_ptr() if affine else None, 1 if affine else 0,
           wT.data_ptr(), dx.data_ptr(), add_d.data_ptr() if use_add else None, 1 if mask_out else 0,
           rows.data_ptr() if use_rows else None, bsd.data_ptr() if use_rows else None, slabs.data_ptr(),
           N, H, W, Cin, Cout, C.stream_ptr())
    gw = torch.zeros(Cout, Cin, 3, 3, device=d)
    C.call('hrnet_wgrad_reduce', slabs.data_ptr(), gw.data_ptr(), ns, Cout, Cin, 3, Cout, Cin, 0, 0, C.stream_ptr())
    got = hh.from_nhwc(dx)
    assert not torch.isnan(got).any()
    tol = TOL[dtype]
    # the stored value is rounded once more in bf16 (sum with the addend)
    tol_dx = tol if dtype == torch.float32 else tol + 2.0 ** -8 * float(addend.abs().max() / v.abs().max()) * use_add
    assert hh.rel_err(got, v) <= tol_dx
    per_img = (got - v).abs().amax((1, 2, 3)) / v.abs().amax()
    assert float(per_img.max()) <= tol_dx
    assert hh.rel_err(gw.cpu(), wr.grad) <= tol
    if use_rows:
        r = rows.double().sum(0).cpu()
        assert not torch.isnan(r).any()
        scale = v.double().abs().sum((0, 2, 3)).max().item()
        assert float((r[0] - want_rows[0]).abs().max()) <= 2 * tol * scale
        scale2 = (v.double() * bsy.double()).abs().sum((0, 2, 3)).max().item()
        assert float((r[1] - want_rows[1]).abs().max()) <= 2 * tol * scale2


def test_fused_backward_in_place_on_the_residual_stream():
    """dx may be the addend's own buffer (the gradient stream of a branch is updated in place)"""
    hh = _h()
    from hipnet import _capi as C
    dtype = torch.bfloat16
    N, H, W, Cc = 2, 16, 16, 32
    g = torch.Generator().manual_seed(3)
    w = _q(torch.randn(Cc, Cc, 3, 3, generator=g) / 17.0, dtype)
    x = _q(torch.randn(N, Cc, H, W, generator=g), dtype)
    dz = _q(torch.randn(N, Cc, H, W, generator=g), dtype)
    stream = _q(torch.randn(N, Cc, H, W, generator=g), dtype)
    a = x.clone().requires_grad_(True)
    F.conv2d(a, w, None, padding=1).backward(dz)
    want = (a.grad + stream) * (x > 0)
    d = hh.DEV
    wT, _, _ = hh.pack_weights(w, dtype, mode=1)
    buf = hh.nhwc(stream, dtype)
    dzd, xd = hh.nhwc(dz, dtype), hh.nhwc(x, dtype)       # (kept alive: the allocator would hand a freed block out again)
    ns = C.call('hrnet_bwd_fused_splits', 1, N, H, W, Cc, Cc)
    slabs = torch.empty(ns, Cc, 9, Cc, device=d)
    C.call('hrnet_conv3x3_bwd_fused', hh.dt_id(dtype), dzd.data_ptr(), None, None, xd.data_ptr(),
           None, None, 0, wT.data_ptr(), buf.data_ptr(), buf.data_ptr(), 1, None, None, slabs.data_ptr(), N, H, W, Cc, Cc,
           C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(buf), want) <= 4e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 16, 16, 32, 32, 128, 3), (3, 20, 24, 64, 64, 64, 3), (2, 16, 16, 128, 128, 32, 3), (2, 16, 16, 64, 256, 7, 1),
                                   (2, 16, 16, 256, 64, 128, 1), (2, 16, 16, 64, 64, 1, 1)])
def test_fused_launch_finishes_its_own_batchnorm_backward_from_rows(dtype, shape):
    """HrBnBwdRef: the launch builds A,B,C from the partial rows a previous launch left and adds dgamma/dbeta -
    same outputs as hrnet_bn_bwd_finalize followed by the launch with `coef` (bit-identical gradients: the
    coefficients are computed by the same f64 formula; only the order of the row sum differs)"""
    import ctypes
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, nrows, ks = shape
    dt = hh.dt_id(dtype)
    if ks == 3 and not C.call('hrnet_bwd_fused_supported', dt, Cin, Cout):
        pytest.skip('shape served by the unfused kernels in fp32')
    if ks == 1 and not C.call('hrnet_bwd_pw_supported', dt, Cin, Cout):
        pytest.skip('the pointwise fused backward is bf16 only')
    g = torch.Generator().manual_seed(11 + Cin + Cout + nrows)
    d = hh.DEV
    w = _q(torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks), dtype)
    x, dz, y = (_q(torch.randn(N, c, H, W, generator=g), dtype) for c in (Cin, Cout, Cout))
    wT, _, _ = hh.pack_weights(w, dtype, mode=1)
    xd, dzd, yd = (hh.nhwc(t, dtype) for t in (x, dz, y))
    rows_in = torch.randn(nrows, 2, Cout, generator=g).to(d)
    gamma, mean = (torch.rand(Cout, generator=g) + 0.5).to(d), torch.randn(Cout, generator=g).to(d)
    invstd = (torch.rand(Cout, generator=g) + 0.5).to(d)
    count = float(N * H * W)
    P = N * H * W
    if ks == 3:
        ns = C.call('hrnet_bwd_fused_splits', dt, N, H, W, Cin, Cout)
    else:
        ns = C.call('hrnet_bwd_pw_splits', dt, P, Cin, Cout)
    outs = []
    for mode in ('finalize', 'inline'):
        dgam, dbet = torch.full((Cout,), 0.25, device=d), torch.full((Cout,), -0.5, device=d)   # accumulate into these
        coef = torch.zeros(3 * Cout, device=d)
        slabs = torch.zeros(ns, Cout, ks * ks, Cin, device=d)
        dx = torch.zeros(N, H, W, Cin, dtype=dtype, device=d)
        ref = None
        if mode == 'finalize':
            C.call('hrnet_bn_bwd_finalize', rows_in.data_ptr(), nrows, Cout, count, gamma.data_ptr(), mean.data_ptr(),
                   invstd.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), 1, C.stream_ptr())
        else:
            ref = C.HrBnBwdRef()
            ref.rows, ref.gamma, ref.save_mean, ref.save_invstd = (t.data_ptr() for t in (rows_in, gamma, mean, invstd))
            ref.dgamma, ref.dbeta, ref.count, ref.nrows, ref.accumulate = dgam.data_ptr(), dbet.data_ptr(), count, nrows, 1
        rp = ctypes.addressof(ref) if ref is not None else None
        cp = coef.data_ptr() if ref is None else None
        if ks == 3:
            C.call('hrnet_conv3x3_bwd_fused_bnref', dt, dzd.data_ptr(), yd.data_ptr(), cp, rp, xd.data_ptr(), None, None, 0,
                   wT.data_ptr(), dx.data_ptr(), None, 1, None, None, slabs.data_ptr(), N, H, W, Cin, Cout, C.stream_ptr())
        else:
            C.call('hrnet_conv1x1_bwd_fused_bnref', dt, dzd.data_ptr(), yd.data_ptr(), cp, rp, xd.data_ptr(), None, None, 0,
                   wT.data_ptr(), dx.data_ptr(), None, 1, None, None, slabs.data_ptr(), P, Cin, Cout, C.stream_ptr())
        torch.cuda.synchronize()
        outs.append((dx.float().cpu(), slabs.sum(0).cpu(), dgam.cpu(), dbet.cpu()))
    (dx_f, gw_f, dg_f, db_f), (dx_i, gw_i, dg_i, db_i) = outs
    assert dx_f.abs().max() > 0
    # dgamma/dbeta: f64 sums in another order, rounded to f32 once
    np.testing.assert_allclose(dg_i.numpy(), dg_f.numpy(), rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(db_i.numpy(), db_f.numpy(), rtol=2e-6, atol=1e-5)
    # the coefficients may differ in the last f32 bit -> the bf16-rounded g in a few places
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert hh.rel_err(dx_i, dx_f) <= tol
    assert hh.rel_err(gw_i, gw_f) <= tol
