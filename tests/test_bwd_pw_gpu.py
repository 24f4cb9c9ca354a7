"""GPU parity of hrnet_conv1x1_bwd_fused (BatchNorm-backward apply + weight gradient + input gradient + residual
addend + ReLU mask + next BatchNorm's backward sums of a POINTWISE conv in one launch) against plain torch fp32
autograd on the CPU (autograd of the Bottleneck body, reference lib/models/pose_hrnet.py:60-105).
bf16 operands, f32 accumulation: <= 3e-2 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 3e-2
DT = torch.bfloat16


def _h():
    import hip_helpers as hh
    return hh


def _q(t):
    return t.to(DT).float()


CASES = [
    # N, H, W, Cin, Cout, affine+relu on x, addend, mask_out, coef, rows
    (2, 16, 16, 64, 256, True, False, True, True, True),      # conv3 of a Bottleneck: x = relu(bn2(y2)), rows for bn2
    (2, 16, 16, 256, 64, False, True, True, True, True),      # conv1: stored block input, residual stream added, rows through LDS
    (2, 16, 16, 64, 64, True, True, True, True, True),        # layer1.0.conv1 (64 -> 64) behind the stem
    (1, 9, 7, 64, 256, True, False, True, True, True),        # 63 pixels: one partial tile
    (3, 11, 13, 256, 64, False, True, True, True, True),      # 429 pixels: ragged last tile
    (1, 8, 8, 64, 256, False, False, False, False, False),    # bare: g = dz, no mask, no statistics
    (12, 64, 64, 64, 256, True, False, True, True, True),     # 768 tiles: every workgroup walks 3 tiles (256 workgroups)
    (12, 64, 64, 256, 64, False, True, True, True, True),
    (2, 16, 16, 256, 64, False, False, False, True, True),    # wide side, statistics without mask or addend
]


@pytest.mark.parametrize('case', CASES)
def test_fused_backward_of_conv1x1_bn(case):
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, affine, use_add, mask_out, use_coef, use_rows = case
    dt = hh.dt_id(DT)
    assert C.call('hrnet_bwd_pw_supported', dt, Cin, Cout) == 1
    assert C.call('hrnet_bwd_pw_supported', hh.dt_id(torch.float32), Cin, Cout) == 0     # fp32 keeps the unfused kernels
    assert C.call('hrnet_bwd_pw_rows_supported', dt, Cin, Cout) == 1
    g = torch.Generator().manual_seed(7 + Cin + Cout + N)
    w = _q(torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin))
    x = _q(torch.randn(N, Cin, H, W, generator=g))
    dz = _q(torch.randn(N, Cout, H, W, generator=g))
    dz = dz * (torch.rand(dz.shape, generator=g) < 0.6)
    y = _q(torch.randn(N, Cout, H, W, generator=g))
    addend = _q(torch.randn(N, Cin, H, W, generator=g))
    bsy = _q(torch.randn(N, Cin, H, W, generator=g))
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.rand(Cin, generator=g) - 0.5
    cA, cB, cC = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2, torch.randn(Cout, generator=g) * 0.1
    # ---- reference ----
    a = x
    if affine:
        a = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    a = _q(a).requires_grad_(True)
    gy = dz
    if use_coef:
        gy = cA.view(1, -1, 1, 1) * dz + cB.view(1, -1, 1, 1) * y + cC.view(1, -1, 1, 1)
    gy = _q(gy)
    wr = w.clone().requires_grad_(True)
    F.conv2d(a, wr, None).backward(gy)
    v = a.grad.clone()
    if use_add:
        v = v + addend
    if mask_out:
        v = v * (a.detach() > 0)
    want_rows = torch.stack([v.double().sum((0, 2, 3)), (v.double() * bsy.double()).sum((0, 2, 3))])
    # ---- device ----
    d = hh.DEV
    wT, _, _ = hh.pack_weights(w, DT, mode=1)
    dzd, yd, xd, add_d, bsd = (hh.nhwc(t, DT) for t in (dz, y, x, addend, bsy))
    coef = torch.cat([cA, cB, cC]).to(d)
    scd, shd = sc.to(d), sh.to(d)
    P = N * H * W
    ns = C.call('hrnet_bwd_pw_splits', dt, P, Cin, Cout)
    tiles = (P + 63) // 64
    assert 1 <= ns <= min(tiles, 256)
    if N >= 12:
        assert tiles > ns
    slabs = torch.full((ns, Cout, Cin), float('nan'), device=d)
    rows = torch.full((ns, 2, Cin), float('nan'), device=d)
    dx = torch.full((N, H, W, Cin), float('nan'), dtype=DT, device=d)
    C.call('hrnet_conv1x1_bwd_fused', dt, dzd.data_ptr(), yd.data_ptr(), coef.data_ptr() if use_coef else None,
           xd.data_ptr(), scd.data_ptr() if affine else None, shd.data_ptr() if affine else None, 1 if affine else 0,
           wT.data_ptr(), dx.data_ptr(), add_d.data_ptr() if use_add else None, 1 if mask_out else 0,
           rows.data_ptr() if use_rows else None, bsd.data_ptr() if use_rows else None, slabs.data_ptr(),
           P, Cin, Cout, C.stream_ptr())
    gw = torch.zeros(Cout, Cin, 1, 1, device=d)
    C.call('hrnet_wgrad_reduce', slabs.data_ptr(), gw.data_ptr(), ns, Cout, Cin, 1, Cout, Cin, 0, 0, C.stream_ptr())
    got = hh.from_nhwc(dx)
    assert not torch.isnan(got).any()
    tol_dx = TOL + 2.0 ** -8 * float(addend.abs().max() / v.abs().max()) * use_add
    assert hh.rel_err(got, v) <= tol_dx
    per_img = (got - v).abs().amax((1, 2, 3)) / v.abs().amax()
    assert float(per_img.max()) <= tol_dx
    assert hh.rel_err(gw.cpu(), wr.grad) <= TOL
    if use_rows:
        r = rows.double().sum(0).cpu()
        assert not torch.isnan(r).any()
        scale = v.double().abs().sum((0, 2, 3)).max().item()
        assert float((r[0] - want_rows[0]).abs().max()) <= 2 * TOL * scale
        scale2 = (v.double() * bsy.double()).abs().sum((0, 2, 3)).max().item()
        assert float((r[1] - want_rows[1]).abs().max()) <= 2 * TOL * scale2


def test_fused_pointwise_backward_in_place_on_the_residual_stream():
    """dx may be the addend's own buffer"""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout = 2, 16, 16, 256, 64
    g = torch.Generator().manual_seed(3)
    w = _q(torch.randn(Cout, Cin, 1, 1, generator=g) / 16.0)
    x = _q(torch.randn(N, Cin, H, W, generator=g))
    dz = _q(torch.randn(N, Cout, H, W, generator=g))
    stream = _q(torch.randn(N, Cin, H, W, generator=g))
    a = x.clone().requires_grad_(True)
    F.conv2d(a, w, None).backward(dz)
    want = (a.grad + stream) * (x > 0)
    d = hh.DEV
    wT, _, _ = hh.pack_weights(w, DT, mode=1)
    buf = hh.nhwc(stream, DT)
    dzd, xd = hh.nhwc(dz, DT), hh.nhwc(x, DT)
    P = N * H * W
    ns = C.call('hrnet_bwd_pw_splits', 1, P, Cin, Cout)
    slabs = torch.empty(ns, Cout, Cin, device=d)
    C.call('hrnet_conv1x1_bwd_fused', 1, dzd.data_ptr(), None, None, xd.data_ptr(), None, None, 0, wT.data_ptr(),
           buf.data_ptr(), buf.data_ptr(), 1, None, None, slabs.data_ptr(), P, Cin, Cout, C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(buf), want) <= 4e-2
