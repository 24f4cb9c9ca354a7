"""GPU parity of the conv kernels ON THE CODE PATH THE BENCHMARK RUNS: at B=64 every 64x64 / 32x32 layer
makes a workgroup walk several pixel tiles (tiles-per-workgroup 2..8: weights resident in LDS across
tiles, BatchNorm / backward statistics accumulated across tiles, the epilogue-tile prefetch) and the wide
layers take several K chunks per tile (weights re-staged). The small shapes of test_kernels_gpu.py never
reach that walk (tiles <= 512 there), so these cases are sized to force it, and each asserts through
hrnet_conv_tile_walk() that it did. Reference: plain torch fp32 on the CPU, same tolerances as
test_kernels_gpu.py (fp32 path 1e-4 relative, bf16 path 3e-2 relative)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 3e-2}
DTYPES = [torch.float32, torch.bfloat16]


def _h():
    import hip_helpers as hh
    return hh


def _q(t, dtype):
    return t.to(dtype).float()


def _walk(N, Ho, Wo, Cout, ks, stride, bwdstats=False, s2d=False):
    from hipnet import _capi as C
    out = (ctypes.c_int * 5)()
    tiles = C.call('hrnet_conv_tile_walk', N, Ho, Wo, Cout, ks, stride, 1 if bwdstats else 0, 1 if s2d else 0, out)
    return tiles, dict(th=out[0], tw=out[1], bn=out[2], tpw=out[3], gx=out[4])


def _kchunks(dtype, Cin, ks, tile_bn_is_8x8x32=False):
    """K chunks per tile (conv_body.h conv_km): 1x1 stages 4 fragment steps (2 when Cin is small), 3x3 one"""
    kstep = 16 if dtype == torch.float32 else 32
    km = (2 if Cin <= 2 * kstep else 4) if ks == 1 else (2 if tile_bn_is_8x8x32 else 1)
    return -(-Cin // (kstep * km))


FWD_CASES = [
    # N, H, W, Cin, Cout, ks, stride, affine, relu, bias, min tiles-per-workgroup, min K chunks (bf16)
    (40, 64, 64, 32, 32, 3, 1, True, True, False, 2, 1),      # the most frequent layer of the net, weights resident
    (71, 48, 48, 32, 32, 3, 1, True, True, False, 2, 1),      # odd tile count: the last workgroup walks one tile
    (64, 64, 64, 64, 64, 3, 1, True, True, False, 4, 2),      # layer1 3x3 at the benchmark batch
    (40, 64, 64, 32, 64, 3, 2, True, True, False, 2, 1),      # stride-2 (fuse-layer down path)
    (130, 16, 16, 128, 128, 3, 1, True, False, False, 2, 4),  # four K chunks, weights re-staged per stage
    (16, 64, 64, 64, 256, 1, 1, True, True, False, 2, 1),     # layer1 expand: the 128-channel GEMM-like tile
    (40, 64, 64, 256, 64, 1, 1, True, False, False, 3, 2),    # layer1 reduce: 2 K chunks x 3 tiles
    (8, 64, 64, 480, 480, 1, 1, False, False, True, 2, 4),    # head 480->480 with bias (conv_fwdb), 4 chunks
    (20, 96, 72, 48, 48, 3, 1, True, True, False, 3, 2),      # w48 geometry: partial tiles + ragged K chunk
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', FWD_CASES)
def test_conv2d_forward_multi_tile_walk(dtype, case):
    hh = _h()
    N, H, W, Cin, Cout, ks, stride, affine, relu, use_bias, min_tpw, min_nch = case
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    tiles, wk = _walk(N, Ho, Wo, Cout, ks, stride)
    assert wk['tpw'] >= min_tpw and wk['gx'] < tiles, (tiles, wk)       # this IS the multi-tile walk
    if dtype == torch.bfloat16:
        assert _kchunks(dtype, Cin, ks, (wk['th'], wk['tw'], wk['bn']) == (8, 8, 32)) >= min_nch
    g = torch.Generator().manual_seed(1000 + Cin + Cout + N)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    w = _q(torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks), dtype)
    sc = (torch.rand(Cin, generator=g) + 0.5) if affine else None
    sh = (torch.rand(Cin, generator=g) - 0.5) if affine else None
    bias = torch.randn(Cout, generator=g) if use_bias else None
    xa = x
    if affine:
        xa = xa * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if relu:
        xa = F.relu(xa)
    xa = _q(xa, dtype)
    ref = F.conv2d(xa, w, bias, stride=stride, padding=pad)
    wp, cop, cip = hh.pack_weights(w, dtype)
    y, st = hh.conv2d(hh.nhwc(x, dtype), wp, N, H, W, Cin, cop, ks, stride, dtype,
                      in_scale=sc.to(hh.DEV) if affine else None, in_shift=sh.to(hh.DEV) if affine else None,
                      bias=bias.to(hh.DEV) if use_bias else None, in_relu=relu, stats=True)
    assert st.shape[0] == wk['gx']                                      # one statistics row per pixel walk
    got = hh.from_nhwc(y, Cout)
    assert hh.rel_err(got, ref) <= TOL[dtype]
    # every image / tile individually (a wrong tile cursor would leave the global maximum intact)
    per_img = (got - ref).abs().amax((1, 2, 3)) / ref.abs().amax()
    assert float(per_img.max()) <= TOL[dtype]
    s = st.double().sum(0).cpu()
    ref_s1 = ref.double().sum((0, 2, 3))
    ref_s2 = (ref.double() ** 2).sum((0, 2, 3))
    # sums over up to 4e5 values per channel: compare against the scale of the summands
    scale1 = ref.double().abs().sum((0, 2, 3)).max().item()
    assert float((s[0, :Cout] - ref_s1).abs().max()) <= (2 * TOL[dtype] + 1e-5) * scale1
    assert hh.rel_err(s[1, :Cout], ref_s2) <= 5 * TOL[dtype]


DG_CASES = [
    # forward-conv view: N, H, W, Cin, Cout, ks, stride, min tiles-per-workgroup
    (40, 64, 64, 32, 32, 3, 1, 2),
    (64, 64, 64, 64, 64, 3, 1, 4),
    (8, 64, 64, 256, 64, 3, 2, 2),      # transition1 256->64 stride 2: four-parity gradient (S2D), 2 K chunks
    (40, 64, 64, 64, 256, 1, 1, 3),     # gradient of the layer1 expand conv: K = 256 channels of dY
    (130, 16, 16, 128, 128, 3, 1, 2),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', DG_CASES)
def test_input_gradient_multi_tile_walk(dtype, case):
    """plain input gradient (conv_dg_kernel) accumulating into an existing gradient buffer"""
    hh = _h()
    N, H, W, Cin, Cout, ks, stride, min_tpw = case
    tiles, wk = _walk(N, H, W, Cin, ks, stride, False, ks == 3 and stride == 2)
    assert wk['tpw'] >= min_tpw and wk['gx'] < tiles, (tiles, wk)
    g = torch.Generator().manual_seed(77 + Cin)
    w = _q(torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks), dtype)
    x = torch.zeros(N, Cin, H, W, requires_grad=True)
    y = F.conv2d(x, w, None, stride=stride, padding=ks // 2)
    dy = _q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    ref = x.grad
    Ho, Wo = y.shape[2], y.shape[3]
    wd, _, _ = hh.pack_weights(w, dtype, mode=1)
    prev = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    out = hh.nhwc(prev, dtype)
    dx, _ = hh.conv2d(hh.nhwc(dy, dtype), wd, N, Ho, Wo, Cout, Cin, ks, stride, dtype, upz=(stride == 2), out=out,
                      accumulate=True, out_hw=(H, W))
    got = hh.from_nhwc(dx) - prev
    # bf16: the stored sum prev + dx is rounded once more
    tol = 2 * TOL[dtype] if dtype == torch.float32 else 2 * TOL[dtype] + 2.0 ** -8 * float(prev.abs().max() / ref.abs().max())
    assert hh.rel_err(got, ref) <= tol
    per_img = (got - ref).abs().amax((1, 2, 3)) / ref.abs().amax()
    assert float(per_img.max()) <= tol


BS_CASES = [
    (40, 64, 64, 32, 32, 3, 1, 2),
    (130, 32, 32, 64, 64, 3, 1, 3),
    (8, 64, 64, 256, 64, 3, 2, 2),
    (40, 64, 64, 64, 256, 1, 1, 3),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('mode', ['bn_relu', 'sum_mask'])
@pytest.mark.parametrize('case', BS_CASES)
def test_backward_statistics_multi_tile_walk(dtype, mode, case):
    """hrnet_conv2d_bwdstats on the walk: gradient + (sum dz, sum dz*y) rows accumulated over several tiles,
    epilogue operands prefetched for the tile being finished while the next tile's loads are in flight."""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, ks, stride, min_tpw = case
    tiles, wk = _walk(N, H, W, Cin, ks, stride, True, ks == 3 and stride == 2)
    assert wk['tpw'] >= min_tpw and wk['gx'] < tiles, (tiles, wk)
    g = torch.Generator().manual_seed(31 + Cout)
    w = _q(torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks), dtype)
    Ho = (H + 2 * (ks // 2) - ks) // stride + 1
    Wo = (W + 2 * (ks // 2) - ks) // stride + 1
    dy = _q(torch.randn(N, Cout, Ho, Wo, generator=g), dtype)
    yraw = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    outv = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    prev = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    sc, sf = torch.rand(Cin, generator=g) + 0.5, torch.rand(Cin, generator=g) - 0.5
    d = hh.DEV
    wd, _, _ = hh.pack_weights(w, dtype, mode=1)
    dyd, yd, od = hh.nhwc(dy, dtype), hh.nhwc(yraw, dtype), hh.nhwc(outv, dtype)
    gx = hh.nhwc(prev, dtype)
    scd, sfd = sc.to(d), sf.to(d)
    rows_n = C.call('hrnet_conv_tiles_bwdstats', N, H, W, Cin, ks, stride)
    assert rows_n == wk['gx']
    rows = torch.full((rows_n, 2, Cin), float('nan'), device=d)
    C.call('hrnet_conv2d_bwdstats', hh.dt_id(dtype), dyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), rows.data_ptr(),
           yd.data_ptr(), od.data_ptr() if mode == 'sum_mask' else None,
           scd.data_ptr() if mode == 'bn_relu' else None, sfd.data_ptr() if mode == 'bn_relu' else None,
           N, Ho, Wo, Cout, H, W, Cin, ks, stride, 1 if stride == 2 else 0, 1, C.stream_ptr())
    v = hh.from_nhwc(gx).double()
    x = torch.zeros(N, Cin, H, W, requires_grad=True)
    F.conv2d(x, w, None, stride=stride, padding=ks // 2).backward(dy)
    tol = 2 * TOL[dtype] if dtype == torch.float32 else 2 * TOL[dtype] + 2.0 ** -8 * float(prev.abs().max() / x.grad.abs().max())
    assert hh.rel_err(v - prev.double(), x.grad) <= tol
    if mode == 'bn_relu':
        m = (yraw * sc.view(1, -1, 1, 1) + sf.view(1, -1, 1, 1)) > 0
    else:
        m = outv > 0
    dz = v * m
    want = torch.stack([dz.sum((0, 2, 3)), (dz * yraw.double()).sum((0, 2, 3))])
    got = rows.double().sum(0).cpu()
    assert not torch.isnan(got).any()
    scale = dz.abs().sum((0, 2, 3)).max().item()
    assert float((got - want).abs().max()) <= 3 * TOL[dtype] * scale


WG_CASES = [
    (40, 64, 64, 32, 32, 3, 1, True),
    (64, 32, 32, 64, 64, 3, 1, True),
    (40, 64, 64, 32, 64, 3, 2, True),
    (16, 64, 64, 256, 64, 1, 1, True),
    (8, 64, 64, 480, 480, 1, 1, False),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', WG_CASES)
def test_weight_gradient_many_tiles_per_split(dtype, case):
    """each split of the weight-gradient kernel walks several pixel tiles (register prefetch of the next
    tile under the MFMAs of the current one): at the small test shapes a split sees a single tile"""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, ks, stride, affine = case
    g = torch.Generator().manual_seed(11 + Cout)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    sc = (torch.rand(Cin, generator=g) + 0.5) if affine else None
    sh = (torch.rand(Cin, generator=g) - 0.5) if affine else None
    xa = x
    if affine:
        xa = _q(F.relu(xa * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), dtype)
    w = torch.randn(Cout, Cin, ks, ks, generator=g, requires_grad=True)
    y = F.conv2d(xa, w, None, stride=stride, padding=ks // 2)
    dy = _q(torch.randn(y.shape, generator=g) / np.sqrt(N * y.shape[2] * y.shape[3]), dtype)
    y.backward(dy)
    Ho, Wo = y.shape[2], y.shape[3]
    ns = C.call('hrnet_wgrad_splits', hh.dt_id(dtype), N, Ho, Wo, Cout, Cin, ks, stride)
    tiles = C.call('hrnet_wgrad_tiles', hh.dt_id(dtype), N, Ho, Wo, Cout, Cin, ks, stride)
    assert tiles >= 2 * ns, (tiles, ns)
    got = hh.wgrad(hh.nhwc(x, dtype), hh.nhwc(dy, dtype), N, H, W, Cin, Ho, Wo, Cout, ks, stride, dtype,
                   in_scale=sc.to(hh.DEV) if affine else None, in_shift=sh.to(hh.DEV) if affine else None,
                   in_relu=affine).cpu()
    assert hh.rel_err(got, w.grad) <= 2 * TOL[dtype]
