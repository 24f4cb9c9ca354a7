"""GPU parity of the individual HIP kernels (called through the C ABI) against plain torch
fp32 on the CPU. fp32 device path: <= 1e-4 relative (exact f32 MFMA, only summation order
differs); bf16 path: <= 3e-2 relative (bf16 operands, f32 accumulation)."""
import os

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
    """round a CPU f32 tensor through the device dtype so both sides see identical inputs"""
    return t.to(dtype).float()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, ks, stride, affine, relu, bias
    (2, 16, 16, 32, 32, 3, 1, True, True, False),
    (1, 64, 64, 32, 32, 3, 1, False, False, False),
    (2, 32, 32, 64, 64, 3, 1, True, True, False),
    (3, 16, 16, 128, 128, 3, 1, True, False, False),
    (2, 8, 8, 256, 256, 3, 1, True, True, False),
    (2, 32, 32, 64, 128, 3, 2, True, True, False),
    (2, 16, 16, 32, 32, 3, 2, False, False, False),
    (2, 16, 16, 256, 32, 3, 1, True, True, False),
    (2, 16, 16, 64, 256, 1, 1, True, True, False),
    (1, 16, 16, 480, 480, 1, 1, False, False, True),
    (2, 8, 8, 480, 32, 1, 1, True, True, True),
    (1, 12, 9, 48, 96, 3, 1, True, True, False),     # w48-style odd sizes, Cin not /32
    (1, 12, 20, 48, 48, 3, 2, False, True, False),
])
def test_conv2d_forward(dtype, case):
    hh = _h()
    N, H, W, Cin, Cout, ks, stride, affine, relu, use_bias = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
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
    xa = _q(xa, dtype)   # the kernel rounds the transformed operand to the compute dtype
    ref = F.conv2d(xa, w, bias, stride=stride, padding=ks // 2)
    xd = hh.nhwc(x, dtype)
    wp, cop, cip = hh.pack_weights(w, dtype)
    y, st = hh.conv2d(xd, wp, N, H, W, Cin, cop, ks, stride, dtype,
                      in_scale=sc.to(hh.DEV) if affine else None, in_shift=sh.to(hh.DEV) if affine else None,
                      bias=bias.to(hh.DEV) if use_bias else None, in_relu=relu, stats=True)
    got = hh.from_nhwc(y, Cout)
    assert hh.rel_err(got, ref) <= TOL[dtype]
    # BatchNorm statistics from the epilogue (f32 accumulators)
    s = st.sum(0).cpu()
    ref_s1 = ref.sum((0, 2, 3))
    ref_s2 = (ref * ref).sum((0, 2, 3))
    assert hh.rel_err(s[0, :Cout], ref_s1) <= 5 * TOL[dtype] + 1e-4
    assert hh.rel_err(s[1, :Cout], ref_s2) <= 5 * TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [
    (2, 16, 16, 32, 32, 3, 1),
    (2, 32, 32, 64, 128, 3, 2),
    (1, 16, 16, 32, 64, 3, 2),
    (2, 16, 16, 64, 256, 1, 1),
    (1, 9, 12, 48, 96, 3, 2),      # odd input extent
    (2, 8, 8, 256, 256, 3, 1),
])
def test_conv2d_input_gradient(dtype, case):
    """dgrad = conv of dY with the transposed/flipped packed kernel (zero-stuffed for stride 2)."""
    hh = _h()
    N, H, W, Cin, Cout, ks, stride = case
    g = torch.Generator().manual_seed(7)
    w = _q(torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks), dtype)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride=stride, padding=ks // 2)
    dy = _q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    ref = x.grad
    Ho, Wo = y.shape[2], y.shape[3]
    wd, _, _ = hh.pack_weights(w, dtype, mode=1)
    dyd = hh.nhwc(dy, dtype)
    prev = torch.randn(N, H, W, Cin, generator=g)
    out = prev.to(dtype).to(hh.DEV)
    dx, _ = hh.conv2d(dyd, wd, N, Ho, Wo, Cout, Cin, ks, stride, dtype, upz=(stride == 2), out=out,
                      accumulate=True, out_hw=(H, W))
    got = hh.from_nhwc(dx) - hh.from_nhwc(prev.to(dtype))
    assert hh.rel_err(got, ref) <= 2 * TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [
    (2, 16, 16, 32, 32, 3, 1, True),
    (1, 64, 64, 32, 32, 3, 1, False),
    (2, 32, 32, 64, 64, 3, 1, True),
    (2, 8, 8, 256, 256, 3, 1, True),
    (2, 32, 32, 64, 128, 3, 2, True),
    (2, 16, 16, 32, 32, 3, 2, False),
    (2, 16, 16, 64, 256, 1, 1, True),
    (2, 16, 16, 480, 480, 1, 1, False),
    (1, 12, 9, 48, 96, 3, 1, True),
    (2, 4, 4, 128, 32, 1, 1, True),
])
def test_conv2d_weight_gradient(dtype, case):
    hh = _h()
    N, H, W, Cin, Cout, ks, stride, affine = case
    g = torch.Generator().manual_seed(11)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    sc = (torch.rand(Cin, generator=g) + 0.5) if affine else None
    sh = (torch.rand(Cin, generator=g) - 0.5) if affine else None
    xa = x
    if affine:
        xa = _q(F.relu(xa * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), dtype)
    w = torch.randn(Cout, Cin, ks, ks, generator=g, requires_grad=True)
    y = F.conv2d(xa, w, None, stride=stride, padding=ks // 2)
    dy = _q(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    Ho, Wo = y.shape[2], y.shape[3]
    got = hh.wgrad(hh.nhwc(x, dtype), hh.nhwc(dy, dtype), N, H, W, Cin, Ho, Wo, Cout, ks, stride, dtype,
                   in_scale=sc.to(hh.DEV) if affine else None, in_shift=sh.to(hh.DEV) if affine else None,
                   in_relu=affine).cpu()
    assert hh.rel_err(got, w.grad) <= 2 * TOL[dtype]


def test_batched_slab_sum_is_bit_identical_to_per_layer_calls():
    """hrnet_wgrad_reduce_table (one launch for several layers) == hrnet_wgrad_reduce per layer"""
    import ctypes
    from hipnet import _capi as C
    d = 'cuda:0'
    g = torch.Generator(device=d).manual_seed(5)
    layers = [(7, 32, 32, 3, 32, 32, 0), (130, 64, 64, 3, 64, 48, 0), (5, 32, 32, 1, 21, 32, 0), (3, 64, 32, 3, 64, 3, 1)]
    ents = (C.HrWredEnt * len(layers))()
    keep, want, got = [], [], []
    block = 0
    for e, (ns, cop, cip, ks, co, ci, kflat) in zip(ents, layers):
        taps = ks * ks
        slabs = torch.randn(ns, cop, 1 if kflat else taps, cip, device=d, generator=g)
        ref = torch.randn(co, ci, ks, ks, device=d, generator=g)
        out = ref.clone()
        C.call('hrnet_wgrad_reduce', slabs.data_ptr(), ref.data_ptr(), ns, cop, cip, ks, co, ci, kflat, 1, C.stream_ptr())
        e.slabs, e.grad = slabs.data_ptr(), out.data_ptr()
        e.nsplit, e.Cout_pad, e.Cin_pad, e.ks, e.Cout, e.Cin, e.kflat, e.accumulate = ns, cop, cip, ks, co, ci, kflat, 1
        e.block0 = block
        block += (co * ci * taps + 63) // 64
        keep.append(slabs); want.append(ref); got.append(out)
    raw = bytes(ctypes.string_at(ctypes.addressof(ents), ctypes.sizeof(ents)))
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(d)
    C.call('hrnet_wgrad_reduce_table', table.data_ptr(), len(layers), block, C.stream_ptr())
    torch.cuda.synchronize()
    for w, o in zip(want, got):
        assert torch.equal(w, o)


@pytest.mark.parametrize('dtype', DTYPES)
def test_stem_im2col_conv_and_wgrad(dtype):
    hh = _h()
    from hipnet import _capi as C
    N, H, W = 2, 32, 48
    g = torch.Generator().manual_seed(3)
    img = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g, requires_grad=True)
    wq = _q(w.detach(), dtype)
    ref = F.conv2d(_q(img, dtype), wq, None, stride=2, padding=1)
    cols = torch.empty(N, H // 2, W // 2, 32, dtype=dtype, device=hh.DEV)
    imgd = img.to(hh.DEV)
    C.call('hrnet_im2col_stem', hh.dt_id(dtype), imgd.data_ptr(), cols.data_ptr(), N, 3, H, W, H // 2,
           W // 2, 32, C.stream_ptr())
    wp, _, _ = hh.pack_weights(wq, dtype, mode=2)
    y, _ = hh.conv2d(cols, wp, N, H // 2, W // 2, 32, 64, 1, 1, dtype)
    assert hh.rel_err(hh.from_nhwc(y), ref) <= TOL[dtype]
    # weight gradient through the flattened-K slab
    y2 = F.conv2d(_q(img, dtype), w, None, stride=2, padding=1)
    dy = _q(torch.randn(y2.shape, generator=g), dtype)
    y2.backward(dy)
    ns = C.call('hrnet_wgrad_splits', hh.dt_id(dtype), N, H // 2, W // 2, 64, 32, 1, 1)
    slabs = torch.zeros(ns, 64, 1, 32, dtype=torch.float32, device=hh.DEV)
    dyd = hh.nhwc(dy, dtype)
    C.call('hrnet_conv2d_wgrad', hh.dt_id(dtype), cols.data_ptr(), dyd.data_ptr(), None, None,
           slabs.data_ptr(), N, H // 2, W // 2, 32, H // 2, W // 2, 64, 1, 1, 0, ns, C.stream_ptr())
    gw = torch.zeros(64, 3, 3, 3, dtype=torch.float32, device=hh.DEV)
    C.call('hrnet_wgrad_reduce', slabs.data_ptr(), gw.data_ptr(), ns, 64, 32, 3, 64, 3, 1, 0, C.stream_ptr())
    assert hh.rel_err(gw.cpu(), w.grad) <= 2 * TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_finalize_train_and_eval(dtype):
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cc = 3, 16, 16, 64
    g = torch.Generator().manual_seed(5)
    x = _q(torch.randn(N, 32, H, W, generator=g), dtype)
    w = _q(torch.randn(Cc, 32, 3, 3, generator=g) * 0.1, dtype)
    y_ref = F.conv2d(x, w, None, padding=1)
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(Cc, generator=g) - 0.5)
        bn.running_mean.copy_(torch.rand(Cc, generator=g))
        bn.running_var.copy_(torch.rand(Cc, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train()
    z_ref = bn(y_ref)
    wp, _, _ = hh.pack_weights(w, dtype)
    y, st = hh.conv2d(hh.nhwc(x, dtype), wp, N, H, W, 32, Cc, 3, 1, dtype, stats=True)
    d = hh.DEV
    gam, bet = bn.weight.detach().to(d), bn.bias.detach().to(d)
    rm, rv = rm0.to(d), rv0.to(d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    scale, shift, mean, invstd = (torch.empty(Cc, device=d) for _ in range(4))
    C.call('hrnet_bn_finalize', st.data_ptr(), st.shape[0], Cc, float(N * H * W), gam.data_ptr(), bet.data_ptr(),
           rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, 1e-5, 1, scale.data_ptr(), shift.data_ptr(),
           mean.data_ptr(), invstd.data_ptr(), C.stream_ptr())
    z = hh.from_nhwc(y) * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)
    assert hh.rel_err(z, z_ref.detach()) <= 2 * TOL[dtype]
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=tol, atol=tol)
    assert int(nbt.item()) == 1
    # eval: scale/shift from the running statistics
    C.call('hrnet_bn_finalize', None, 0, Cc, 1.0, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(),
           None, 0.1, 1e-5, 0, scale.data_ptr(), shift.data_ptr(), None, None, C.stream_ptr())
    ref_scale = gam.cpu() / torch.sqrt(rv.cpu() + 1e-5)
    np.testing.assert_allclose(scale.cpu().numpy(), ref_scale.numpy(), rtol=1e-5)
    np.testing.assert_allclose(shift.cpu().numpy(), (bet.cpu() - rm.cpu() * ref_scale).numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(3, 20, 24, 32, 32, 3), (2, 8, 8, 64, 64, 3), (2, 16, 16, 128, 128, 3), (2, 16, 16, 256, 64, 1),
                                   (40, 64, 64, 32, 32, 3), (2, 9, 7, 48, 48, 3)])
@pytest.mark.parametrize('mode', ['affine', 'sums'])
def test_conv_with_the_residual_sum_in_its_prologue(dtype, shape, mode):
    """hrnet_conv2d_sum: a = relu(bn(x) + x2) formed while staging, y = conv(a) with its batch statistics, and a
    written out once per pixel - against hrnet_sum_terms followed by hrnet_conv2d on the device (same arithmetic:
    identical `side`, y within rounding order) and against torch (pose_hrnet.py:54-55 + :44 of the next block)."""
    import ctypes
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, ks = shape
    g = torch.Generator().manual_seed(5 + Cin + Cout + N + ks)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    x2 = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    w = _q(torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks), dtype)
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.rand(Cin, generator=g) - 0.5
    d = hh.DEV
    cnt = float(N * H * W)
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    sc = gamma / torch.sqrt(var + 1e-5)
    sh = beta - mean * sc
    a_ref = _q(F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + x2), dtype)
    y_ref = F.conv2d(a_ref, w, None, padding=ks // 2)
    xd, x2d = hh.nhwc(x, dtype), hh.nhwc(x2, dtype)
    wp, _, _ = hh.pack_weights(w, dtype)
    side = torch.full((N, H, W, Cin), float('nan'), dtype=dtype, device=d)
    y = torch.full((N, H, W, Cout), float('nan'), dtype=dtype, device=d)
    osums = torch.zeros(8, 2, Cout, device=d)
    scd, shd, gd, bd = sc.to(d), sh.to(d), gamma.to(d), beta.to(d)
    sums = torch.zeros(8, 2, Cin, device=d)
    sums[3, 0] = (x.double().sum((0, 2, 3))).float().to(d)
    sums[5, 1] = ((x.double() ** 2).sum((0, 2, 3))).float().to(d)
    if mode == 'affine':
        C.call('hrnet_conv2d_sum', hh.dt_id(dtype), xd.data_ptr(), x2d.data_ptr(), wp.data_ptr(), scd.data_ptr(), shd.data_ptr(),
               None, None, None, 0.0, 0.0, side.data_ptr(), y.data_ptr(), osums.data_ptr(), 1, N, H, W, Cin, Cout, ks, C.stream_ptr())
    else:
        C.call('hrnet_conv2d_sum', hh.dt_id(dtype), xd.data_ptr(), x2d.data_ptr(), wp.data_ptr(), None, None,
               sums.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1.0 / cnt, 1e-5, side.data_ptr(), y.data_ptr(), osums.data_ptr(), 1,
               N, H, W, Cin, Cout, ks, C.stream_ptr())
    a_got = hh.from_nhwc(side)
    assert not torch.isnan(a_got).any() and not torch.isnan(y.float()).any()
    tol = TOL[dtype] if mode == 'affine' else 2 * TOL[dtype] + (0 if dtype == torch.float32 else 1e-2)
    assert hh.rel_err(a_got, a_ref) <= (1e-5 if dtype == torch.float32 else 2.0 ** -7)
    assert hh.rel_err(hh.from_nhwc(y), y_ref) <= tol
    s1 = osums.double().sum(0).cpu()
    assert hh.rel_err(s1[0], y_ref.double().sum((0, 2, 3))) <= 5 * tol + 1e-3
    assert hh.rel_err(s1[1], (y_ref.double() ** 2).sum((0, 2, 3))) <= 5 * tol
    if mode == 'affine':
        # the unfused pair on the device: the same sum values bit for bit
        out = torch.empty(N, H, W, Cin, dtype=dtype, device=d)
        C.call('hrnet_sum_terms', hh.dt_id(dtype), out.data_ptr(), N, H, W, Cin, 2, hh.ptr_array([xd, x2d]),
               hh.ptr_array([scd, None]), hh.ptr_array([shd, None]), hh.int_array([0, 0]), hh.int_array([0, 0]), 1,
               C.stream_ptr())
        assert torch.equal(out.float().cpu(), side.float().cpu())


@pytest.mark.parametrize('shape', [(2, 16, 16, 480, 480), (1, 9, 7, 480, 480), (3, 8, 8, 256, 256), (2, 8, 8, 512, 512),
                                   (1, 16, 16, 480, 256)])
def test_head_gemm_forward_with_bias_and_statistics_and_input_gradient(shape):
    """the GEMM kernel behind hrnet_conv2d for the head's 1x1 480 -> 480 layer (bf16): forward with bias and
    atomically accumulated batch statistics (hrnet_conv2d_bnref), and the plain launch with the transposed
    packed weights that is its input gradient (pose_hrnet.py:334-340 and its autograd)"""
    hh = _h()
    from hipnet import _capi as C
    dtype = torch.bfloat16
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(3 + Cin + Cout + N)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    w = _q(torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin), dtype)
    b = torch.randn(Cout, generator=g)
    d = hh.DEV
    xd = hh.nhwc(x, dtype)
    wp, _, _ = hh.pack_weights(w, dtype)
    y = torch.full((N, H, W, Cout), float('nan'), dtype=dtype, device=d)
    sums = torch.zeros(8, 2, Cout, device=d)
    bd = b.to(d)
    C.call('hrnet_conv2d_bnref', hh.dt_id(dtype), xd.data_ptr(), wp.data_ptr(), None, None, None, 0.0, 0.0, bd.data_ptr(),
           y.data_ptr(), sums.data_ptr(), N, H, W, Cin, H, W, Cout, 1, 1, 0, C.stream_ptr())
    ref = F.conv2d(x, w, b)
    got = hh.from_nhwc(y)
    assert not torch.isnan(got).any()
    assert hh.rel_err(got, ref) <= TOL[dtype]
    s = sums.double().sum(0).cpu()
    assert hh.rel_err(s[0], ref.double().sum((0, 2, 3))) <= 5 * TOL[dtype] + 1e-3
    assert hh.rel_err(s[1], (ref.double() ** 2).sum((0, 2, 3))) <= 5 * TOL[dtype]
    # input gradient: dx = dy * W as a 1x1 conv with the transposed packed weights, no bias, no statistics
    dy = _q(torch.randn(N, Cout, H, W, generator=g), dtype)
    wd, _, _ = hh.pack_weights(w, dtype, mode=1)
    dx = torch.full((N, H, W, Cin), float('nan'), dtype=dtype, device=d)
    dyd = hh.nhwc(dy, dtype)
    C.call('hrnet_conv2d', hh.dt_id(dtype), dyd.data_ptr(), wd.data_ptr(), None, None, None, dx.data_ptr(), None,
           N, H, W, Cout, H, W, Cin, 1, 1, 0, 0, 0, C.stream_ptr())
    want = F.conv_transpose2d(dy, w)
    assert hh.rel_err(hh.from_nhwc(dx), want) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 64, 64, 128, 384, 3), (1, 64, 64, 128, 384, 24), (2, 20, 28, 32, 48, 6), (1, 16, 16, 64, 32, 18)])
def test_dilated_conv_as_nine_displaced_pointwise_launches(dtype, shape):
    """hrnet_conv2d_dilated3x3 against F.conv2d(dilation=d, padding=d): the offset-generating convs of
    pose_hrnet_PoseAggr (reference lib/models/pose_hrnet_PoseAggr.py:497-506; 128 -> 21*18 channels, d = 3..24,
    including d larger than the map borders reach)"""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, dil = shape
    g = torch.Generator().manual_seed(9 + Cin + Cout + dil)
    x = _q(torch.randn(N, Cin, H, W, generator=g), dtype)
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9), dtype)
    ref = F.conv2d(x, w, None, padding=dil, dilation=dil)
    xd = hh.nhwc(x, dtype)
    taps = [hh.pack_weights(w[:, :, t // 3, t % 3].reshape(Cout, Cin, 1, 1).contiguous(), dtype)[0] for t in range(9)]
    wt = torch.stack(taps)                       # [9][Cout_pad * Cin_pad]
    y = torch.full((N, H, W, Cout), float('nan'), dtype=dtype, device=hh.DEV)
    C.call('hrnet_conv2d_dilated3x3', hh.dt_id(dtype), xd.data_ptr(), wt.data_ptr(), wt.stride(0) * wt.element_size(),
           y.data_ptr(), N, H, W, Cin, Cout, dil, C.stream_ptr())
    got = hh.from_nhwc(y)
    assert not torch.isnan(got).any()
    # bf16: the running sum is rounded after every tap (nine roundings of y)
    assert hh.rel_err(got, ref) <= (TOL[dtype] if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_pack_table_matches_the_per_layer_pack(dtype):
    """hrnet_pack_weights_table (every conv of the network in one launch, rows staged through LDS) against
    hrnet_pack_weights layer by layer, bit for bit: forward layout, transposed/flipped input-gradient layout and
    the flattened stem, with padded and unpadded channel counts (17 joints -> 32, 3 -> 8, 480, 48-wide branches)."""
    import ctypes
    hh = _h()
    from hipnet import _capi as C
    g = torch.Generator().manual_seed(77)
    shapes = [(64, 3, 3, 2), (32, 32, 3, 0), (32, 32, 3, 1), (64, 256, 1, 0), (64, 256, 1, 1), (17, 480, 1, 0),
              (17, 480, 1, 1), (480, 480, 1, 1), (384, 384, 3, 1), (384, 192, 3, 0), (48, 96, 3, 1), (21, 20, 3, 0),
              (21, 20, 3, 1)]
    ws, outs, wants = [], [], []
    ents = (C.HrPackEnt * len(shapes))()
    block = 0
    for e, (co, ci, ks, mode) in zip(ents, shapes):
        w = torch.randn(co, ci, ks, ks, generator=g)
        want, cop, cip = hh.pack_weights(w, dtype, mode=mode)
        wd = w.to(hh.DEV).contiguous()
        out = torch.full_like(want, float('nan'))
        ws.append(wd); outs.append(out); wants.append(want)
        e.w, e.out = wd.data_ptr(), out.data_ptr()
        e.Cout, e.Cin, e.ks, e.Cout_pad, e.Cin_pad, e.mode, e.block0 = co, ci, ks, cop, cip, mode, block
        block += C.call('hrnet_pack_blocks', cop, cip, ks, mode)
    raw = bytes(ctypes.string_at(ctypes.addressof(ents), ctypes.sizeof(ents)))
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(hh.DEV)
    C.call('hrnet_pack_weights_table', hh.dt_id(dtype), table.data_ptr(), len(shapes), block, C.stream_ptr())
    for shp, out, want in zip(shapes, outs, wants):
        assert torch.equal(out.float().cpu(), want.float().cpu()), shp


@pytest.mark.parametrize('dtype', DTYPES)
def test_consumer_side_batchnorm_conv_sum_and_table_finalize(dtype):
    """hrnet_conv2d_bnref / hrnet_sum_terms_bnref / hrnet_bn_finalize_table: a conv accumulates its batch sums with
    float atomics; the next conv and a residual sum read that output through BatchNorm(+ReLU) built on the fly from
    the sums; one table launch produces scale/shift/mean/invstd + running statistics. Reference: torch BatchNorm2d
    in training mode (pose_hrnet.py:41-57 conv-bn-relu-conv / bn + residual)."""
    import ctypes
    hh = _h()
    from hipnet import _capi as C
    N, H, W, C1, C2 = 40, 64, 64, 32, 32        # 640 pixel tiles: the multi-tile walk, 320 workgroups adding into 8 copies
    g = torch.Generator().manual_seed(23)
    x = _q(torch.randn(N, C1, H, W, generator=g), dtype)
    w1 = _q(torch.randn(C2, C1, 3, 3, generator=g) / 17.0, dtype)
    w2 = _q(torch.randn(C2, C2, 3, 3, generator=g) / 17.0, dtype)
    bn = torch.nn.BatchNorm2d(C2)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C2, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(C2, generator=g) - 0.5)
    bn.train()
    y1 = F.conv2d(x, w1, None, padding=1)
    a1 = _q(F.relu(bn(_q(y1, dtype))).detach(), dtype)
    y2 = F.conv2d(a1, w2, None, padding=1)
    res = F.relu(F.batch_norm(_q(y1, dtype), None, None, bn.weight, bn.bias, True) + x)   # relu(bn(y1) + x)
    d = hh.DEV
    wp1, _, _ = hh.pack_weights(w1, dtype)
    wp2, _, _ = hh.pack_weights(w2, dtype)
    xd = hh.nhwc(x, dtype)
    sums1 = torch.zeros(8, 2, C2, device=d)
    sums2 = torch.zeros(8, 2, C2, device=d)
    y1d = torch.empty(N, H, W, C2, dtype=dtype, device=d)
    y2d = torch.empty(N, H, W, C2, dtype=dtype, device=d)
    gam, bet = bn.weight.detach().to(d), bn.bias.detach().to(d)
    cnt = float(N * H * W)
    C.call('hrnet_conv2d_bnref', hh.dt_id(dtype), xd.data_ptr(), wp1.data_ptr(), None, None, None, 0.0, 0.0, None,
           y1d.data_ptr(), sums1.data_ptr(), N, H, W, C1, H, W, C2, 3, 1, 0, C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(y1d), y1) <= TOL[dtype]
    s = sums1.double().sum(0).cpu()
    assert hh.rel_err(s[1], (y1.double() ** 2).sum((0, 2, 3))) <= 5 * TOL[dtype]
    # consumer conv: relu(bn(y1)) built from the sums on the fly
    C.call('hrnet_conv2d_bnref', hh.dt_id(dtype), y1d.data_ptr(), wp2.data_ptr(), sums1.data_ptr(), gam.data_ptr(),
           bet.data_ptr(), 1.0 / cnt, 1e-5, None, y2d.data_ptr(), sums2.data_ptr(), N, H, W, C2, H, W, C2, 3, 1, 1,
           C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(y2d), y2) <= 2 * TOL[dtype] + (0 if dtype == torch.float32 else 1e-2)
    # consumer sum: relu(bn(y1) + x)
    out = torch.empty(N, H, W, C2, dtype=dtype, device=d)
    srcs = [y1d, xd]
    inv = (ctypes.c_float * 4)(1.0 / cnt, 0.0, 0.0, 0.0)
    gb = torch.cat([gam, bet])                      # gamma with beta = gamma + C, as the flat parameter buffer holds them
    C.call('hrnet_sum_terms_bnref', hh.dt_id(dtype), out.data_ptr(), N, H, W, C2, 2, hh.ptr_array(srcs),
           hh.ptr_array([sums1, None]), hh.ptr_array([gb, None]), hh.int_array([0, 0]), hh.int_array([0, 0]), 1, 1,
           inv, 1e-5, C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(out), res.detach()) <= 2 * TOL[dtype]
    # table finalize: arrays for the backward pass + running statistics
    rm, rv = torch.zeros(C2, device=d), torch.ones(C2, device=d)
    nbt = torch.zeros((), dtype=torch.int64, device=d)
    scale, shift, mean, invstd = (torch.empty(C2, device=d) for _ in range(4))
    ent = (C.HrBnEnt * 1)()
    e = ent[0]
    e.sums, e.gamma, e.beta = sums1.data_ptr(), gam.data_ptr(), bet.data_ptr()
    e.running_mean, e.running_var, e.num_batches_tracked = rm.data_ptr(), rv.data_ptr(), nbt.data_ptr()
    e.scale, e.shift, e.mean, e.invstd = scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    e.count, e.momentum, e.eps, e.C, e.block0 = cnt, 0.1, 1e-5, C2, 0
    raw = bytes(ctypes.string_at(ctypes.addressof(ent), ctypes.sizeof(ent)))
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(d)
    C.call('hrnet_bn_finalize_table', table.data_ptr(), 1, 1, C.stream_ptr())
    ref_bn = torch.nn.BatchNorm2d(C2)
    ref_bn.train()
    ref_bn(y1)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(rm.cpu().numpy(), ref_bn.running_mean.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(rv.cpu().numpy(), ref_bn.running_var.numpy(), rtol=tol, atol=tol)
    assert int(nbt.item()) == 1
    m = y1.mean((0, 2, 3))
    r = 1.0 / torch.sqrt(y1.var((0, 2, 3), unbiased=False) + 1e-5)
    np.testing.assert_allclose(mean.cpu().numpy(), m.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(invstd.cpu().numpy(), r.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(scale.cpu().numpy(), (bn.weight.detach() * r).numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize('dtype', DTYPES)
def test_sum_terms_fuse_with_nearest_upsampling(dtype):
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cc = 2, 16, 16, 32
    g = torch.Generator().manual_seed(9)
    a = _q(torch.randn(N, Cc, H, W, generator=g), dtype)
    b = _q(torch.randn(N, Cc, H // 2, W // 2, generator=g), dtype)
    c = _q(torch.randn(N, Cc, H // 4, W // 4, generator=g), dtype)
    sb, tb = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    sc, tc = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    aff = lambda t, s, o: t * s.view(1, -1, 1, 1) + o.view(1, -1, 1, 1)
    ref = F.relu(a + F.interpolate(aff(b, sb, tb), scale_factor=2, mode='nearest')
                 + F.interpolate(F.relu(aff(c, sc, tc)), scale_factor=4, mode='nearest'))
    d = hh.DEV
    ts = [hh.nhwc(a, dtype), hh.nhwc(b, dtype), hh.nhwc(c, dtype)]
    scs = [None, sb.to(d), sc.to(d)]
    shs = [None, tb.to(d), tc.to(d)]
    out = torch.empty(N, H, W, Cc, dtype=dtype, device=d)
    C.call('hrnet_sum_terms', hh.dt_id(dtype), out.data_ptr(), N, H, W, Cc, 3, hh.ptr_array(ts), hh.ptr_array(scs),
           hh.ptr_array(shs), hh.int_array([0, 1, 2]), hh.int_array([0, 0, 1]), 1, C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(out), ref) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('sh', [0, 1, 3])
def test_batchnorm_backward_through_upsampled_sum(dtype, sh):
    """out = relu(ident + up(bn(y))): d(y), d(gamma), d(beta) vs autograd."""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cc = 2, 4, 4, 32
    f = 1 << sh
    g = torch.Generator().manual_seed(13 + sh)
    y = _q(torch.randn(N, Cc, H, W, generator=g), dtype).requires_grad_(True)
    ident = _q(torch.randn(N, Cc, H * f, W * f, generator=g), dtype)
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(Cc, generator=g) - 0.5)
    bn.train()
    z = bn(y)
    up = F.interpolate(z, scale_factor=f, mode='nearest') if sh else z
    out = F.relu(ident + up)
    gout = _q(torch.randn(out.shape, generator=g), dtype)
    out.backward(gout)
    d = hh.DEV
    mean = y.detach().mean((0, 2, 3))
    var = y.detach().var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = (bn.weight.detach() * invstd).to(d)
    shift = (bn.bias.detach() - mean * bn.weight.detach() * invstd).to(d)
    yd, gd, od = hh.nhwc(y.detach(), dtype), hh.nhwc(gout, dtype), hh.nhwc(_q(out.detach(), dtype), dtype)
    blocks = C.call('hrnet_reduce_blocks', N, H, W, Cc)
    part = torch.empty(blocks, 2, Cc, device=d)
    C.call('hrnet_bn_bwd_reduce', hh.dt_id(dtype), part.data_ptr(), gd.data_ptr(), od.data_ptr(), yd.data_ptr(),
           scale.data_ptr(), shift.data_ptr(), N, H, W, Cc, sh, 0, C.stream_ptr())
    dgam, dbet, coef = torch.zeros(Cc, device=d), torch.zeros(Cc, device=d), torch.empty(3 * Cc, device=d)
    gam_d, mean_d, invstd_d = bn.weight.detach().to(d), mean.to(d), invstd.to(d)   # keep alive across the call
    C.call('hrnet_bn_bwd_finalize', part.data_ptr(), blocks, Cc, float(N * H * W), gam_d.data_ptr(),
           mean_d.data_ptr(), invstd_d.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), 0,
           C.stream_ptr())
    dy = torch.empty(N, H, W, Cc, dtype=dtype, device=d)
    C.call('hrnet_grad_term', hh.dt_id(dtype), dy.data_ptr(), gd.data_ptr(), od.data_ptr(), yd.data_ptr(),
           scale.data_ptr(), shift.data_ptr(), coef.data_ptr(), N, H, W, Cc, sh, 0, 0, C.stream_ptr())
    tol = 5 * TOL[dtype]
    assert hh.rel_err(dgam.cpu(), bn.weight.grad) <= tol
    assert hh.rel_err(dbet.cpu(), bn.bias.grad) <= tol
    assert hh.rel_err(hh.from_nhwc(dy), y.grad) <= tol


@pytest.mark.parametrize('dtype', DTYPES)
def test_grad_term_large_tensor_variant(dtype):
    """tensors of >= 4 grid-stride steps per thread take grad_term_rows_kernel (per-channel coefficients kept
    in registers): BatchNorm + ReLU backward apply on a head-sized 480-channel tensor, vs the formula"""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cc = 18, 64, 64, 480
    g = torch.Generator().manual_seed(17)
    gr = _q(torch.randn(N, H, W, Cc, generator=g), dtype)
    y = _q(torch.randn(N, H, W, Cc, generator=g), dtype)
    mask = _q(torch.randn(N, H, W, Cc, generator=g), dtype)
    prev = _q(torch.randn(N, H, W, Cc, generator=g), dtype)
    sc, sf = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    coef = torch.cat([torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1])
    dz = gr * (mask > 0) * ((y * sc + sf) > 0)
    want = prev + coef[:Cc] * dz + coef[Cc:2 * Cc] * y + coef[2 * Cc:]
    d = hh.DEV
    dst = prev.to(dtype).to(d)
    gd, yd, md = gr.to(dtype).to(d), y.to(dtype).to(d), mask.to(dtype).to(d)
    scd, sfd, cd = sc.to(d), sf.to(d), coef.to(d)
    C.call('hrnet_grad_term', hh.dt_id(dtype), dst.data_ptr(), gd.data_ptr(), md.data_ptr(), yd.data_ptr(), scd.data_ptr(),
           sfd.data_ptr(), cd.data_ptr(), N, H, W, Cc, 0, 1, 1, C.stream_ptr())
    assert hh.rel_err(dst.float().cpu(), want) <= (1e-5 if dtype == torch.float32 else 2e-2)
    # the two-destination form (BatchNorm term + identity term of one residual sum)
    dst1 = torch.empty_like(dst)
    dst2 = prev.to(dtype).to(d)
    C.call('hrnet_grad_term2', hh.dt_id(dtype), dst1.data_ptr(), dst2.data_ptr(), gd.data_ptr(), md.data_ptr(), yd.data_ptr(),
           None, None, cd.data_ptr(), N, H, W, Cc, 0, 1, C.stream_ptr())
    dz2 = gr * (mask > 0)
    assert hh.rel_err(dst1.float().cpu(), coef[:Cc] * dz2 + coef[Cc:2 * Cc] * y + coef[2 * Cc:]) <= (1e-5 if dtype == torch.float32 else 2e-2)
    assert hh.rel_err(dst2.float().cpu(), prev + dz2) <= (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('mode', ['bn_relu', 'bn_plain', 'sum_mask'])
@pytest.mark.parametrize('case', [(2, 16, 16, 32, 32, 3, 1), (2, 16, 16, 64, 128, 3, 2), (2, 8, 8, 256, 64, 1, 1),
                                  (2, 16, 16, 256, 64, 1, 1)])     # wide 1x1 output: its own tile choice
def test_dgrad_epilogue_gathers_batchnorm_backward_sums(dtype, mode, case):
    """hrnet_conv2d_bwdstats: the input-gradient conv also leaves (sum dz, sum dz*y) rows of the
    BatchNorm behind its output; checked against the same sums taken from the stored gradient."""
    hh = _h()
    from hipnet import _capi as C
    N, H, W, Cin, Cout, ks, stride = case
    g = torch.Generator().manual_seed(31)
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
    rows = torch.full((rows_n, 2, Cin), float('nan'), device=d)
    C.call('hrnet_conv2d_bwdstats', hh.dt_id(dtype), dyd.data_ptr(), wd.data_ptr(), gx.data_ptr(), rows.data_ptr(),
           yd.data_ptr(), od.data_ptr() if mode == 'sum_mask' else None,
           scd.data_ptr() if mode == 'bn_relu' else None, sfd.data_ptr() if mode == 'bn_relu' else None,
           N, Ho, Wo, Cout, H, W, Cin, ks, stride, 1 if stride == 2 else 0, 1, C.stream_ptr())
    v = hh.from_nhwc(gx).double()                      # finished gradient as stored
    # the plain input gradient is still right
    x = torch.zeros(N, Cin, H, W, requires_grad=True)
    F.conv2d(x, w, None, stride=stride, padding=ks // 2).backward(dy)
    assert hh.rel_err(v - prev.double(), x.grad) <= 2 * TOL[dtype]
    if mode == 'bn_relu':
        m = (yraw * sc.view(1, -1, 1, 1) + sf.view(1, -1, 1, 1)) > 0
    elif mode == 'sum_mask':
        m = outv > 0
    else:
        m = torch.ones_like(yraw, dtype=torch.bool)
    dz = v * m
    want = torch.stack([dz.sum((0, 2, 3)), (dz * yraw.double()).sum((0, 2, 3))])
    got = rows.double().sum(0).cpu()
    assert not torch.isnan(got).any()
    # the epilogue sums the f32 value before it is rounded for storage; `want` uses the stored one
    scale = dz.abs().sum((0, 2, 3)).max().item()
    assert float((got - want).abs().max()) <= 3 * TOL[dtype] * scale


@pytest.mark.parametrize('align', [False, True])
@pytest.mark.parametrize('dtype', DTYPES)
def test_bilinear_concat_forward_backward(dtype, align):
    hh = _h()
    from hipnet import _capi as C
    N, H, W = 2, 16, 16
    cs = [32, 64, 128, 256]
    g = torch.Generator().manual_seed(21)
    xs = [_q(torch.randn(N, c, H >> j, W >> j, generator=g), dtype).requires_grad_(True) for j, c in enumerate(cs)]
    ups = [xs[0]] + [F.interpolate(t, size=(H, W), mode='bilinear', align_corners=align) for t in xs[1:]]
    cat = torch.cat(ups, 1)
    gcat = _q(torch.randn(cat.shape, generator=g), dtype)
    cat.backward(gcat)
    d = hh.DEV
    xd = [hh.nhwc(t.detach(), dtype) for t in xs]
    out = torch.empty(N, H, W, sum(cs), dtype=dtype, device=d)
    hs, ws = [H >> j for j in range(4)], [W >> j for j in range(4)]
    C.call('hrnet_bilinear_cat', hh.dt_id(dtype), out.data_ptr(), hh.ptr_array(xd), hh.int_array(hs),
           hh.int_array(ws), hh.int_array(cs), 4, N, H, W, 1 if align else 0, C.stream_ptr())
    assert hh.rel_err(hh.from_nhwc(out), cat.detach()) <= TOL[dtype]
    dxs = [torch.empty_like(t) for t in xd]
    gcd = hh.nhwc(gcat, dtype)
    C.call('hrnet_bilinear_cat_bwd', hh.dt_id(dtype), gcd.data_ptr(), hh.ptr_array(dxs),
           hh.int_array(hs), hh.int_array(ws), hh.int_array(cs), 4, N, H, W, 1 if align else 0, 0, C.stream_ptr())
    for t, dx in zip(xs, dxs):
        assert hh.rel_err(hh.from_nhwc(dx), t.grad) <= 2 * TOL[dtype]


def test_layout_conversions_and_bias_grad():
    hh = _h()
    from hipnet import _capi as C
    for dtype in DTYPES:
        N, H, W, Cp, Cc = 2, 8, 12, 32, 21
        g = torch.Generator().manual_seed(2)
        x = _q(torch.randn(N, Cc, H, W, generator=g), dtype)
        xd = hh.nhwc(x, dtype, cpad=Cp)
        out = torch.empty(N, Cc, H, W, device=hh.DEV)
        C.call('hrnet_nhwc_to_nchw', hh.dt_id(dtype), xd.data_ptr(), out.data_ptr(), N, H, W, Cp, Cc, C.stream_ptr())
        assert torch.equal(out.cpu(), x)
        back = torch.full((N, H, W, Cp), 7.0, dtype=dtype, device=hh.DEV)
        xdev = x.to(hh.DEV)
        C.call('hrnet_nchw_to_nhwc', hh.dt_id(dtype), xdev.data_ptr(), back.data_ptr(), N, H, W, Cp, Cc,
               C.stream_ptr())
        assert torch.equal(back.float().cpu()[..., :Cc], x.permute(0, 2, 3, 1))
        assert float(back.float().abs()[..., Cc:].max()) == 0.0
        blocks = C.call('hrnet_reduce_blocks', 1, 1, N * H * W, Cp)
        scratch = torch.empty(blocks * Cp, device=hh.DEV)
        db = torch.ones(Cc, device=hh.DEV)
        C.call('hrnet_bias_grad', hh.dt_id(dtype), xd.data_ptr(), db.data_ptr(), scratch.data_ptr(), N * H * W, Cp, Cc,
               1, C.stream_ptr())
        np.testing.assert_allclose(db.cpu().numpy(), 1.0 + x.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)


def test_losses_and_decode_against_oracle(golden_dir):
    """HeatmapLoss / JointsMSELoss / get_final_preds kernels vs the oracle and the reference fixture."""
    import os
    hh = _h()
    from hipnet import _capi as C
    from oracle import hrnet_cpu as O
    gz = np.load(os.path.join(golden_dir, 'micro.npz'))
    d = hh.DEV
    p, t = torch.from_numpy(gz['hl_pred']), torch.from_numpy(gz['hl_gt'])
    B, K, H, W = p.shape
    pd, td = p.to(d), t.to(d)
    for mode, key in ((0, 'hl_l2'), (1, 'hl_l1')):
        part, loss = torch.empty(B * K, device=d), torch.empty(1, device=d)
        C.call('hrnet_heatmap_loss_fwd', pd.data_ptr(), td.data_ptr(), part.data_ptr(), loss.data_ptr(), B * K, H * W,
               mode, C.stream_ptr())
        assert abs(loss.item() - float(gz[key])) <= 1e-5 * float(gz[key])
        pr = p.clone().requires_grad_(True)
        O.heatmap_loss(pr, t, 'l2' if mode == 0 else 'l1').backward()
        gout, dp = torch.full((1,), 1.0, device=d), torch.empty_like(pd)
        C.call('hrnet_heatmap_loss_bwd', pd.data_ptr(), td.data_ptr(), gout.data_ptr(), dp.data_ptr(), B * K, H * W, mode,
               C.stream_ptr())
        assert hh.rel_err(dp.cpu(), pr.grad) <= 1e-6
    # key-point loss
    pp, pg, vis = (torch.from_numpy(gz[k]) for k in ('jm_pred', 'jm_gt', 'jm_vis'))
    Bj, Kj = pp.shape[:2]
    for v, key in ((vis, 'jm_vis_loss'), (None, 'jm_novis_loss'), (torch.zeros(Bj, Kj), 'jm_allinvis_loss')):
        loss = torch.empty(1, device=d)
        vd = v.to(d) if v is not None else None
        ppd, pgd, one = pp.to(d), pg.to(d), torch.ones(1, device=d)   # keep device copies alive
        C.call('hrnet_joints_loss_fwd', ppd.data_ptr(), pgd.data_ptr(), C.ptr(vd), loss.data_ptr(), Bj, Kj,
               C.stream_ptr())
        assert abs(loss.item() - float(gz[key])) <= 1e-5 * max(float(gz[key]), 1.0)
        pr = pp.clone().requires_grad_(True)
        O.joints_mse_loss(pr, pg, v).backward()
        dp = torch.empty(Bj, Kj, 2, device=d)
        C.call('hrnet_joints_loss_bwd', ppd.data_ptr(), pgd.data_ptr(), C.ptr(vd), one.data_ptr(), dp.data_ptr(), Bj, Kj,
               C.stream_ptr())
        assert np.abs(dp.cpu().numpy() - pr.grad.numpy()).max() <= 1e-6
    # argmax decode incl. ties (bit-exact integer work) and the non-square H-stride rule
    hm = torch.from_numpy(gz['am_hm'])
    preds = torch.empty(hm.shape[0], hm.shape[1], 2, device=d)
    hmd = hm.to(d)
    C.call('hrnet_decode_argmax', hmd.data_ptr(), preds.data_ptr(), None, hm.shape[0] * hm.shape[1], hm.shape[2],
           hm.shape[3], 0, C.stream_ptr())
    assert np.array_equal(preds.cpu().numpy(), gz['am_pred'])
    ns = torch.randn(2, 3, 4, 6)
    preds = torch.empty(2, 3, 2, device=d)
    mv = torch.empty(2, 3, device=d)
    nsd = ns.to(d)
    C.call('hrnet_decode_argmax', nsd.data_ptr(), preds.data_ptr(), mv.data_ptr(), 6, 4, 6, 0, C.stream_ptr())
    assert torch.equal(preds.cpu(), O.get_final_preds(ns, use_softmax=False))
    C.call('hrnet_decode_argmax', nsd.data_ptr(), preds.data_ptr(), mv.data_ptr(), 6, 4, 6, 1, C.stream_ptr())
    rp, rm = O.get_max_preds(ns.numpy())
    assert np.array_equal(preds.cpu().numpy(), rp) and np.array_equal(mv.cpu().numpy()[..., None], rm)
    # expectation decode + backward
    hs = torch.rand(2, 21, 64, 64)
    preds = torch.empty(2, 21, 2, device=d)
    hsd = hs.to(d)
    C.call('hrnet_decode_expectation', hsd.data_ptr(), preds.data_ptr(), 42, 64, 64, C.stream_ptr())
    ref = O.get_final_preds(hs, use_softmax=True)
    assert hh.rel_err(preds.cpu(), ref) <= 1e-5
    hr = hs.clone().requires_grad_(True)
    gp = torch.randn(2, 21, 2)
    (O.get_final_preds(hr, True) * gp).sum().backward()
    dh = torch.empty_like(hs, device=d)
    gpd = gp.to(d)
    C.call('hrnet_decode_expectation_bwd', gpd.data_ptr(), dh.data_ptr(), 42, 64, 64, 0, C.stream_ptr())
    assert hh.rel_err(dh.cpu(), hr.grad) <= 1e-6


def test_final_preds_with_post_process_and_image_space_map():
    """core.inference.get_final_preds (reference lib/core/inference.py:49-85) vs the looped restatement"""
    from config import get_cfg_defaults
    from core.inference import get_final_preds
    from oracle import hrnet_cpu as O
    rng = np.random.default_rng(3)
    hms = rng.random((3, 21, 24, 20)).astype(np.float32)
    hms[0, 0] = 0.0                                   # all-zero map: prediction zeroed
    hms[1, 2, 0, 5] = 2.0                             # border maximum: no refinement
    center = rng.random((3, 2)).astype(np.float32) * 100 + 50
    scale = rng.random((3, 2)).astype(np.float32) + 0.5
    for pp in (False, True):
        cfg = get_cfg_defaults()
        cfg.TEST.POST_PROCESS = pp
        got, gmax = get_final_preds(cfg, hms, center, scale)
        want, wmax = O.final_preds_oracle(pp, hms, center, scale)
        assert np.array_equal(gmax, wmax)
        assert np.abs(got - want).max() <= 1e-3, np.abs(got - want).max()


def test_core_inference_matches_reference_fixture(golden_dir):
    """SURVEY 8 a14 / a15 pinned: core.inference.get_max_preds / get_final_preds (HIP arg-max kernel behind them)
    against outputs of the reference's own lib/core/inference.py:18-85 (tests/golden/make_golden_inference.py).
    Arg-max coordinates and maxima bit-exact (ties, never-positive and all-zero maps, non-square maps); image-space
    coordinates to 1e-3 of a pixel (the reference applies a float64 3-point affine, the product the closed-form
    similarity it equals without rotation)."""
    from config import get_cfg_defaults
    from core.inference import get_final_preds, get_max_preds
    g = np.load(os.path.join(golden_dir, 'inference_preds.npz'))
    for tag in ('sq', 'rect'):
        preds, maxvals = get_max_preds(g['hm_' + tag])
        assert np.array_equal(preds, g['preds_' + tag]) and np.array_equal(maxvals, g['maxvals_' + tag])
        assert preds.dtype == np.float32 and maxvals.shape == g['maxvals_' + tag].shape
    for pp in (0, 1):
        cfg = get_cfg_defaults()
        cfg.TEST.POST_PROCESS = bool(pp)
        got, gmax = get_final_preds(cfg, g['fp_hm'].copy(), g['fp_center'], g['fp_scale'])
        assert np.array_equal(gmax, g['fp_maxvals_pp%d' % pp])
        assert np.abs(got - g['fp_preds_pp%d' % pp]).max() <= 1e-3, np.abs(got - g['fp_preds_pp%d' % pp]).max()


def test_spatial_softmax_with_temperature_forward_backward():
    """pose_hrnet_softmax.py:520-524: softmax over each map times a (trainable) temperature"""
    from hipnet import _capi as C
    d = 'cuda:0'
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(3, 21, 24, 20, generator=g) * 3).requires_grad_(True)
    temp = torch.tensor(1.7, requires_grad=True)
    out = F.softmax(x.view(3, 21, -1) * temp, dim=2).view(x.shape)
    gout = torch.randn(x.shape, generator=g)
    out.backward(gout)
    xd, td, gd = x.detach().to(d), temp.detach().to(d).reshape(1), gout.to(d)
    od, dxd, part = torch.empty_like(xd), torch.empty_like(xd), torch.empty(63, device=d)
    C.call('hrnet_spatial_softmax_fwd', xd.data_ptr(), td.data_ptr(), od.data_ptr(), 63, 480, C.stream_ptr())
    C.call('hrnet_spatial_softmax_bwd', xd.data_ptr(), od.data_ptr(), gd.data_ptr(), td.data_ptr(), dxd.data_ptr(),
           part.data_ptr(), 63, 480, C.stream_ptr())
    assert float((od.cpu() - out.detach()).abs().max()) <= 1e-6
    assert abs(float(od.sum()) - 63.0) <= 1e-3
    assert float((dxd.cpu() - x.grad).abs().max()) <= 1e-5 * max(1.0, float(x.grad.abs().max()))
    assert abs(float(part.sum()) - float(temp.grad)) <= 1e-4 * max(1.0, abs(float(temp.grad)))


def test_device_side_targets_and_normalisation_match_the_host_pipeline():
    """Gaussian targets (reference target_generators.py:14-53, restated in hipnet/synth.py and
    pinned there against the reference-shaped fixtures) and ToTensor+Normalize, as single launches."""
    from dataset.target_generators import HeatmapGenerator, gaussian_targets, normalize_u8
    from hipnet import synth
    rng = np.random.default_rng(11)
    pose = (rng.random((5, 21, 2)) * 80 - 8).astype(np.float32)          # some joints fall outside
    pose[0, 0] = [0.2, 63.9]
    pose[0, 1] = [63.99, 0.0]
    vis = rng.random((5, 21, 1)) < 0.8
    want = synth.gaussian_heatmaps(pose, vis, 64, 64, 2)
    got = gaussian_targets(torch.from_numpy(pose).cuda(), torch.from_numpy(vis).cuda(), 64, 64, 2).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-7
    assert np.array_equal(got == 0, want == 0)
    gen = HeatmapGenerator(64, 21, sigma=2)
    j = np.concatenate([pose[1], vis[1].astype(np.float32)], axis=1)
    assert np.abs(gen(j).cpu().numpy() - want[1]).max() <= 1e-7
    want_rect = synth.gaussian_heatmaps(pose[:2], vis[:2], 48, 40, 2)      # non-square maps
    got_rect = gaussian_targets(torch.from_numpy(pose[:2]).cuda(), torch.from_numpy(vis[:2]).cuda(), 48, 40, 2)
    assert np.abs(got_rect.cpu().numpy() - want_rect).max() <= 1e-7
    u8 = torch.from_numpy(rng.integers(0, 256, (3, 32, 24, 3), dtype=np.uint8))
    ref = (u8.float() / 255.0 - torch.tensor(synth.IMAGENET_MEAN)) / torch.tensor(synth.IMAGENET_STD)
    out = normalize_u8(u8.cuda()).cpu()
    assert float((out - ref.permute(0, 3, 1, 2)).abs().max()) <= 1e-6


def test_decode_and_softmax_kernels_against_the_reference_centre_of_mass(golden_dir):
    """the HIP expectation decode and spatial-softmax kernels vs the reference's integrate_tensor_2d fixture"""
    from hipnet import _capi as C
    from utils.heatmap_decoding import get_final_preds
    g = np.load(os.path.join(golden_dir, 'decode_crosscheck.npz'))
    d = 'cuda:0'
    pos = torch.from_numpy(g['pos']).to(d)
    got = get_final_preds(pos, use_softmax=True).cpu().numpy()
    assert np.abs(got - g['coords_normalised']).max() <= 2e-5
    raw = torch.from_numpy(g['raw']).to(d)
    one = torch.ones(1, device=d)
    soft = torch.empty_like(raw)
    C.call('hrnet_spatial_softmax_fwd', raw.data_ptr(), one.data_ptr(), soft.data_ptr(), 63, 480, C.stream_ptr())
    assert np.abs(soft.cpu().numpy() - g['softmax_maps']).max() <= 1e-7
    got2 = get_final_preds(soft, use_softmax=True).cpu().numpy()
    assert np.abs(got2 - g['coords_softmax']).max() <= 2e-5


def test_adam_step_matches_torch():
    hh = _h()
    from hipnet import _capi as C
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(10000, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3, weight_decay=1e-4)
    d = hh.DEV
    p, m, v = p0.to(d), torch.zeros(10000, device=d), torch.zeros(10000, device=d)
    for step in range(1, 4):
        gr = torch.randn(10000, generator=g)
        p_ref.grad = gr.clone()
        opt.step()
        grd = gr.to(d)
        C.call('hrnet_adam_step', p.data_ptr(), grd.data_ptr(), m.data_ptr(), v.data_ptr(), 10000, 1e-3, 0.9,
               0.999, 1e-8, 1e-4, step, 1.0, C.stream_ptr())
    assert np.abs(p.cpu().numpy() - p_ref.detach().numpy()).max() <= 1e-6


def test_bad_arguments_raise_not_crash():
    hh = _h()
    from hipnet import _capi as C
    x = torch.zeros(1, 8, 8, 30, device=hh.DEV)
    with pytest.raises(RuntimeError, match='Cin'):
        C.call('hrnet_conv2d', 0, x.data_ptr(), x.data_ptr(), None, None, None, x.data_ptr(), None, 1, 8, 8, 30, 8, 8,
               32, 3, 1, 0, 0, 0, C.stream_ptr())
    with pytest.raises(RuntimeError, match='kernel size'):
        C.call('hrnet_conv2d', 0, x.data_ptr(), x.data_ptr(), None, None, None, x.data_ptr(), None, 1, 8, 8, 32, 8, 8,
               32, 5, 1, 0, 0, 0, C.stream_ptr())
