"""The head without its concat (reference lib/models/pose_hrnet.py:560-566): last_layer[0] applied to
cat(x0, up(x1), up(x2), up(x3)) equals W0 x0 + up(W1 x1) + up(W2 x2) + up(W3 x3) + bias, because a 1x1 convolution
commutes with bilinear upsampling. These tests pin every piece of that form against PyTorch's own ops on the CPU:
  hrnet_head_mix             against F.interpolate(mode='bilinear') + torch.cat + F.conv2d (the reference's lines),
  hrnet_upsample_bilinear_t  against autograd of F.interpolate,
  weight gradients / packed weights of a COLUMN SLICE of the 480x480 weight (row pitch) against the dense forms.
Tolerances: bf16 storage of the products t_j and of y (2^-8 relative per rounding), f32 accumulation everywhere."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = torch.bfloat16


def _h():
    import hip_helpers as hh
    return hh


def _C():
    from hipnet import _capi as C
    return C


def _pp(tensors):
    arr = (ctypes.c_void_p * max(1, len(tensors)))()
    for k, t in enumerate(tensors):
        arr[k] = t.data_ptr()
    return arr


def _ip(vals):
    return (ctypes.c_int * max(1, len(vals)))(*vals)


def _pack_slice(w, c0, c1, DT=DT):
    """forward layout [Cout][c1-c0] of the column slice, through the table packer's row pitch"""
    hh, C = _h(), _C()
    co, ci = w.shape[0], w.shape[1]
    wd = w.float().to(hh.DEV).contiguous()
    out = torch.empty(co * (c1 - c0), dtype=DT, device=hh.DEV)
    ent = (C.HrPackEnt * 1)()
    e = ent[0]
    e.w, e.out = wd.data_ptr() + 4 * c0, out.data_ptr()
    e.Cout, e.Cin, e.ks, e.Cout_pad, e.Cin_pad, e.mode, e.block0, e.ld = co, c1 - c0, 1, co, c1 - c0, 0, 0, ci
    blocks = C.call('hrnet_pack_blocks', co, c1 - c0, 1, 0)
    raw = bytes(ctypes.string_at(ctypes.addressof(ent), ctypes.sizeof(ent)))
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(hh.DEV)
    C.call('hrnet_pack_weights_table', C.dtype_id(DT), table.data_ptr(), 1, blocks, C.stream_ptr())
    hh.sync()
    return out


MIX_CASES = [
    # N, H, W, channels of the branches, Cout, align_corners
    (2, 64, 64, (32, 64, 128, 256), 480, False),      # the w32 head
    (1, 16, 24, (32, 16, 8, 8), 64, False),           # ragged pixel block (384 pixels = 3 blocks), small widths
    (3, 8, 8, (64, 32), 96, False),                   # two branches, 192 pixels: a partial last block
    (1, 32, 32, (32,), 48, False),                    # no low-resolution term: a plain 1x1 conv + statistics
    (2, 32, 32, (32, 64, 128, 256), 480, True),       # align_corners=True (pose_hrnet_softmax.py:499-503)
    (2, 48, 40, (48, 96, 192, 384), 720, False),      # the w48 head: K = 48 (a half-empty second K step), ragged tiles
]


@pytest.mark.parametrize('DT', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('rows_mode', [0, 1])
@pytest.mark.parametrize('case', MIX_CASES)
def test_head_mix_equals_upsample_cat_conv(case, rows_mode, DT):
    hh, C = _h(), _C()
    N, H, W, cs, Cout, align = case
    f32 = DT == torch.float32
    g = torch.Generator().manual_seed(11 + H + Cout + len(cs))
    xs = [torch.randn(N, c, H >> j, W >> j, generator=g) for j, c in enumerate(cs)]
    xs = [x.to(DT).float() for x in xs]                                   # the values the device tensors hold
    ctot = sum(cs)
    w = (torch.randn(Cout, ctot, 1, 1, generator=g) / np.sqrt(ctot)).to(DT).float()
    bias = torch.randn(Cout, generator=g) * 0.1
    # the reference's lines: F.upsample x3, torch.cat, last_layer[0]
    ups = [xs[0]] + [F.interpolate(x, size=(H, W), mode='bilinear', align_corners=align) for x in xs[1:]]
    ref = F.conv2d(torch.cat(ups, 1), w, bias)                            # f32 on the CPU
    # device: t_j at branch resolution, then the mix launch
    offs = np.concatenate([[0], np.cumsum(cs)])
    xd = [hh.nhwc(x, DT) for x in xs]
    ts = []
    for j in range(1, len(cs)):
        wj = _pack_slice(w, int(offs[j]), int(offs[j + 1]), DT)
        t, _ = hh.conv2d(xd[j], wj, N, H >> j, W >> j, cs[j], Cout, 1, 1, DT)
        ts.append(t)
    w0 = _pack_slice(w, 0, cs[0], DT)
    y = torch.full((N, H, W, Cout), float('nan'), dtype=DT, device=hh.DEV)
    nrows = C.call('hrnet_head_mix_rows', N, H, W)
    stats = torch.zeros((nrows if rows_mode else 8), 2, Cout, dtype=torch.float32, device=hh.DEV)
    bd = bias.to(hh.DEV)
    if C.call('hrnet_head_mix_supported', C.dtype_id(DT), cs[0], Cout) != 1:
        pytest.skip('the fp32 form takes up to 512 output channels (the engine keeps the concat form there)')
    C.call('hrnet_head_mix', C.dtype_id(DT), xd[0].data_ptr(), w0.data_ptr(), bd.data_ptr(), y.data_ptr(), stats.data_ptr(), rows_mode,
           _pp(ts), _ip([H >> j for j in range(1, len(cs))]), _ip([W >> j for j in range(1, len(cs))]), len(ts),
           N, H, W, cs[0], Cout, 1 if align else 0, C.stream_ptr())
    hh.sync()
    got = hh.from_nhwc(y)
    scale = float(ref.abs().max())
    # bf16: two roundings on the way (t_j, y), 2^-8 each, relative to the magnitudes involved; f32: summation order
    tol = 2e-5 if f32 else 1.2e-2
    assert float((got - ref).abs().max()) <= tol * scale, (float((got - ref).abs().max()), scale)
    # the statistics are those of the f32 values the kernel formed (before the bf16 store)
    s = stats.sum(0).cpu()
    cnt = N * H * W
    assert torch.allclose(s[0] / cnt, ref.mean((0, 2, 3)), atol=2e-3 * scale)
    assert torch.allclose(s[1] / cnt, (ref * ref).mean((0, 2, 3)), rtol=2e-2, atol=1e-4 * scale * scale)


def test_head_mix_rows_are_reproducible():
    """rows mode: one statistics row per workgroup, no atomics - two launches agree bit for bit"""
    hh, C = _h(), _C()
    N, H, W, C0, Cout = 2, 32, 32, 32, 480
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(N, H, W, C0, generator=g).to(DT).to(hh.DEV)
    t1 = torch.randn(N, H // 2, W // 2, Cout, generator=g).to(DT).to(hh.DEV)
    w0 = torch.randn(Cout, C0, generator=g).to(DT).to(hh.DEV)
    outs = []
    for _ in range(2):
        y = torch.empty(N, H, W, Cout, dtype=DT, device=hh.DEV)
        rows = torch.zeros(C.call('hrnet_head_mix_rows', N, H, W), 2, Cout, device=hh.DEV)
        C.call('hrnet_head_mix', 1, x0.data_ptr(), w0.data_ptr(), None, y.data_ptr(), rows.data_ptr(), 1,
               _pp([t1]), _ip([H // 2]), _ip([W // 2]), 1, N, H, W, C0, Cout, 0, C.stream_ptr())
        hh.sync()
        outs.append((y.view(torch.int16).clone(), rows.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


UPT_CASES = [
    # N, H, W, C, output sizes, align
    (2, 64, 64, 480, [(32, 32)], False),
    (2, 64, 64, 480, [(16, 16)], False),
    (2, 64, 64, 480, [(8, 8)], False),
    (2, 64, 64, 480, [(32, 32), (16, 16), (8, 8)], False),     # the head: three scales, one pass over G (tile form)
    (1, 48, 40, 72, [(24, 20), (12, 10), (6, 5)], False),      # w48-like maps: tiles that overhang the image
    (1, 32, 48, 64, [(8, 12)], False),        # non-square, scale 4
    (1, 24, 24, 40, [(9, 9)], False),         # non-integer scale (8/3): streamed form, clamped taps at both borders
    (2, 32, 32, 96, [(16, 16)], True),        # align_corners=True: streamed form
    (1, 16, 16, 32, [(16, 16)], False),       # same size: identity (streamed form)
]


@pytest.mark.parametrize('streamed', [0, 1])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('case', UPT_CASES)
def test_upsample_transpose_equals_autograd_of_interpolate(case, dtype, streamed):
    hh, C = _h(), _C()
    N, H, W, Cc, sizes, align = case
    g = torch.Generator().manual_seed(3 + H + sizes[0][0] + Cc)
    G = torch.randn(N, Cc, H, W, generator=g).to(dtype).float()
    refs = []
    for hs, ws in sizes:
        x = torch.zeros(N, Cc, hs, ws, requires_grad=True)
        F.interpolate(x, size=(H, W), mode='bilinear', align_corners=align).backward(G)
        refs.append(x.grad)
    gd = hh.nhwc(G, dtype)
    outs = [torch.full((N, hs, ws, Cc), float('nan'), dtype=dtype, device=hh.DEV) for hs, ws in sizes]
    C.call('hrnet_upsample_bilinear_t', C.dtype_id(dtype), gd.data_ptr(), _pp(outs), _ip([h for h, _ in sizes]),
           _ip([w for _, w in sizes]), len(sizes), N, H, W, Cc, 1 if align else 0, streamed, C.stream_ptr())
    hh.sync()
    for out, ref in zip(outs, refs):
        got = hh.from_nhwc(out)
        tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2e-6) * float(ref.abs().max())
        assert float((got - ref).abs().max()) <= tol, (float((got - ref).abs().max()), float(ref.abs().max()))


@pytest.mark.parametrize('atomic', [1, 0])
def test_weight_gradient_of_a_column_slice(atomic):
    """dW[:, c0:c1] of a 1x1 weight [Cout][Ctot]: the launch sees Cin = c1-c0 input channels and a gradient whose rows
    are Ctot floats apart (HR_OP_WGRAD i[15] / HrWredEnt.ld); everything outside the slice stays untouched"""
    hh, C = _h(), _C()
    N, H, W, Cin, Cout, Ctot, c0 = 3, 16, 16, 64, 480, 480, 32
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
    ref = hh.wgrad(x, dy, N, H, W, Cin, H, W, Cout, 1, 1, DT)[:, :, 0, 0]            # dense [Cout][Cin]
    base = torch.randn(Cout, Ctot, generator=g).to(hh.DEV)
    grad = base.clone()
    ns = C.call('hrnet_wgrad_splits', 1, N, H, W, Cout, Cin, 1, 1)
    op = C.HrOp()
    op.kind = C.OP_WGRAD
    for k, v in enumerate((1, N, H, W, Cin, H, W, Cout, 1, 1, 0, ns, atomic, Cout, Cin, Ctot if atomic else 0)):
        op.i[k] = v
    slabs = torch.zeros(ns, Cout, Cin, device=hh.DEV)
    dst = grad.data_ptr() + 4 * c0 if atomic else slabs.data_ptr()
    for k, t in enumerate((x.data_ptr(), dy.data_ptr(), None, None, dst)):
        op.p[k] = t
    C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
    if not atomic:
        ent = (C.HrWredEnt * 1)()
        e = ent[0]
        e.slabs, e.grad = slabs.data_ptr(), grad.data_ptr() + 4 * c0
        e.nsplit, e.Cout_pad, e.Cin_pad, e.ks, e.Cout, e.Cin, e.kflat, e.accumulate, e.block0, e.ld = (
            ns, Cout, Cin, 1, Cout, Cin, 0, 1, 0, Ctot)
        raw = bytes(ctypes.string_at(ctypes.addressof(ent), ctypes.sizeof(ent)))
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(hh.DEV)
        C.call('hrnet_wgrad_reduce_table', table.data_ptr(), 1, (Cout * Cin + 63) // 64, C.stream_ptr())
    hh.sync()
    want = base.clone()
    want[:, c0:c0 + Cin] += ref
    assert float((grad - want).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(grad[:, :c0], base[:, :c0]) and torch.equal(grad[:, c0 + Cin:], base[:, c0 + Cin:])


def test_packed_column_slice_equals_dense_pack_of_the_slice():
    hh = _h()
    g = torch.Generator().manual_seed(23)
    w = torch.randn(480, 480, 1, 1, generator=g)
    got = _pack_slice(w, 96, 224)
    ref, _, _ = hh.pack_weights(w[:, 96:224].contiguous(), DT, mode=0)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize('DT', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('shape', [(2, 64, 64, 480), (1, 24, 40, 96), (3, 16, 16, 720)])
def test_head_backward_fuses_last_layer_dgrad_with_the_batchnorm_backward(shape, DT):
    """hrnet_head_bwd against autograd's pieces written out: dz = W3^T dHM (last_layer[3], pose_hrnet.py:341-346),
    masked by the ReLU behind last_layer[1] (BatchNorm2d), then (mode 1) the BatchNorm-backward sums and (mode 2)
    G = A*dz + B*y + C - what a dgrad conv, a reduction and an apply pass over a stored dz did before"""
    hh, C = _h(), _C()
    N, H, W, Cc = shape
    if C.call('hrnet_head_mix_supported', C.dtype_id(DT), 32, Cc) != 1:
        pytest.skip('the fp32 form takes up to 512 channels')
    g = torch.Generator().manual_seed(29 + H + Cc)
    nj = 21
    dhm = torch.zeros(N, H, W, 32)
    dhm[..., :nj] = torch.randn(N, H, W, nj, generator=g)
    dhm = dhm.to(DT)
    w3 = (torch.randn(nj, Cc, 1, 1, generator=g) / np.sqrt(Cc)).to(DT).float()
    y = torch.randn(N, H, W, Cc, generator=g).to(DT)
    scale = torch.rand(Cc, generator=g) + 0.5
    shift = torch.randn(Cc, generator=g) * 0.3
    coef = torch.randn(3, Cc, generator=g)
    # reference
    dz = dhm.float()[..., :nj] @ w3[:, :, 0, 0]                       # [N,H,W,Cc]
    yf = y.float()
    dzm = torch.where(yf * scale + shift > 0, dz, torch.zeros(()))
    s1, s2 = dzm.sum((0, 1, 2)), (dzm * yf).sum((0, 1, 2))
    G = coef[0] * dzm + coef[1] * yf + coef[2]
    # device
    wT, _, _ = hh.pack_weights(w3, DT, mode=1, cout_pad=32)            # [Cin = Cc][Cout_pad = 32]
    dd, yd = dhm.to(hh.DEV), y.to(hh.DEV)
    sc, sf, cf = scale.to(hh.DEV), shift.to(hh.DEV), coef.to(hh.DEV).contiguous()
    rows = torch.full((C.call('hrnet_head_mix_rows', N, H, W), 2, Cc), float('nan'), device=hh.DEV)
    C.call('hrnet_head_bwd', C.dtype_id(DT), 1, dd.data_ptr(), wT.data_ptr(), yd.data_ptr(), rows.data_ptr(),
           sc.data_ptr(), sf.data_ptr(), None, 1, N, H, W, 32, Cc, C.stream_ptr())
    out = torch.full((N, H, W, Cc), float('nan'), dtype=DT, device=hh.DEV)
    C.call('hrnet_head_bwd', C.dtype_id(DT), 2, dd.data_ptr(), wT.data_ptr(), yd.data_ptr(), out.data_ptr(),
           sc.data_ptr(), sf.data_ptr(), cf.data_ptr(), 1, N, H, W, 32, Cc, C.stream_ptr())
    hh.sync()
    r = rows.double().sum(0).cpu()
    tol = 1e-4 * float(dzm.abs().sum((0, 1, 2)).max())                # f32 sums of exact products, another order
    assert float((r[0] - s1.double()).abs().max()) <= tol and float((r[1] - (dzm * yf).double().sum((0, 1, 2))).abs().max()) <= 3 * tol
    err = float((out.float().cpu() - G).abs().max())
    assert err <= (2.0 ** -8 if DT == torch.bfloat16 else 1e-5) * float(G.abs().max()), (err, float(G.abs().max()))
