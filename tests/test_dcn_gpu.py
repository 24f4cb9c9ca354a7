"""Deformable convolution on the HIP library vs the CPU oracle and the reference's own invariants
(lib/deformable_conv/test.py). Tolerances: f32 arithmetic, sums of <= a few hundred products:
1e-4 absolute on O(1..10) values; grad_input uses float atomics (order-dependent) like the
reference's col2im - except on the PoseAggr geometry, whose single-pass backward sums it in fixed point."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn_cpu as D

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(seed, B, C, H, W, Co, k, groups, DG, stride, pad, dil, scale=2.0):
    rng = np.random.default_rng(seed)
    Ho, Wo = D._out_size(H, W, k, k, (stride, stride), (pad, pad), (dil, dil))
    return dict(input=rng.standard_normal((B, C, H, W)).astype(np.float32),
                offset=(rng.standard_normal((B, DG * 2 * k * k, Ho, Wo)) * scale).astype(np.float32),
                weight=rng.standard_normal((Co, C // groups, k, k)).astype(np.float32),
                bias=rng.random(Co).astype(np.float32), stride=(stride, stride), padding=(pad, pad),
                dilation=(dil, dil), groups=groups, DG=DG)


def _run(c, with_grad=False, seed=9, im2col_step=64):
    from deformable_conv import DeformConvFunction
    t = {k: torch.from_numpy(c[k]).to(DEV).requires_grad_(with_grad) for k in ('input', 'offset', 'weight', 'bias')}
    out = DeformConvFunction.apply(t['input'], t['offset'], t['weight'], t['bias'], c['stride'], c['padding'],
                                   c['dilation'], c['groups'], c['DG'], im2col_step)
    if not with_grad:
        return out.cpu().numpy()
    go = np.random.default_rng(seed).standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(go).to(DEV))
    return out.detach().cpu().numpy(), go, {k: v.grad.cpu().numpy() for k, v in t.items()}


def test_zero_offset_equals_conv2d_with_groups():
    """reference test.py:37-69"""
    from deformable_conv import DeformConv
    torch.manual_seed(3)
    dcn = DeformConv(4, 4, (3, 3), stride=1, padding=1, dilation=1, groups=2, deformable_groups=1,
                     im2col_step=1).to(DEV)
    x = torch.randn(2, 4, 4, 4, device=DEV)
    off = torch.zeros(2, 18, 4, 4, device=DEV)
    out = dcn(x, off)
    ref = F.conv2d(x.cpu(), dcn.weight.detach().cpu(), dcn.bias.detach().cpu(), 1, 1, 1, 2)
    assert float((out.cpu() - ref).abs().max()) < 1e-5


def test_identity_kernel_is_exact():
    """reference test.py:113-141 (d < 1e-10: the op must be exact here)"""
    from deformable_conv import DeformConv
    dcn = DeformConv(4, 4, (3, 3), stride=1, padding=1, dilation=1, groups=2, deformable_groups=1).to(DEV)
    with torch.no_grad():
        dcn.weight.zero_()
        dcn.bias.zero_()
        for q in range(4):
            dcn.weight[q, q % 2, 1, 1] = 1.0
    x = torch.randn(2, 4, 4, 4, device=DEV)
    out = dcn(x, torch.zeros(2, 18, 4, 4, device=DEV))
    assert float((out - x).abs().max()) == 0.0


def test_im2col_step_does_not_change_the_result():
    """reference test.py:177-248"""
    c = _case(11, 2, 4, 4, 4, 4, 3, 2, 1, 1, 1, 1)
    a, _, ga = _run(c, True, im2col_step=1)
    b, _, gb = _run(c, True, im2col_step=2)
    assert np.array_equal(a, b)
    for k in ('offset', 'weight', 'bias'):
        assert np.array_equal(ga[k], gb[k]), k
    assert np.abs(ga['input'] - gb['input']).max() < 1e-5          # atomics: order may differ
    from deformable_conv import DeformConvFunction
    with pytest.raises(ValueError, match='im2col_step'):
        z = torch.zeros
        DeformConvFunction.apply(z(3, 4, 4, 4, device=DEV), z(3, 18, 4, 4, device=DEV), z(4, 4, 3, 3, device=DEV),
                                 None, 1, 1, 1, 1, 1, 2)


CASES = [
    # B, C, H, W, Co, k, groups, DG, stride, pad, dil
    (2, 4, 4, 4, 4, 3, 2, 1, 1, 1, 1),          # the reference test's shape
    (2, 4, 9, 7, 6, 3, 1, 2, 2, 1, 1),          # stride 2, ragged size
    (1, 6, 11, 13, 4, 3, 2, 3, 1, 2, 2),        # deformable groups straddle conv groups
    (3, 21, 16, 16, 21, 3, 1, 21, 1, 6, 6),     # PoseAggr geometry (pose_hrnet_PoseAggr.py:500-516)
    (1, 8, 10, 10, 40, 3, 1, 1, 1, 1, 1),       # > 32 output channels: two forward passes
    (2, 3, 8, 8, 5, 1, 1, 1, 1, 0, 1),          # 1x1 kernel
    (1, 2, 6, 20, 2, 5, 1, 2, 1, 2, 1),         # 5x5 kernel
    (1, 80, 6, 6, 8, 3, 1, 1, 1, 1, 1),         # 92 KB weight slice in LDS, global-atomics path
    (2, 6, 12, 12, 6, 3, 1, 2, 1, 1, 1),        # 3 channels per deformable group on the LDS path
    # forward with the planes staged through LDS (dcn_fwd_planes_kernel: >= 1024 output pixels per image)
    (2, 21, 64, 64, 21, 3, 1, 21, 1, 3, 3),     # PoseAggr at its full plane size
    (1, 6, 40, 41, 10, 3, 2, 3, 1, 1, 1),       # 1640 pixels: a partly idle second block; conv groups x shared offsets
    (1, 4, 80, 72, 8, 3, 1, 2, 2, 1, 1),        # stride 2: 5760-float planes, 1440 output pixels
]


@pytest.mark.parametrize('shape', CASES)
def test_forward_and_backward_match_the_oracle(shape):
    c = _case(21 + sum(shape), *shape)
    out, go, g = _run(c, True)
    c64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    ref = D.deform_conv_forward(**c64)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out - ref).max() < 2e-5 * scale
    gi, goff, gw, gb = D.deform_conv_backward(c64['input'], c64['offset'], c64['weight'], go.astype(np.float64),
                                              c['stride'], c['padding'], c['dilation'], c['groups'], c['DG'])
    # (the offset gradient is discontinuous where a sample position crosses a pixel centre: positions within f32 rounding
    # of an integer are left out, as in test_config5_backward_at_full_size - none at the small sizes, a handful of the
    # 3 M elements of the 64x64 case)
    B, C, H, W, Co, k, groups, DG, stride, pad, dil = shape
    Ho, Wo = D._out_size(H, W, k, k, c['stride'], c['padding'], c['dilation'])
    h, w_ = D._sample_positions(c64['offset'], (B, Ho, Wo), k, k, c['stride'], c['padding'], c['dilation'], DG)
    near = (np.abs(h - np.round(h)) < 2e-4) | (np.abs(w_ - np.round(w_)) < 2e-4)          # [B, DG, K, Ho, Wo]
    keep = ~np.repeat(near[:, :, :, None], 2, axis=3).reshape(goff.shape)
    assert keep.mean() > 0.99
    for name, want in (('input', gi), ('offset', goff), ('weight', gw), ('bias', gb)):
        s = max(1.0, float(np.abs(want).max()))
        m = keep if name == 'offset' else 1.0
        assert np.abs((g[name] - want) * m).max() < 5e-5 * s, name


_FWD_CHILD = '''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from deformable_conv import DeformConvFunction
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.standard_normal((3, 21, 64, 64)).astype(np.float32)).cuda()
w = torch.from_numpy((rng.standard_normal((21, 21, 3, 3)) * 0.1).astype(np.float32)).cuda()
b = torch.from_numpy(rng.random(21).astype(np.float32)).cuda()
off = torch.from_numpy((rng.standard_normal((3, 378, 64, 64)) * 7).astype(np.float32)).cuda()
out = DeformConvFunction.apply(x, off, w, b, 1, 6, 6, 1, 21, 64)
np.save(sys.argv[2], out.cpu().numpy())
'''


def test_forward_through_lds_planes_equals_the_gather_kernel_bit_for_bit(tmp_path):
    """dcn_fwd_planes_kernel (planes staged in LDS with a border of zeros, corners read without bounds tests) against
    dcn_fwd_kernel (four guarded L2 gathers per sample) on the PoseAggr geometry with offsets of several pixels - many
    samples on and beyond the image border: the same arithmetic in the same order, so the same bits. The gather kernel
    runs in a child process (its switch, HRNET_DCN_FWD_PLANES=0 with HRNET_MEASURE=1, is read once per process)."""
    import os
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'hrnet-hand-pose-estimation_amd', 'lib')
    outs = []
    for planes in ('1', '0'):
        f = str(tmp_path / ('fwd_planes_%s.npy' % planes))
        env = dict(os.environ, HRNET_MEASURE='1', HRNET_DCN_FWD_PLANES=planes)
        r = subprocess.run([sys.executable, '-c', _FWD_CHILD, lib, f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 0.1
    assert np.array_equal(outs[0], outs[1])


def test_large_offsets_fall_outside_and_read_zero():
    c = _case(5, 2, 4, 8, 8, 4, 3, 1, 1, 1, 1, 1, scale=20.0)
    out = _run(c)
    c64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    assert np.abs(out - D.deform_conv_forward(**c64)).max() < 1e-4
    c['offset'][:] = 1e6
    out = _run(c)
    assert np.array_equal(out, np.broadcast_to(c['bias'].reshape(1, -1, 1, 1), out.shape))


def test_poseaggr_scale_linearity_and_zero_offset():
    """config 5 size (B=64, 21 ch, 64x64, dilation 6): size-independent properties - zero offsets equal
    the dilated convolution, and the op is linear in the input for fixed offsets."""
    rng = np.random.default_rng(0)
    B = 64
    x = torch.from_numpy(rng.standard_normal((B, 21, 64, 64)).astype(np.float32)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((21, 21, 3, 3)) * 0.1).astype(np.float32)).to(DEV)
    from deformable_conv import DeformConvFunction
    zero = torch.zeros(B, 21 * 18, 64, 64, device=DEV)
    out = DeformConvFunction.apply(x, zero, w, None, 1, 6, 6, 1, 21, 64)
    ref = F.conv2d(x, w, None, 1, 6, 6)
    assert float((out - ref).abs().max()) < 1e-4
    off = torch.from_numpy((rng.standard_normal((B, 21 * 18, 64, 64)) * 3).astype(np.float32)).to(DEV)
    x2 = torch.from_numpy(rng.standard_normal((B, 21, 64, 64)).astype(np.float32)).to(DEV)
    f = lambda t: DeformConvFunction.apply(t, off, w, None, 1, 6, 6, 1, 21, 64)
    lin = f(x + 2 * x2) - (f(x) + 2 * f(x2))
    assert float(lin.abs().max()) < 1e-4


def test_config5_backward_at_full_size():
    """BASELINE config 5 at its FULL size (B=64, 21 channels, 64x64, deformable groups 21, dilation 12 - the middle
    dilation of pose_hrnet_PoseAggr.py:508-516), BACKWARD: (a) zero offsets: input / weight / bias gradients equal
    F.conv2d's autograd (reference test.py:37-69 applied to the gradients); (b) the backward is linear in grad_output
    for fixed offsets (reference test.py:262-302 checks the same pass for im2col_step invariance); (c) input and
    offset gradients of images 0-1 of the B=64 launch against the float64 oracle run on those two images (both are
    per-image quantities), the weight gradient of a two-image launch against the oracle's."""
    from deformable_conv import DeformConvFunction
    rng = np.random.default_rng(5)
    B, Cc, H, dil = 64, 21, 64, 12
    x = torch.from_numpy(rng.standard_normal((B, Cc, H, H)).astype(np.float32)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((Cc, Cc, 3, 3)) * 0.1).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.random(Cc).astype(np.float32)).to(DEV)
    go = torch.from_numpy(rng.standard_normal((B, Cc, H, H)).astype(np.float32)).to(DEV)
    go2 = torch.from_numpy(rng.standard_normal((B, Cc, H, H)).astype(np.float32)).to(DEV)

    def grads(offset, g, xs=x):
        t = [v.clone().requires_grad_(True) for v in (xs, offset, w, b)]
        out = DeformConvFunction.apply(t[0], t[1], t[2], t[3], 1, dil, dil, 1, Cc, 64)
        out.backward(g)
        return [v.grad for v in t]

    # (a) zero offsets
    zero = torch.zeros(B, Cc * 18, H, H, device=DEV)
    gx, _, gw, gb = grads(zero, go)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.conv2d(xr, wr, br, 1, dil, dil).backward(go)
    assert float((gx - xr.grad).abs().max()) <= 1e-4 * max(1.0, float(xr.grad.abs().max()))
    assert float((gw - wr.grad).abs().max()) <= 2e-4 * max(1.0, float(wr.grad.abs().max()))
    assert float((gb - br.grad).abs().max()) <= 2e-4 * max(1.0, float(br.grad.abs().max()))
    # (b) linearity in grad_output, offsets of a few pixels
    off = torch.from_numpy((rng.standard_normal((B, Cc * 18, H, H)) * 3).astype(np.float32)).to(DEV)
    g1, g2, g12 = grads(off, go), grads(off, go2), grads(off, go + 2 * go2)
    for name, a1, a2, a12 in zip(('input', 'offset', 'weight', 'bias'), g1, g2, g12):
        want = a1 + 2 * a2
        assert float((a12 - want).abs().max()) <= 2e-4 * max(1.0, float(want.abs().max())), name
    # (c) the oracle on images 0-1
    c = dict(input=x[:2].cpu().numpy().astype(np.float64), offset=off[:2].cpu().numpy().astype(np.float64),
             weight=w.cpu().numpy().astype(np.float64))
    gi, goff, gw_ref, gb_ref = D.deform_conv_backward(c['input'], c['offset'], c['weight'],
                                                      go[:2].cpu().numpy().astype(np.float64), (1, 1), (dil, dil),
                                                      (dil, dil), 1, Cc)
    # the offset gradient is the derivative of a bilinear sample: discontinuous where the sample position crosses a
    # pixel centre (or the -1 / H border rule, deform_im2col_cuda.cuh:173) - a position within f32 rounding of an integer
    # may fall on the other side in the float64 oracle (a handful of the 3 M elements at this size): those are left out
    h, w_ = D._sample_positions(c['offset'], (2, H, H), 3, 3, (1, 1), (dil, dil), (dil, dil), Cc)
    near = (np.abs(h - np.round(h)) < 2e-4) | (np.abs(w_ - np.round(w_)) < 2e-4)          # [B, DG, K, Ho, Wo]
    keep = ~np.repeat(near[:, :, :, None], 2, axis=3).reshape(2, Cc * 18, H, H)
    assert keep.mean() > 0.999
    got_in = g1[0][:2].cpu().numpy()
    assert np.abs(got_in - gi).max() <= 5e-5 * max(1.0, float(np.abs(gi).max())), 'input'
    got_off = g1[1][:2].cpu().numpy()
    assert np.abs((got_off - goff) * keep).max() <= 5e-5 * max(1.0, float(np.abs(goff).max())), 'offset'
    _, _, gw2, gb2 = grads(off[:2].contiguous(), go[:2].contiguous(), xs=x[:2].contiguous())
    assert np.abs(gw2.cpu().numpy() - gw_ref).max() <= 5e-5 * max(1.0, float(np.abs(gw_ref).max()))
    assert np.abs(gb2.cpu().numpy() - gb_ref).max() <= 5e-5 * max(1.0, float(np.abs(gb_ref).max()))


def test_poseaggr_input_gradient_is_reproducible_and_scale_free():
    """The single-pass backward of the PoseAggr geometry accumulates the scattered input gradient in a 64-bit fixed-point
    LDS plane (csrc/dcn.hip, dcn_bwd_fused_kernel): (a) two launches on the same operands give the SAME BITS (integer
    addition commutes; float atomics did not); (b) the plane's scale follows the operands - gradients scaled by 2^-60 and
    by 2^+40 give the same result scaled by that power of two, bit for bit (a fixed scale would flush the first to zero
    and overflow the second); (c) the result stays within the float64 oracle's tolerance (reference rule:
    deform_im2col_cuda.cuh:192-247, col2im by atomicAdd); (d) an all-zero grad_output gives zeros, an inf gives NaN
    planes for that image only."""
    from deformable_conv import DeformConvFunction
    rng = np.random.default_rng(11)
    B, Cc, H, dil = 6, 21, 64, 3
    x = torch.from_numpy(rng.standard_normal((B, Cc, H, H)).astype(np.float32)).to(DEV)
    w = torch.from_numpy((rng.standard_normal((Cc, Cc, 3, 3)) * 0.1).astype(np.float32)).to(DEV)
    off = torch.from_numpy((rng.standard_normal((B, Cc * 18, H, H)) * 4).astype(np.float32)).to(DEV)
    go = torch.from_numpy(rng.standard_normal((B, Cc, H, H)).astype(np.float32)).to(DEV)

    def gin(g):
        t = x.clone().requires_grad_(True)
        out = DeformConvFunction.apply(t, off, w, None, 1, dil, dil, 1, Cc, 64)
        out.backward(g)
        return t.grad

    a, b = gin(go), gin(go)
    assert torch.equal(a, b)
    for e in (-60, 40):
        sc = float(2.0 ** e)
        assert torch.equal(gin(go * sc), a * sc), e
    gi, _, _, _ = D.deform_conv_backward(x[:1].cpu().numpy().astype(np.float64), off[:1].cpu().numpy().astype(np.float64),
                                         w.cpu().numpy().astype(np.float64), go[:1].cpu().numpy().astype(np.float64),
                                         (1, 1), (dil, dil), (dil, dil), 1, Cc)
    assert np.abs(a[:1].cpu().numpy() - gi).max() <= 2e-5 * max(1.0, float(np.abs(gi).max()))
    assert float(gin(torch.zeros_like(go)).abs().max()) == 0.0
    bad = go.clone()
    bad[2, 5, 7, 9] = float('inf')
    gb = gin(bad)
    assert bool(torch.isnan(gb[2]).all()) and torch.equal(gb[[0, 1, 3, 4, 5]], a[[0, 1, 3, 4, 5]])


def test_bad_arguments_fail_loudly():
    from deformable_conv import DeformConvFunction
    z = torch.zeros
    with pytest.raises(ValueError, match='offset shape'):
        DeformConvFunction.apply(z(1, 4, 4, 4, device=DEV), z(1, 18, 5, 4, device=DEV), z(4, 4, 3, 3, device=DEV),
                                 None, 1, 1, 1, 1, 1, 64)
    with pytest.raises(TypeError):
        DeformConvFunction.apply(z(1, 4, 4, 4, device=DEV).half(), z(1, 18, 4, 4, device=DEV),
                                 z(4, 4, 3, 3, device=DEV), None, 1, 1, 1, 1, 1, 64)
    with pytest.raises(RuntimeError, match='deformable_groups'):
        DeformConvFunction.apply(z(1, 4, 4, 4, device=DEV), z(1, 54, 4, 4, device=DEV), z(4, 4, 3, 3, device=DEV),
                                 None, 1, 1, 1, 1, 3, 64)


# ---- modulated form (DCNv2; reference functions/modulated_deform_conv_func.py, modules/modulated_deform_conv.py) ------
def _mrun(c, with_grad=False, seed=9, im2col_step=64):
    from deformable_conv import ModulatedDeformConvFunction
    t = {k: torch.from_numpy(c[k]).to(DEV).requires_grad_(with_grad) for k in ('input', 'offset', 'mask', 'weight', 'bias')}
    out = ModulatedDeformConvFunction.apply(t['input'], t['offset'], t['mask'], t['weight'], t['bias'], c['stride'],
                                            c['padding'], c['dilation'], c['groups'], c['DG'], im2col_step)
    if not with_grad:
        return out.cpu().numpy()
    go = np.random.default_rng(seed).standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(go).to(DEV))
    return out.detach().cpu().numpy(), go, {k: v.grad.cpu().numpy() for k, v in t.items()}


def _mcase(seed, *shape, **kw):
    c = _case(seed, *shape, **kw)
    B, _, Ho, Wo = c['offset'].shape
    k = c['weight'].shape[2]
    c['mask'] = np.random.default_rng(seed + 7).random((B, c['DG'] * k * k, Ho, Wo)).astype(np.float32)
    return c


MCASES = [
    (2, 4, 4, 4, 4, 3, 2, 1, 1, 1, 1),          # the reference test's shape (test.py:69-110)
    (2, 4, 9, 7, 6, 3, 1, 2, 2, 1, 1),          # stride 2, ragged size, two deformable groups
    (1, 6, 11, 13, 4, 3, 2, 3, 1, 2, 2),        # deformable groups straddle conv groups
    (3, 21, 16, 16, 21, 3, 1, 21, 1, 6, 6),     # PoseAggr geometry
    (1, 8, 10, 10, 40, 3, 1, 1, 1, 1, 1),       # > 32 output channels
]


def test_modulated_forward_with_planes_through_lds():
    """the modulated forward at >= 1024 output pixels per image takes dcn_fwd_planes_kernel (the mask multiplies every
    sample there too): 1640 pixels, conv groups 2, three channels per deformable group, against the float64 oracle"""
    shape = (1, 6, 40, 41, 10, 3, 2, 2, 1, 1, 1)
    c = _mcase(77, *shape)
    out = _mrun(c)
    c64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    ref = D.modulated_deform_conv_forward(c64['input'], c64['offset'], c64['mask'], c64['weight'], c64['bias'],
                                          c['stride'], c['padding'], c['dilation'], c['groups'], c['DG'])
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize('shape', MCASES)
def test_modulated_forward_and_backward_match_the_oracle(shape):
    c = _mcase(31 + sum(shape), *shape)
    out, go, g = _mrun(c, True)
    c64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    ref = D.modulated_deform_conv_forward(c64['input'], c64['offset'], c64['mask'], c64['weight'], c64['bias'],
                                          c['stride'], c['padding'], c['dilation'], c['groups'], c['DG'])
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, float(np.abs(ref).max()))
    gi, goff, gm, gw, gb = D.modulated_deform_conv_backward(c64['input'], c64['offset'], c64['mask'], c64['weight'],
                                                            go.astype(np.float64), c['stride'], c['padding'],
                                                            c['dilation'], c['groups'], c['DG'])
    for name, want in (('input', gi), ('offset', goff), ('mask', gm), ('weight', gw), ('bias', gb)):
        s = max(1.0, float(np.abs(want).max()))
        assert np.abs(g[name] - want).max() < 5e-5 * s, name


def test_modulated_invariants_of_the_reference_test():
    """zero offsets + unit mask == nn.Conv2d incl. groups (test.py:69-110); unit mask == the v1 operator, forward and
    gradients; im2col_step changes nothing (test.py:219-260, 304-349); the Pack module starts as half a convolution"""
    from deformable_conv import DeformConvFunction, ModulatedDeformConvPack
    c = _mcase(41, 2, 4, 8, 8, 6, 3, 2, 2, 1, 1, 1)
    c['offset'][:] = 0
    c['mask'][:] = 1
    out = _mrun(c)
    ref = F.conv2d(torch.from_numpy(c['input']), torch.from_numpy(c['weight']), torch.from_numpy(c['bias']), 1, 1, 1, 2)
    assert np.abs(out - ref.numpy()).max() < 1e-5
    c = _mcase(42, 2, 4, 8, 8, 6, 3, 2, 2, 1, 1, 1)
    c['mask'][:] = 1
    a, go, ga = _mrun(c, True, im2col_step=1)
    b, _, gb = _mrun(c, True, im2col_step=2)
    assert np.array_equal(a, b)
    for k in ('offset', 'mask', 'weight', 'bias'):
        assert np.array_equal(ga[k], gb[k]), k
    v1, _, g1 = _run({k: v for k, v in c.items() if k != 'mask'}, True)
    assert np.abs(a - v1).max() < 1e-5
    for k in ('offset', 'weight', 'bias'):
        assert np.abs(ga[k] - g1[k]).max() < 1e-4 * max(1.0, float(np.abs(g1[k]).max())), k
    assert np.abs(ga['input'] - g1['input']).max() < 1e-4 * max(1.0, float(np.abs(g1['input']).max()))
    torch.manual_seed(0)
    m = ModulatedDeformConvPack(4, 6, 3, 1, 1, deformable_groups=2).to(DEV)
    assert sorted(k for k, _ in m.named_parameters()) == ['bias', 'conv_offset_mask.bias', 'conv_offset_mask.weight', 'weight']
    x = torch.randn(2, 4, 8, 8, device=DEV)
    want = 0.5 * F.conv2d(x, m.weight, None, 1, 1) + m.bias.view(1, -1, 1, 1)          # sigmoid(0) = 0.5 on every sample
    assert float((m(x) - want).abs().max()) < 1e-5
    with pytest.raises(ValueError, match='mask shape'):
        from deformable_conv import ModulatedDeformConvFunction
        z = torch.zeros
        ModulatedDeformConvFunction.apply(z(1, 4, 4, 4, device=DEV), z(1, 18, 4, 4, device=DEV), z(1, 8, 4, 4, device=DEV),
                                          z(4, 4, 3, 3, device=DEV), None, 1, 1, 1, 1, 1, 64)
