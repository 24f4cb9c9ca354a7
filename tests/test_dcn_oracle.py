"""Pin oracle/dcn_cpu.py by the invariants the reference's own lib/deformable_conv/test.py checks
(the reference op is CUDA-only, so these invariants are the available anchors). CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn_cpu as D


def _case(seed, B=2, C=4, H=5, W=6, Co=4, k=3, groups=2, DG=1, stride=1, pad=1, dil=1, scale=2.0):
    rng = np.random.default_rng(seed)
    Ho, Wo = D._out_size(H, W, k, k, (stride, stride), (pad, pad), (dil, dil))
    return dict(input=rng.standard_normal((B, C, H, W)),
                offset=rng.standard_normal((B, DG * 2 * k * k, Ho, Wo)) * scale,
                weight=rng.standard_normal((Co, C // groups, k, k)),
                bias=rng.random(Co), stride=(stride, stride), padding=(pad, pad), dilation=(dil, dil),
                groups=groups, DG=DG)


@pytest.mark.parametrize('groups,stride,pad,dil', [(2, 1, 1, 1), (1, 2, 1, 1), (1, 1, 3, 3), (2, 1, 0, 1)])
def test_zero_offset_is_plain_convolution(groups, stride, pad, dil):
    """reference test.py:37-69 (groups=2, d < 1e-5)"""
    c = _case(1, H=9, W=8, groups=groups, stride=stride, pad=pad, dil=dil)
    c['offset'][:] = 0
    out = D.deform_conv_forward(**c)
    ref = F.conv2d(torch.from_numpy(c['input']), torch.from_numpy(c['weight']), torch.from_numpy(c['bias']),
                   stride, pad, dil, groups).numpy()
    assert np.abs(out - ref).max() < 1e-10


def test_identity_kernel_returns_input():
    """reference test.py:113-141: centre-tap identity weights, zero offsets -> output == input"""
    c = _case(2, groups=2)
    c['offset'][:] = 0
    c['weight'][:] = 0
    c['bias'][:] = 0
    Co, Cg = c['weight'].shape[:2]
    for q in range(Co):
        c['weight'][q, q % (Co // 2), 1, 1] = 1.0
    assert np.abs(D.deform_conv_forward(**c) - c['input']).max() < 1e-12


def test_integer_offsets_shift_the_taps():
    """an integer offset moves every tap by whole pixels; samples at or beyond H read zero
    (deform_im2col_cuda.cuh:173)"""
    c = _case(3, groups=1, C=2, Co=3)
    c['offset'][:] = 0
    c['offset'][:, 0::2] = 1.0          # dy = +1 for every tap
    out = D.deform_conv_forward(**c)
    # tap i of output row y now reads row y + i: a plain conv whose zero padding sits entirely below
    # the image (rows >= H are outside for the reference's rule too)
    padded = F.pad(torch.from_numpy(c['input']), (1, 1, 0, 2))
    ref = F.conv2d(padded, torch.from_numpy(c['weight']), torch.from_numpy(c['bias']), 1, 0).numpy()
    assert np.abs(out - ref).max() < 1e-10


@pytest.mark.parametrize('groups,DG,dil', [(1, 1, 1), (2, 2, 1), (1, 4, 2), (2, 1, 1)])
def test_explicit_backward_matches_autograd_of_torch_form(groups, DG, dil):
    c = _case(4, groups=groups, DG=DG, dil=dil, pad=dil)
    out_np = D.deform_conv_forward(**c)
    t = {k: torch.from_numpy(c[k]).requires_grad_(True) for k in ('input', 'offset', 'weight', 'bias')}
    out_t = D.deform_conv_torch(t['input'], t['offset'], t['weight'], t['bias'], c['stride'], c['padding'],
                                c['dilation'], groups, DG)
    assert np.abs(out_t.detach().numpy() - out_np).max() < 1e-10
    go = np.random.default_rng(5).standard_normal(out_np.shape)
    out_t.backward(torch.from_numpy(go))
    gi, goff, gw, gb = D.deform_conv_backward(c['input'], c['offset'], c['weight'], go, c['stride'],
                                              c['padding'], c['dilation'], groups, DG)
    for name, got in (('input', gi), ('offset', goff), ('weight', gw), ('bias', gb)):
        assert np.abs(got - t[name].grad.numpy()).max() < 1e-9, name


def test_gradcheck_with_the_reference_tolerances():
    """reference test.py:377-400: float64, offsets ~ 2*randn, eps=1e-3, atol=1e-3, rtol=1e-2"""
    c = _case(6, B=2, C=4, H=4, W=4, Co=4, groups=2, DG=1)

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, offset, weight, bias):
            ctx.save_for_backward(input, offset, weight)
            return torch.from_numpy(D.deform_conv_forward(input.numpy(), offset.numpy(), weight.numpy(),
                                                          bias.numpy(), c['stride'], c['padding'],
                                                          c['dilation'], c['groups'], c['DG']))

        @staticmethod
        def backward(ctx, g):
            i, o, w = (v.numpy() for v in ctx.saved_tensors)
            r = D.deform_conv_backward(i, o, w, g.numpy(), c['stride'], c['padding'], c['dilation'],
                                       c['groups'], c['DG'])
            return tuple(torch.from_numpy(np.ascontiguousarray(v)) for v in r)
    args = [torch.from_numpy(c[k]).requires_grad_(True) for k in ('input', 'offset', 'weight', 'bias')]
    assert torch.autograd.gradcheck(Fn.apply, args, eps=1e-3, atol=1e-3, rtol=1e-2, raise_exception=True)


def test_module_surface_without_gpu():
    from deformable_conv import DeformConv, DeformConvPack, DeformConvFunction
    m = DeformConv(21, 21, 3, stride=1, padding=6, dilation=6, deformable_groups=21, bias=False)
    assert tuple(m.weight.shape) == (21, 21, 3, 3) and tuple(m.bias.shape) == (21,)
    assert m.bias.requires_grad is False and list(m.state_dict()) == ['weight', 'bias']
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 21, 8, 8), torch.zeros(1, 18, 8, 8))
    with pytest.raises(RuntimeError, match='no CPU path'):
        m(torch.zeros(1, 21, 8, 8), torch.zeros(1, 21 * 18, 8, 8))
    with pytest.raises(ValueError):
        DeformConv(5, 4, 3, 1, 1, groups=2)
    p = DeformConvPack(4, 4, 3, 1, 1)
    assert float(p.conv_offset.weight.abs().max()) == 0.0 and p.conv_offset.lr_mult == 0.1
    assert list(p.state_dict()) == ['weight', 'bias', 'conv_offset.weight', 'conv_offset.bias']


@pytest.mark.parametrize('DG,stride,pad,dil', [(1, 1, 1, 1), (2, 2, 1, 1), (4, 1, 2, 2)])
def test_vectorised_oracle_equals_the_reference_loops(DG, stride, pad, dil):
    """oracle/dcn_loops.py restates the reference's three device loops index by index (column layout,
    offset channel order, `> -1` / `<= -1` boundary rules, C-style int() truncation in col2im): the vectorised
    oracle must agree with it on columns, input gradient and offset gradient."""
    from oracle import dcn_loops as L
    rng = np.random.default_rng(13 + DG)
    B, C, H, W, kh, kw = 2, 4, 5, 4, 3, 3
    st, pd, dl = (stride, stride), (pad, pad), (dil, dil)
    Ho, Wo = D._out_size(H, W, kh, kw, st, pd, dl)
    im = rng.standard_normal((B, C, H, W))
    off = rng.standard_normal((B, DG * 2 * kh * kw, Ho, Wo)) * 2
    off[0, 0, 0, 0] = -50.0                              # far outside
    off[0, 1, 0, 0] = -1.0 - (0 * st[1] - pd[1])         # w_im exactly -1: excluded by the `> -1` rule
    cols = L.columns(im, off, kh, kw, st, pd, dl, DG)                      # [C*K][B][Ho][Wo]
    mine = D.deform_columns(im, off, kh, kw, st, pd, dl, DG)               # [B][C][K][Ho][Wo]
    assert np.abs(cols.reshape(C, kh * kw, B, Ho, Wo).transpose(2, 0, 1, 3, 4) - mine).max() < 1e-12
    # gradients with groups=1, identity-like use of the backward: feed a random column gradient through both
    Co = 3
    wgt = rng.standard_normal((Co, C, kh, kw))
    go = rng.standard_normal((B, Co, Ho, Wo))
    gi, goff, _, _ = D.deform_conv_backward(im, off, wgt, go, st, pd, dl, 1, DG)
    gcol = np.einsum('ok,bop->kbp', wgt.reshape(Co, -1), go.reshape(B, Co, -1)).reshape(C * kh * kw, B, Ho, Wo)
    assert np.abs(L.col2im(gcol, off, im.shape, kh, kw, st, pd, dl, DG) - gi).max() < 1e-10
    assert np.abs(L.col2im_coord(gcol, im, off, kh, kw, st, pd, dl, DG) - goff).max() < 1e-10


# ---- modulated form (DCNv2): the invariants the reference's test.py checks for it -----------------------------------
def _mcase(seed, **kw):
    c = _case(seed, **kw)
    rng = np.random.default_rng(seed + 100)
    B, _, Ho, Wo = c['offset'].shape
    k = c['weight'].shape[2]
    c['mask'] = rng.random((B, c['DG'] * k * k, Ho, Wo))
    return c


def _margs(c):
    return (c['input'], c['offset'], c['mask'], c['weight'], c['bias'], c['stride'], c['padding'], c['dilation'],
            c['groups'], c['DG'])


@pytest.mark.parametrize('groups,stride,pad,dil', [(2, 1, 1, 1), (1, 2, 1, 1), (1, 1, 3, 3)])
def test_modulated_zero_offset_unit_mask_is_plain_convolution(groups, stride, pad, dil):
    """reference test.py:69-110 (check_mdconv_zero_offset: offsets 0, mask 1 -> nn.Conv2d); a constant mask m scales the
    bias-free part by m"""
    c = _mcase(11, H=9, W=8, groups=groups, stride=stride, pad=pad, dil=dil)
    c['offset'][:] = 0
    c['mask'][:] = 1.0
    out = D.modulated_deform_conv_forward(*_margs(c))
    conv = lambda bias: F.conv2d(torch.from_numpy(c['input']), torch.from_numpy(c['weight']), bias, stride, pad, dil,
                                 groups).numpy()
    assert np.abs(out - conv(torch.from_numpy(c['bias']))).max() < 1e-10
    c['mask'][:] = 0.5
    out = D.modulated_deform_conv_forward(*_margs(c))
    assert np.abs(out - (0.5 * conv(None) + c['bias'].reshape(1, -1, 1, 1))).max() < 1e-10


def test_modulated_unit_mask_equals_v1_and_identity_kernel():
    """mask == 1: the v1 operator (forward and every gradient); identity kernel: reference test.py:142-181"""
    c = _mcase(12, groups=2, DG=2)
    c['mask'][:] = 1.0
    v1 = {k: v for k, v in c.items() if k != 'mask'}
    assert np.abs(D.modulated_deform_conv_forward(*_margs(c)) - D.deform_conv_forward(**v1)).max() < 1e-12
    go = np.random.default_rng(5).standard_normal(D.deform_conv_forward(**v1).shape)
    gi, goff, gm, gw, gb = D.modulated_deform_conv_backward(c['input'], c['offset'], c['mask'], c['weight'], go,
                                                            c['stride'], c['padding'], c['dilation'], c['groups'], c['DG'])
    ri, roff, rw, rb = D.deform_conv_backward(c['input'], c['offset'], c['weight'], go, c['stride'], c['padding'],
                                              c['dilation'], c['groups'], c['DG'])
    for a, b in ((gi, ri), (goff, roff), (gw, rw), (gb, rb)):
        assert np.abs(a - b).max() < 1e-12
    c['offset'][:] = 0
    c['weight'][:] = 0
    c['bias'][:] = 0
    Co = c['weight'].shape[0]
    for q in range(Co):
        c['weight'][q, q % (Co // 2), 1, 1] = 1.0
    assert np.abs(D.modulated_deform_conv_forward(*_margs(c)) - c['input']).max() < 1e-12


@pytest.mark.parametrize('groups,DG,dil', [(1, 1, 1), (2, 2, 1), (1, 4, 2)])
def test_modulated_explicit_backward_matches_autograd_and_finite_differences(groups, DG, dil):
    """the explicit col2im / coordinate / mask formulas against autograd of the torch form, and the torch form against
    finite differences with the reference's gradcheck settings (test.py:405-434: float64, eps 1e-3, atol 1e-3, rtol 1e-2)"""
    c = _mcase(13, groups=groups, DG=DG, dil=dil, pad=dil)
    t = {k: torch.from_numpy(c[k]).requires_grad_(True) for k in ('input', 'offset', 'mask', 'weight', 'bias')}
    out = D.deform_conv_torch(t['input'], t['offset'], t['weight'], t['bias'], c['stride'], c['padding'], c['dilation'],
                              c['groups'], c['DG'], mask=t['mask'])
    assert np.abs(out.detach().numpy() - D.modulated_deform_conv_forward(*_margs(c))).max() < 1e-10
    go = np.random.default_rng(6).standard_normal(tuple(out.shape))
    out.backward(torch.from_numpy(go))
    gi, goff, gm, gw, gb = D.modulated_deform_conv_backward(c['input'], c['offset'], c['mask'], c['weight'], go,
                                                            c['stride'], c['padding'], c['dilation'], c['groups'], c['DG'])
    for name, got in (('input', gi), ('offset', goff), ('mask', gm), ('weight', gw), ('bias', gb)):
        assert np.abs(got - t[name].grad.numpy()).max() < 1e-9, name


def test_modulated_gradcheck_with_the_reference_tolerances():
    """reference test.py:405-434: float64, eps=1e-3, atol=1e-3, rtol=1e-2 - through the EXPLICIT numpy forward / backward
    (the geometry and offsets of the v1 check above, whose sample positions stay clear of the bilinear kinks)"""
    c = _case(6, B=2, C=4, H=4, W=4, Co=4, groups=2, DG=1)
    c['mask'] = np.random.default_rng(7).random((2, 9, 4, 4)) + 0.25

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, offset, mask, weight, bias):
            ctx.save_for_backward(input, offset, mask, weight)
            return torch.from_numpy(D.modulated_deform_conv_forward(
                input.numpy(), offset.numpy(), mask.numpy(), weight.numpy(), bias.numpy(), c['stride'], c['padding'],
                c['dilation'], c['groups'], c['DG']))

        @staticmethod
        def backward(ctx, g):
            i, o, m, w = (v.numpy() for v in ctx.saved_tensors)
            r = D.modulated_deform_conv_backward(i, o, m, w, g.numpy(), c['stride'], c['padding'], c['dilation'],
                                                 c['groups'], c['DG'])
            return tuple(torch.from_numpy(np.ascontiguousarray(v)) for v in r)
    args = [torch.from_numpy(c[k]).requires_grad_(True) for k in ('input', 'offset', 'mask', 'weight', 'bias')]
    assert torch.autograd.gradcheck(Fn.apply, args, eps=1e-3, atol=1e-3, rtol=1e-2, raise_exception=True)
