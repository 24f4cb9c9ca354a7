"""Op-by-op autograd layers over the C ABI (one HIP launch per step, no recorded program): the building blocks of
the parts of the reference that sit outside the recorded HRNet programs - today the aggregation head of
pose_hrnet_PoseAggr in TRAINING mode (reference lib/models/pose_hrnet_PoseAggr.py:612-646 and its autograd).

Activations are NHWC tensors in the compute dtype (channel counts padded as the kernels need: multiples of 8 in,
16 out); parameters stay the module's f32 OIHW tensors and are packed per call (these paths are small and not
performance critical). Every forward / backward below is a hrnet_* launch; torch supplies memory, views and the
autograd graph only.
"""
import ctypes

import torch

from . import _capi as C


def _dt(t):
    return 0 if t.dtype == torch.float32 else 1


def _p(t):
    return None if t is None else t.data_ptr()


def _pack(w, dtype, mode, ks):
    """OIHW f32 -> packed kernel layout (mode 0 forward / 1 transposed input-gradient copy)"""
    co, ci = w.shape[0], w.shape[1]
    cop, cip = (co + 15) // 16 * 16, (ci + 7) // 8 * 8
    out = torch.empty(cop * ks * ks * cip, dtype=dtype, device=w.device)
    wf = w.detach().float().contiguous()
    C.call('hrnet_pack_weights', 0 if dtype == torch.float32 else 1, wf.data_ptr(), out.data_ptr(), co, ci, ks, cop, cip,
           mode, C.stream_ptr())
    return out, cop, cip


def _conv_backward(x, gy, weight, ks, need_gx):
    """input gradient (conv with the transposed packed kernel) and weight gradient (slabs + reduce) of y = conv(x)"""
    N, H, W, cin = x.shape
    cout = gy.shape[3]
    dt = _dt(x)
    gx = None
    if need_gx:
        wd, _, _ = _pack(weight, x.dtype, 1, ks)
        gx = torch.empty_like(x)
        C.call('hrnet_conv2d', dt, gy.data_ptr(), wd.data_ptr(), None, None, None, gx.data_ptr(), None, N, H, W, cout,
               H, W, cin, ks, 1, 0, 0, 0, C.stream_ptr())
    ns = C.call('hrnet_wgrad_splits', dt, N, H, W, cout, cin, ks, 1)
    slabs = torch.empty(ns * cout * ks * ks * cin, dtype=torch.float32, device=x.device)
    C.call('hrnet_conv2d_wgrad', dt, x.data_ptr(), gy.data_ptr(), None, None, slabs.data_ptr(), N, H, W, cin, H, W,
           cout, ks, 1, 0, ns, C.stream_ptr())
    gw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
    C.call('hrnet_wgrad_reduce', slabs.data_ptr(), gw.data_ptr(), ns, cout, cin, ks, weight.shape[0], weight.shape[1],
           0, 0, C.stream_ptr())
    return gx, gw


class ConvBN(torch.autograd.Function):
    """z = relu?(BatchNorm_train(conv(x))): conv with batch statistics, finalize (running statistics updated),
    apply; backward: BatchNorm backward (reduce, finalize, apply) then the conv's two gradients."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn_module, relu):
        N, H, W, cin = x.shape
        ks = weight.shape[2]
        dt = _dt(x)
        wf, cop, cip = _pack(weight, x.dtype, 0, ks)
        assert cip == cin and cop == gamma.numel(), (cip, cin, cop, gamma.numel())
        y = torch.empty((N, H, W, cop), dtype=x.dtype, device=x.device)
        tiles = C.call('hrnet_conv_tiles', N, H, W, cop, ks, 1)
        rows = torch.empty(tiles * 2 * cop, dtype=torch.float32, device=x.device)
        C.call('hrnet_conv2d', dt, x.data_ptr(), wf.data_ptr(), None, None, None, y.data_ptr(), rows.data_ptr(), N, H, W,
               cin, H, W, cop, ks, 1, 0, 0, 0, C.stream_ptr())
        scale, shift, mean, invstd = (torch.empty(cop, dtype=torch.float32, device=x.device) for _ in range(4))
        m = bn_module
        C.call('hrnet_bn_finalize', rows.data_ptr(), tiles, cop, float(N * H * W), gamma.data_ptr(), beta.data_ptr(),
               m.running_mean.data_ptr(), m.running_var.data_ptr(), m.num_batches_tracked.data_ptr(),
               m.momentum if m.momentum is not None else 0.1, m.eps, 1, scale.data_ptr(), shift.data_ptr(),
               mean.data_ptr(), invstd.data_ptr(), C.stream_ptr())
        z = torch.empty_like(y)
        C.call('hrnet_sum_terms', dt, z.data_ptr(), N, H, W, cop, 1, (ctypes.c_void_p * 1)(y.data_ptr()),
               (ctypes.c_void_p * 1)(scale.data_ptr()), (ctypes.c_void_p * 1)(shift.data_ptr()), (ctypes.c_int * 1)(0),
               (ctypes.c_int * 1)(1 if relu else 0), 0, C.stream_ptr())
        ctx.save_for_backward(x, weight, gamma, y, scale, shift, mean, invstd)
        ctx.relu = bool(relu)
        return z

    @staticmethod
    def backward(ctx, gz):
        x, weight, gamma, y, scale, shift, mean, invstd = ctx.saved_tensors
        N, H, W, cop = y.shape
        dt = _dt(x)
        gz = gz.contiguous()
        blocks = C.call('hrnet_reduce_blocks', N, H, W, cop)
        part = torch.empty(blocks * 2 * cop, dtype=torch.float32, device=x.device)
        C.call('hrnet_bn_bwd_reduce', dt, part.data_ptr(), gz.data_ptr(), None, y.data_ptr(), scale.data_ptr(),
               shift.data_ptr(), N, H, W, cop, 0, 1 if ctx.relu else 0, C.stream_ptr())
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        coef = torch.empty(3 * cop, dtype=torch.float32, device=x.device)
        C.call('hrnet_bn_bwd_finalize', part.data_ptr(), blocks, cop, float(N * H * W), gamma.data_ptr(), mean.data_ptr(),
               invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), 0, C.stream_ptr())
        gy = torch.empty_like(y)
        C.call('hrnet_grad_term', dt, gy.data_ptr(), gz.data_ptr(), None, y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
               coef.data_ptr(), N, H, W, cop, 0, 1 if ctx.relu else 0, 0, C.stream_ptr())
        gx, gw = _conv_backward(x, gy, weight, weight.shape[2], ctx.needs_input_grad[0])
        return gx, gw, dgamma, dbeta, None, None


class AddReLU(torch.autograd.Function):
    """out = relu(a + b) (the residual add that closes a BasicBlock, pose_hrnet.py:54-55)"""

    @staticmethod
    def forward(ctx, a, b):
        N, H, W, c = a.shape
        out = torch.empty_like(a)
        C.call('hrnet_sum_terms', _dt(a), out.data_ptr(), N, H, W, c, 2, (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr()),
               (ctypes.c_void_p * 2)(None, None), (ctypes.c_void_p * 2)(None, None), (ctypes.c_int * 2)(0, 0),
               (ctypes.c_int * 2)(0, 0), 1, C.stream_ptr())
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        N, H, W, c = out.shape
        g = g.contiguous()
        d = torch.empty_like(out)
        C.call('hrnet_grad_term', _dt(out), d.data_ptr(), g.data_ptr(), out.data_ptr(), None, None, None, None, N, H, W, c,
               0, 0, 0, C.stream_ptr())
        return d, d


class ToNHWC(torch.autograd.Function):
    """NCHW f32 -> NHWC compute dtype, channels zero-padded to cpad"""

    @staticmethod
    def forward(ctx, x, cpad, dtype):
        N, c, H, W = x.shape
        out = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
        C.call('hrnet_nchw_to_nhwc', 0 if dtype == torch.float32 else 1, x.contiguous().data_ptr(), out.data_ptr(), N, H, W,
               cpad, c, C.stream_ptr())
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, cpad = g.shape
        out = torch.empty((N, ctx.c, H, W), dtype=torch.float32, device=g.device)
        C.call('hrnet_nhwc_to_nchw', _dt(g), g.contiguous().data_ptr(), out.data_ptr(), N, H, W, cpad, ctx.c, C.stream_ptr())
        return out, None, None


class ToNCHW(torch.autograd.Function):
    """NHWC compute dtype -> NCHW f32, first c channels"""

    @staticmethod
    def forward(ctx, x, c):
        N, H, W, cpad = x.shape
        out = torch.empty((N, c, H, W), dtype=torch.float32, device=x.device)
        C.call('hrnet_nhwc_to_nchw', _dt(x), x.contiguous().data_ptr(), out.data_ptr(), N, H, W, cpad, c, C.stream_ptr())
        ctx.cpad, ctx.dtype = cpad, x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        N, c, H, W = g.shape
        out = torch.empty((N, H, W, ctx.cpad), dtype=ctx.dtype, device=g.device)
        C.call('hrnet_nchw_to_nhwc', 0 if ctx.dtype == torch.float32 else 1, g.contiguous().float().data_ptr(),
               out.data_ptr(), N, H, W, ctx.cpad, c, C.stream_ptr())
        return out, None


def _tap_matrices(weight, dtype, transposed):
    """the nine 1x1 matrices of a 3x3 kernel, packed; transposed: [Cin][Cout] of the mirrored tap (input gradient)"""
    w = weight.detach().float()
    mats = []
    for t in range(9):
        src = 8 - t if transposed else t
        m = w[:, :, src // 3, src % 3]
        m = (m.t() if transposed else m).contiguous().reshape(m.shape[1] if transposed else m.shape[0], -1, 1, 1)
        mats.append(_pack(m, dtype, 0, 1))
    cop, cip = mats[0][1], mats[0][2]
    return torch.stack([m[0] for m in mats]), cop, cip


class DilatedConv(torch.autograd.Function):
    """y = conv3x3(x, dilation d, padding d), no bias (the offset convs, pose_hrnet_PoseAggr.py:497-506).
    backward: the input gradient is the same op with the transposed, mirrored taps; the weight gradient of tap t
    is a 1x1 weight gradient against the input displaced by that tap (a displaced copy made by a 1x1 identity conv)."""

    @staticmethod
    def forward(ctx, x, weight, dilation):
        N, H, W, cin = x.shape
        taps, cop, cip = _tap_matrices(weight, x.dtype, False)
        assert cip == cin
        y = torch.empty((N, H, W, cop), dtype=x.dtype, device=x.device)
        C.call('hrnet_conv2d_dilated3x3', _dt(x), x.data_ptr(), taps.data_ptr(), taps.stride(0) * taps.element_size(),
               y.data_ptr(), N, H, W, cin, cop, dilation, C.stream_ptr())
        ctx.save_for_backward(x, weight)
        ctx.d = dilation
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        N, H, W, cin = x.shape
        cop = gy.shape[3]
        dt, d = _dt(x), ctx.d
        gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            # (padded output channels of y carry zero weights: pad the transposed matrices' K to cop)
            w = weight.detach().float()
            wpad = torch.zeros((cop, cin, 3, 3), dtype=torch.float32, device=x.device)
            wpad[:w.shape[0], :w.shape[1]] = w
            taps_t, _, _ = _tap_matrices(wpad, x.dtype, True)
            gx = torch.empty_like(x)
            C.call('hrnet_conv2d_dilated3x3', dt, gy.data_ptr(), taps_t.data_ptr(), taps_t.stride(0) * taps_t.element_size(),
                   gx.data_ptr(), N, H, W, cop, cin, d, C.stream_ptr())
        # weight gradient, tap by tap
        eye = torch.eye(cin, dtype=torch.float32, device=x.device).reshape(cin, cin, 1, 1)
        ident, _, _ = _pack(eye, x.dtype, 0, 1)
        gw = torch.zeros(weight.shape, dtype=torch.float32, device=x.device)
        co, ci = weight.shape[0], weight.shape[1]
        ns = C.call('hrnet_wgrad_splits', dt, N, H, W, cop, cin, 1, 1)
        slabs = torch.empty(ns * cop * cin, dtype=torch.float32, device=x.device)
        xs = torch.empty_like(x)
        g1 = torch.empty((co, ci, 1, 1), dtype=torch.float32, device=x.device)
        for t in range(9):
            # xs[p] = x[p + shift_t] (zero outside): one 1x1 identity conv over the displaced window
            op = C.HrOp()
            op.kind = C.OP_CONV
            for k, v in enumerate((dt, N, H, W, cin, H, W, cin, 1, 1, 0, 0, 0)):
                op.i[k] = v
            op.i[15], op.i[16] = (t // 3 - 1) * d, (t % 3 - 1) * d
            op.p[0], op.p[1], op.p[5] = x.data_ptr(), ident.data_ptr(), xs.data_ptr()
            C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
            C.call('hrnet_conv2d_wgrad', dt, xs.data_ptr(), gy.data_ptr(), None, None, slabs.data_ptr(), N, H, W, cin, H, W,
                   cop, 1, 1, 0, ns, C.stream_ptr())
            C.call('hrnet_wgrad_reduce', slabs.data_ptr(), g1.data_ptr(), ns, cop, cin, 1, co, ci, 0, 0, C.stream_ptr())
            gw[:, :, t // 3, t % 3] = g1[:, :, 0, 0]
        return gx, gw, None


class LinComb(torch.autograd.Function):
    """out = sum_j coef_j * src_j over f32 tensors of one shape"""

    @staticmethod
    def forward(ctx, coefs, *srcs):
        srcs = [s.contiguous() for s in srcs]
        out = torch.empty_like(srcs[0])
        C.call('hrnet_lincomb_f32', out.data_ptr(), out.numel(), len(srcs),
               (ctypes.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs]), (ctypes.c_float * len(srcs))(*coefs),
               C.stream_ptr())
        ctx.coefs = list(coefs)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs = []
        for j, c in enumerate(ctx.coefs):
            if ctx.needs_input_grad[1 + j]:
                o = torch.empty_like(g)
                C.call('hrnet_lincomb_f32', o.data_ptr(), o.numel(), 1, (ctypes.c_void_p * 1)(g.data_ptr()),
                       (ctypes.c_float * 1)(c), C.stream_ptr())
                outs.append(o)
            else:
                outs.append(None)
        return (None,) + tuple(outs)


def basic_block(x, unit):
    """BasicBlock forward in training mode (pose_hrnet.py:41-57) over the container module `unit`"""
    z1 = ConvBN.apply(x, unit.conv1.weight, unit.bn1.weight, unit.bn1.bias, unit.bn1, True)
    z2 = ConvBN.apply(z1, unit.conv2.weight, unit.bn2.weight, unit.bn2.bias, unit.bn2, False)
    res = x
    if hasattr(unit, 'downsample'):
        ds = unit.downsample
        res = ConvBN.apply(x, ds[0].weight, ds[1].weight, ds[1].bias, ds[1], False)
    return AddReLU.apply(z2, res)
