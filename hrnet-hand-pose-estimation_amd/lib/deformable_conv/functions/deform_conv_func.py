"""DeformConvFunction: same call signature and gradient tuple as the reference
(lib/deformable_conv/functions/deform_conv_func.py:15-66), bound to hrnet_deform_conv_forward /
hrnet_deform_conv_backward of libhrnet_hip.so instead of the DCN CUDA extension.

im2col_step is accepted and validated like the reference (the batch must be divisible by
min(batch, im2col_step), src/cuda/deform_conv_cuda.cu:52-55) but the kernels keep no column buffer,
so it does not change the result - the invariant the reference's own test.py:218 checks.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from hipnet import _capi as C


def _geometry(input, weight, stride, padding, dilation):
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    H, W = int(input.shape[2]), int(input.shape[3])
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    return kh, kw, Ho, Wo


def _check(input, offset, weight, group, deformable_groups, im2col_step, Ho, Wo):
    for name, t in (('input', input), ('offset', offset), ('weight', weight)):
        if not t.is_cuda:
            raise RuntimeError('deform_conv: {} is not on the GPU - there is no CPU path'.format(name))
        if t.dtype != torch.float32:
            raise TypeError('deform_conv: {} must be float32, got {}'.format(name, t.dtype))
    B, Cin = int(input.shape[0]), int(input.shape[1])
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    if int(weight.shape[1]) * group != Cin:
        raise ValueError('deform_conv: weight expects {} input channels, got {}'.format(
            int(weight.shape[1]) * group, Cin))
    if tuple(offset.shape) != (B, deformable_groups * 2 * kh * kw, Ho, Wo):
        raise ValueError('deform_conv: offset shape {} != {}'.format(
            tuple(offset.shape), (B, deformable_groups * 2 * kh * kw, Ho, Wo)))
    step = min(B, int(im2col_step))
    if step <= 0 or B % step != 0:
        raise ValueError('deform_conv: batch {} must divide im2col_step {}'.format(B, step))


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, bias, stride, padding, dilation, group, deformable_groups,
                im2col_step):
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.group, ctx.deformable_groups, ctx.im2col_step = int(group), int(deformable_groups), int(im2col_step)
        kh, kw, Ho, Wo = _geometry(input, weight, ctx.stride, ctx.padding, ctx.dilation)
        _check(input, offset, weight, ctx.group, ctx.deformable_groups, im2col_step, Ho, Wo)
        input, offset, weight = input.contiguous(), offset.contiguous(), weight.contiguous()
        bias_c = bias.contiguous() if bias is not None else None
        B, Cin, H, W = (int(v) for v in input.shape)
        Co = int(weight.shape[0])
        out = torch.empty(B, Co, Ho, Wo, device=input.device, dtype=torch.float32)
        C.call('hrnet_deform_conv_forward', C.ptr(input), C.ptr(offset), C.ptr(weight), C.ptr(bias_c),
               C.ptr(out), B, Cin, H, W, Co, kh, kw, ctx.stride[0], ctx.stride[1], ctx.padding[0],
               ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.group, ctx.deformable_groups,
               C.stream_ptr())
        ctx.save_for_backward(input, offset, weight, bias)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight, bias = ctx.saved_tensors
        grad_output = grad_output.contiguous().float()
        B, Cin, H, W = (int(v) for v in input.shape)
        Co, Cg, kh, kw = (int(v) for v in weight.shape)
        Ho, Wo = int(grad_output.shape[2]), int(grad_output.shape[3])
        grad_input = torch.empty_like(input)
        grad_offset = torch.empty_like(offset)
        grad_weight = torch.empty_like(weight)
        grad_bias = torch.empty(Co, device=input.device, dtype=torch.float32) if bias is not None else None
        blocks = C.call('hrnet_deform_conv_wgrad_blocks', B, Ho, Wo)
        scratch = torch.empty(blocks * (Co // ctx.group) * Cg * kh * kw, device=input.device,
                              dtype=torch.float32)
        C.call('hrnet_deform_conv_backward', C.ptr(input), C.ptr(offset), C.ptr(weight), C.ptr(grad_output),
               C.ptr(grad_input), C.ptr(grad_offset), C.ptr(grad_weight), C.ptr(grad_bias), C.ptr(scratch),
               B, Cin, H, W, Co, kh, kw, ctx.stride[0], ctx.stride[1], ctx.padding[0], ctx.padding[1],
               ctx.dilation[0], ctx.dilation[1], ctx.group, ctx.deformable_groups, C.stream_ptr())
        return grad_input, grad_offset, grad_weight, grad_bias, None, None, None, None, None, None
