"""ModulatedDeformConvFunction (DCNv2): same call signature and gradient tuple as the reference
(lib/deformable_conv/functions/modulated_deform_conv_func.py:15-57), bound to hrnet_modulated_deform_conv_forward /
_backward of libhrnet_hip.so instead of the DCN CUDA extension: deformable convolution whose every sample is multiplied
by mask[b, dg * kh * kw + k, y, x]. im2col_step is validated like the reference and changes nothing (no column buffer).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from hipnet import _capi as C

from .deform_conv_func import _check, _geometry


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups,
                im2col_step):
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = int(groups), int(deformable_groups), int(im2col_step)
        kh, kw, Ho, Wo = _geometry(input, weight, ctx.stride, ctx.padding, ctx.dilation)
        _check(input, offset, weight, ctx.groups, ctx.deformable_groups, im2col_step, Ho, Wo)
        B = int(input.shape[0])
        if not mask.is_cuda or mask.dtype != torch.float32:
            raise TypeError('modulated_deform_conv: mask must be a float32 GPU tensor')
        if tuple(mask.shape) != (B, ctx.deformable_groups * kh * kw, Ho, Wo):
            raise ValueError('modulated_deform_conv: mask shape {} != {}'.format(
                tuple(mask.shape), (B, ctx.deformable_groups * kh * kw, Ho, Wo)))
        input, offset, mask, weight = input.contiguous(), offset.contiguous(), mask.contiguous(), weight.contiguous()
        bias_c = bias.contiguous() if bias is not None else None
        _, Cin, H, W = (int(v) for v in input.shape)
        Co = int(weight.shape[0])
        out = torch.empty(B, Co, Ho, Wo, device=input.device, dtype=torch.float32)
        C.call('hrnet_modulated_deform_conv_forward', C.ptr(input), C.ptr(offset), C.ptr(mask), C.ptr(weight),
               C.ptr(bias_c), C.ptr(out), B, Cin, H, W, Co, kh, kw, ctx.stride[0], ctx.stride[1], ctx.padding[0],
               ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups, ctx.deformable_groups, C.stream_ptr())
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        grad_output = grad_output.contiguous().float()
        B, Cin, H, W = (int(v) for v in input.shape)
        Co, Cg, kh, kw = (int(v) for v in weight.shape)
        Ho, Wo = int(grad_output.shape[2]), int(grad_output.shape[3])
        grad_input, grad_offset, grad_mask = torch.empty_like(input), torch.empty_like(offset), torch.empty_like(mask)
        grad_weight = torch.empty_like(weight)
        grad_bias = torch.empty(Co, device=input.device, dtype=torch.float32) if bias is not None else None
        blocks = C.call('hrnet_deform_conv_wgrad_blocks', B, Ho, Wo)
        scratch = torch.empty(blocks * (Co // ctx.groups) * Cg * kh * kw, device=input.device, dtype=torch.float32)
        C.call('hrnet_modulated_deform_conv_backward', C.ptr(input), C.ptr(offset), C.ptr(mask), C.ptr(weight),
               C.ptr(grad_output), C.ptr(grad_input), C.ptr(grad_offset), C.ptr(grad_mask), C.ptr(grad_weight),
               C.ptr(grad_bias), C.ptr(scratch), B, Cin, H, W, Co, kh, kw, ctx.stride[0], ctx.stride[1],
               ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups, ctx.deformable_groups,
               C.stream_ptr())
        return grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None, None, None
