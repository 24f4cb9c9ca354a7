"""DeformConv / DeformConvPack modules (reference lib/deformable_conv/modules/deform_conv.py:14-99:
constructor arguments, parameter names and shapes, kaiming-uniform(a=sqrt(5)) weights, uniform bias
kept as a frozen (still added) Parameter when bias=False, zero-initialised offset predictor with lr_mult)."""
import math

import torch
from torch import nn
from torch.nn import init
from torch.nn.modules.utils import _pair

from ..functions.deform_conv_func import DeformConvFunction


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1,
                 deformable_groups=1, im2col_step=64, bias=True):
        super(DeformConv, self).__init__()
        if in_channels % groups != 0:
            raise ValueError('in_channels {} must be divisible by groups {}'.format(in_channels, groups))
        if out_channels % groups != 0:
            raise ValueError('out_channels {} must be divisible by groups {}'.format(out_channels, groups))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups, self.im2col_step = groups, deformable_groups, im2col_step
        self.use_bias = bias
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()
        if not self.use_bias:
            self.bias.requires_grad = False

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input, offset):
        expect = 2 * self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        assert expect == offset.shape[1], 'offset has {} channels, expected {}'.format(offset.shape[1], expect)
        return DeformConvFunction.apply(input, offset, self.weight, self.bias, self.stride, self.padding,
                                        self.dilation, self.groups, self.deformable_groups, self.im2col_step)


_DeformConv = DeformConvFunction.apply


class DeformConvPack(DeformConv):
    """DeformConv that predicts its own offsets with a plain convolution (zero-initialised, so the
    module starts out as an ordinary convolution)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1,
                 deformable_groups=1, im2col_step=64, bias=True, lr_mult=0.1):
        super(DeformConvPack, self).__init__(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                             groups, deformable_groups, im2col_step, bias)
        n_off = self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = nn.Conv2d(self.in_channels, n_off, kernel_size=self.kernel_size, stride=self.stride,
                                     padding=self.padding, bias=True)
        self.conv_offset.lr_mult = lr_mult
        self.init_offset()

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset.weight.zero_()
            self.conv_offset.bias.zero_()

    def forward(self, input):
        offset = self.conv_offset(input)
        return DeformConvFunction.apply(input, offset, self.weight, self.bias, self.stride, self.padding,
                                        self.dilation, self.groups, self.deformable_groups, self.im2col_step)
