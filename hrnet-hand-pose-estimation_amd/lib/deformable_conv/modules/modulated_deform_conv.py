"""ModulatedDeformConv / ModulatedDeformConvPack (DCNv2; reference lib/deformable_conv/modules/modulated_deform_conv.py:
14-104: constructor arguments, parameter names and shapes, initialisation as DeformConv; the Pack variant predicts
2K offsets + K mask logits per deformable group with one zero-initialised convolution `conv_offset_mask`, splits its
output in three chunks (offsets = cat(chunk 1, chunk 2), mask = sigmoid(chunk 3)) - so a fresh module is an ordinary
convolution scaled by 0.5)."""
import torch
from torch import nn

from ..functions.modulated_deform_conv_func import ModulatedDeformConvFunction
from .deform_conv import DeformConv


class ModulatedDeformConv(DeformConv):
    """same parameters as DeformConv (weight, always-present bias frozen when bias=False)"""

    def forward(self, input, offset, mask):
        k = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        assert 2 * k == offset.shape[1], 'offset has {} channels, expected {}'.format(offset.shape[1], 2 * k)
        assert k == mask.shape[1], 'mask has {} channels, expected {}'.format(mask.shape[1], k)
        return ModulatedDeformConvFunction.apply(input, offset, mask, self.weight, self.bias, self.stride,
                                                 self.padding, self.dilation, self.groups, self.deformable_groups,
                                                 self.im2col_step)


_ModulatedDeformConv = ModulatedDeformConvFunction.apply


class ModulatedDeformConvPack(ModulatedDeformConv):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1,
                 deformable_groups=1, im2col_step=64, bias=True, lr_mult=0.1):
        super(ModulatedDeformConvPack, self).__init__(in_channels, out_channels, kernel_size, stride, padding,
                                                      dilation, groups, deformable_groups, im2col_step, bias)
        n = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(self.in_channels, n, kernel_size=self.kernel_size, stride=self.stride,
                                          padding=self.padding, bias=True)
        self.conv_offset_mask.lr_mult = lr_mult
        self.init_offset()

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, input):
        out = self.conv_offset_mask(input)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return ModulatedDeformConvFunction.apply(input, offset, mask, self.weight, self.bias, self.stride,
                                                 self.padding, self.dilation, self.groups, self.deformable_groups,
                                                 self.im2col_step)
