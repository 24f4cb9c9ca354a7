"""Deformable convolution v1 on the HIP library (mirror of the reference's lib/deformable_conv
package surface for the one operator the pose networks use)."""
from .functions.deform_conv_func import DeformConvFunction
from .modules.deform_conv import DeformConv, DeformConvPack, _DeformConv

__all__ = ['DeformConvFunction', 'DeformConv', 'DeformConvPack', '_DeformConv']
