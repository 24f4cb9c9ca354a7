"""Deformable convolution v1 and its modulated form (v2) on the HIP library (mirror of the reference's
lib/deformable_conv package surface: functions/ and modules/ of the two convolution operators; the PS-ROI pooling
operator belongs to detection models outside the pose path)."""
from .functions.deform_conv_func import DeformConvFunction
from .functions.modulated_deform_conv_func import ModulatedDeformConvFunction
from .modules.deform_conv import DeformConv, DeformConvPack, _DeformConv
from .modules.modulated_deform_conv import ModulatedDeformConv, ModulatedDeformConvPack, _ModulatedDeformConv

__all__ = ['DeformConvFunction', 'DeformConv', 'DeformConvPack', '_DeformConv',
           'ModulatedDeformConvFunction', 'ModulatedDeformConv', 'ModulatedDeformConvPack', '_ModulatedDeformConv']
