from .deform_conv import DeformConv, DeformConvPack, _DeformConv
from .modulated_deform_conv import ModulatedDeformConv, ModulatedDeformConvPack, _ModulatedDeformConv
