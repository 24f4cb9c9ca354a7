from .deform_conv import DeformConv, DeformConvPack, _DeformConv
