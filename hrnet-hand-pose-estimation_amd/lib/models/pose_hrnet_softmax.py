"""pose_hrnet_softmax — the reference's trainable-softmax head variant on the HIP path
(reference lib/models/pose_hrnet_softmax.py).

Differences from pose_hrnet (reference file:line):
  * the three lower-resolution branches are up-sampled with align_corners=True (:499-501);
  * inter_feat is the 480-channel concatenation (:505), not stage 3's first branch;
  * the head output goes through a spatial softmax scaled by `trainable_temp` (:520-524), a scalar
    Parameter that is trainable iff cfg.MODEL.TRAINABLE_SOFTMAX (:355);
  * forward returns (heatmap_pred, inter_feat, trainable_temp) (:528).
state_dict(): the reference's keys in the reference's order ('trainable_temp' first, then the backbone).
"""
import torch
import torch.nn as nn

from hipnet import _capi as C
from models.pose_hrnet import (BN_MOMENTUM, BasicBlock, Bottleneck, HighResolutionModule,  # noqa: F401
                               PoseHighResolutionNet as _Base, blocks_dict)


class _SpatialSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, temp):
        x = x.contiguous()
        b, k, h, w = x.shape
        t = temp.detach().reshape(1).float().contiguous()
        out = torch.empty_like(x)
        C.call('hrnet_spatial_softmax_fwd', x.data_ptr(), t.data_ptr(), out.data_ptr(), b * k, h * w, C.stream_ptr())
        ctx.save_for_backward(x, out, t)
        ctx.temp_shape = temp.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        x, out, t = ctx.saved_tensors
        gout = gout.contiguous().float()
        b, k, h, w = x.shape
        dx = torch.empty_like(x)
        part = torch.empty(b * k, dtype=torch.float32, device=x.device)
        C.call('hrnet_spatial_softmax_bwd', x.data_ptr(), out.data_ptr(), gout.data_ptr(), t.data_ptr(),
               dx.data_ptr(), part.data_ptr(), b * k, h * w, C.stream_ptr())
        return dx, part.sum().reshape(ctx.temp_shape)


class PoseHighResolutionNet(_Base):
    head_align_corners = True      # read by hipnet.engine.Plan
    inter_from_cat = True

    def __init__(self, cfg, **kwargs):
        # the temperature is registered before the backbone so that state_dict() lists it first, as the
        # reference's does (own parameters precede sub-modules)
        nn.Module.__init__(self)
        self.trainable_temp = nn.Parameter(torch.tensor(1.0), requires_grad=bool(cfg.MODEL.TRAINABLE_SOFTMAX))
        self._build(cfg, **kwargs)

    def forward(self, x):
        hm, inter = super(PoseHighResolutionNet, self).forward(x)
        if self.training and torch.is_grad_enabled():
            heat = _SpatialSoftmax.apply(hm, self.trainable_temp)
        else:
            with torch.no_grad():
                heat = _SpatialSoftmax.apply(hm, self.trainable_temp)
        return heat, inter, self.trainable_temp


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
