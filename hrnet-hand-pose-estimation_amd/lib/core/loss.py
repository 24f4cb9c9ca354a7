"""Loss modules of the hot path (reference lib/core/loss.py:15-50) on HIP kernels.

HeatmapLoss   mean over (B,K) of the per-map sum of (pred-gt)^2 (mode 'l2') or |pred-gt| ('l1')
JointsMSELoss visibility-weighted mean L2 norm of key-point errors (despite its name)

Both are autograd Functions over the C ABI (hrnet_heatmap_loss_*, hrnet_joints_loss_*); inputs
must be HIP tensors - there is no CPU path.
"""
import torch
import torch.nn as nn

from hipnet import _capi as C


def _dev_f32(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('{}: expected a HIP-device tensor (no CPU path in this build)'.format(what))
    return t.contiguous().float()


class _HeatmapLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mode):
        bk = pred.shape[0] * pred.shape[1] if pred.dim() == 4 else pred.shape[0]
        hw = pred.shape[-1] * pred.shape[-2]
        partial = torch.empty(bk, dtype=torch.float32, device=pred.device)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        C.call('hrnet_heatmap_loss_fwd', pred.data_ptr(), gt.data_ptr(), partial.data_ptr(), loss.data_ptr(), bk, hw,
               mode, C.stream_ptr())
        ctx.save_for_backward(pred, gt)
        ctx.mode, ctx.bk, ctx.hw = mode, bk, hw
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        pred, gt = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        g = gout.contiguous().float().reshape(1)
        C.call('hrnet_heatmap_loss_bwd', pred.data_ptr(), gt.data_ptr(), g.data_ptr(), dpred.data_ptr(), ctx.bk, ctx.hw,
               ctx.mode, C.stream_ptr())
        return dpred, None, None


class HeatmapLoss(nn.Module):
    def __init__(self, mode='l2'):
        super().__init__()
        if mode not in ('l2', 'l1'):
            raise ValueError("HeatmapLoss mode must be 'l2' or 'l1'")
        self.mode = mode

    def forward(self, pred, gt):
        assert pred.size() == gt.size(), \
            'Heatmap loss error: prediced heatmaps have size {}, but the groundtruth has {}'.format(pred.shape, gt.shape)
        pred = _dev_f32(pred, 'HeatmapLoss')
        gt = _dev_f32(gt, 'HeatmapLoss').detach()
        return _HeatmapLossFn.apply(pred, gt, 0 if self.mode == 'l2' else 1)


class _JointsLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, vis):
        b, k = pred.shape[0], pred.shape[1]
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        C.call('hrnet_joints_loss_fwd', pred.data_ptr(), gt.data_ptr(), C.ptr(vis), loss.data_ptr(), b, k,
               C.stream_ptr())
        ctx.save_for_backward(pred, gt, vis)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        pred, gt, vis = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        g = gout.contiguous().float().reshape(1)
        C.call('hrnet_joints_loss_bwd', pred.data_ptr(), gt.data_ptr(), C.ptr(vis), g.data_ptr(), dpred.data_ptr(),
               pred.shape[0], pred.shape[1], C.stream_ptr())
        return dpred, None, None


class JointsMSELoss(nn.Module):
    """pose2D_pred, pose2D_gt: B x K x 2; visibility: B x K (optional)."""

    def forward(self, pose2D_pred, pose2D_gt, visibility=None):
        pred = _dev_f32(pose2D_pred, 'JointsMSELoss')
        gt = _dev_f32(pose2D_gt, 'JointsMSELoss').detach()
        if pred.dim() != 3 or pred.shape[2] != 2:
            raise ValueError('JointsMSELoss expects B x K x 2 key points')
        vis = None
        if visibility is not None:
            vis = _dev_f32(visibility.to(pred.device), 'JointsMSELoss').reshape(pred.shape[0], pred.shape[1]).detach()
        return _JointsLossFn.apply(pred, gt, vis)
