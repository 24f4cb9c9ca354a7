"""TEST INFRASTRUCTURE ONLY - CPU restatement (torch, any float dtype) of the aggregation head of the reference's
pose_hrnet_PoseAggr (lib/models/pose_hrnet_PoseAggr.py:593-646): frame differences, the offset-feature BasicBlock
chain in eval mode (:441-485), five dilated offset convs (:497-506), five deformable convolutions (:508-516, via
oracle/dcn_cpu.py), the 0.2-weighted sum and the temporal mix (:632-640).

Parity unpinned: the reference's deformable convolution is CUDA-only and unbuildable here, so no
golden vectors of this model exist; the pieces are the torch ops the reference calls plus oracle/dcn_cpu.py, which
is pinned by the invariants of the reference's own DCN tests. Only tests/ import this module."""
import torch
import torch.nn.functional as F

from oracle import dcn_cpu

FRAME_WEIGHTS = (0.1, 0.25, 0.3, 0.25, 0.1)      # prev2, prev1, current, next1, next2 (:640)


def _bn(x, sd, p, training=False, eps=1e-5):
    if training:      # batch statistics (the running statistics are not needed by the comparison)
        return F.batch_norm(x, None, None, sd[p + '.weight'], sd[p + '.bias'], True, 0.0, eps)
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def offset_feats(x, sd, nblocks=20, training=False):
    """BasicBlock chain (pose_hrnet_PoseAggr.py:28-57 blocks)"""
    for k in range(nblocks):
        p = 'offset_feats.{}'.format(k)
        out = F.relu(_bn(F.conv2d(x, sd[p + '.conv1.weight'], None, padding=1), sd, p + '.bn1', training))
        out = _bn(F.conv2d(out, sd[p + '.conv2.weight'], None, padding=1), sd, p + '.bn2', training)
        res = x
        if (p + '.downsample.0.weight') in sd:
            res = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], None), sd, p + '.downsample.1', training)
        x = F.relu(out + res)
    return x


def aggregate(logits, sd, dilation_rates=(3, 6, 12, 18, 24), training=False):
    """logits (5B, nj, H, W) ordered [prev2 | prev1 | current | next1 | next2] -> (B, nj, H, W)"""
    T, nj, H, W = logits.shape
    B = T // 5
    ref = logits[2 * B:3 * B].repeat(5, 1, 1, 1)
    feats = offset_feats(ref - logits, sd, training=training)
    warped = 0
    for k, d in enumerate(dilation_rates, 1):
        off = F.conv2d(feats, sd['offsets{}.weight'.format(k)], None, padding=d, dilation=d)
        warped = warped + dcn_cpu.deform_conv_torch(logits, off, sd['deform_conv{}.weight'.format(k)],
                                                    sd['deform_conv{}.bias'.format(k)], (1, 1), (d, d), (d, d), 1, nj)
    warped = 0.2 * warped
    out = 0
    for g, wgt in enumerate(FRAME_WEIGHTS):
        out = out + wgt * warped[g * B:(g + 1) * B]
    return out


def heatmaps(agg, temp):
    flat = agg.reshape(agg.shape[0], agg.shape[1], -1)
    return F.softmax(flat * temp, dim=2).reshape(agg.shape)
