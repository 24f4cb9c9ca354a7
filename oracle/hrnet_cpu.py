"""ORACLE — test infrastructure only (never imported by the product path).

A CPU restatement, in plain torch functional ops (fp32 or fp64), of the
reference's HRNet hot path. Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file.

Parity pin: `tests/test_oracle_golden.py` checks this restatement against
fixtures produced by running the reference's own module
(/root/reference/lib/models/pose_hrnet.py, lib/core/loss.py) in the build
container (generator: tests/golden/make_golden.py).  The expectation decode
(`use_softmax=True`) calls the un-vendored, un-pinned `kornia` in the reference
(lib/utils/heatmap_decoding.py:100) and is restated from its published
definition: PARITY UNPINNED at that one boundary.

Each function cites the reference lines it follows.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default (pose_hrnet.py:34 passes only momentum)
BN_MOMENTUM = 0.1      # pose_hrnet.py:18; fuse/transition BNs use the default, also 0.1


class Params:
    """state_dict wrapper: `p('layer1.0.conv1.weight')`; records BN buffer updates."""

    def __init__(self, state, training):
        self.s = state
        self.training = training
        self.new_stats = {}

    def __call__(self, key):
        return self.s[key]

    def has(self, key):
        return key in self.s


def _bn(P, x, prefix):
    """nn.BatchNorm2d forward (train: batch stats + running update; eval: running stats)."""
    w, b = P(prefix + '.weight'), P(prefix + '.bias')
    rm, rv = P(prefix + '.running_mean'), P(prefix + '.running_var')
    if P.training:
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        P.new_stats[prefix + '.running_mean'] = rm2
        P.new_stats[prefix + '.running_var'] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def _conv(P, x, prefix, stride=1):
    w = P(prefix + '.weight')
    b = P(prefix + '.bias') if P.has(prefix + '.bias') else None
    return F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)


def basic_block(P, x, pre):
    """BasicBlock.forward, pose_hrnet.py:41-57."""
    out = F.relu(_bn(P, _conv(P, x, pre + '.conv1'), pre + '.bn1'))
    out = _bn(P, _conv(P, out, pre + '.conv2'), pre + '.bn2')
    return F.relu(out + x)


def bottleneck(P, x, pre):
    """Bottleneck.forward, pose_hrnet.py:78-98 (1x1 -> 3x3 -> 1x1, expansion 4)."""
    out = F.relu(_bn(P, _conv(P, x, pre + '.conv1'), pre + '.bn1'))
    out = F.relu(_bn(P, _conv(P, out, pre + '.conv2'), pre + '.bn2'))
    out = _bn(P, _conv(P, out, pre + '.conv3'), pre + '.bn3')
    res = x
    if P.has(pre + '.downsample.0.weight'):
        res = _bn(P, _conv(P, x, pre + '.downsample.0'), pre + '.downsample.1')
    return F.relu(out + res)


def hr_module(P, xs, pre, num_blocks):
    """HighResolutionModule.forward, pose_hrnet.py:247-266 (+ fuse layers :187-242)."""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for k in range(num_blocks[i]):
            xs[i] = basic_block(P, xs[i], '{}.branches.{}.{}'.format(pre, i, k))
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                f = '{}.fuse_layers.{}.{}'.format(pre, i, j)
                t = _bn(P, _conv(P, xs[j], f + '.0'), f + '.1')
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:
                t = xs[j]
                for k in range(i - j):
                    f = '{}.fuse_layers.{}.{}.{}'.format(pre, i, j, k)
                    t = _bn(P, _conv(P, t, f + '.0', stride=2), f + '.1')
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def _transition(P, ys, pre, n_pre, n_cur):
    """_make_transition_layer semantics, pose_hrnet.py:419-458 / forward :521-546."""
    xs = []
    for i in range(n_cur):
        if i < n_pre:
            if P.has('{}.{}.0.weight'.format(pre, i)):
                t = F.relu(_bn(P, _conv(P, ys[i], '{}.{}.0'.format(pre, i)), '{}.{}.1'.format(pre, i)))
            else:
                t = ys[i]
        else:
            t = ys[-1]
            for j in range(i + 1 - n_pre):
                q = '{}.{}.{}'.format(pre, i, j)
                t = F.relu(_bn(P, _conv(P, t, q + '.0', stride=2), q + '.1'))
        xs.append(t)
    return xs


def hrnet_forward(state, extra, x, training=False, softmax_head=False):
    """PoseHighResolutionNet.forward, pose_hrnet.py:511-568.

    state: {key: tensor}; extra: cfg.MODEL.EXTRA-like mapping with STAGE2..4.
    Returns (heatmaps, inter_feat, new_running_stats).
    softmax_head=True: the pose_hrnet_softmax variant (pose_hrnet_softmax.py:497-528): align_corners=True
    up-sampling, inter_feat = the concatenation, spatial softmax times state['trainable_temp'].
    """
    P = Params(state, training)
    x = F.relu(_bn(P, _conv(P, x, 'conv1', 2), 'bn1'))
    x = F.relu(_bn(P, _conv(P, x, 'conv2', 2), 'bn2'))
    for k in range(4):
        x = bottleneck(P, x, 'layer1.{}'.format(k))
    ys = [x]
    inter_feat = None
    for s in (2, 3, 4):
        sc = extra['STAGE{}'.format(s)]
        nbr = sc['NUM_BRANCHES']
        xs = _transition(P, ys, 'transition{}'.format(s - 1), len(ys), nbr)
        for m in range(sc['NUM_MODULES']):
            xs = hr_module(P, xs, 'stage{}.{}'.format(s, m), sc['NUM_BLOCKS'])
        ys = xs
        if s == 3:
            inter_feat = ys[0]
    h, w = ys[0].shape[2], ys[0].shape[3]
    ups = [ys[0]] + [F.interpolate(t, size=(h, w), mode='bilinear', align_corners=softmax_head) for t in ys[1:]]
    z = torch.cat(ups, 1)
    if softmax_head:
        inter_feat = z
    z = F.relu(_bn(P, _conv(P, z, 'last_layer.0'), 'last_layer.1'))
    z = _conv(P, z, 'last_layer.3')
    if softmax_head:
        flat = z.reshape(z.shape[0], z.shape[1], -1)
        z = F.softmax(flat * state['trainable_temp'], dim=2).reshape(z.shape)
    return z, inter_feat, P.new_stats


def heatmap_loss(pred, gt, mode='l2'):
    """HeatmapLoss.forward, lib/core/loss.py:19-28."""
    assert pred.size() == gt.size()
    d = (pred - gt) ** 2 if mode == 'l2' else (pred - gt).abs()
    return d.sum(-1).sum(-1).mean()


def joints_mse_loss(pred, gt, visibility=None):
    """JointsMSELoss.forward, lib/core/loss.py:37-50 (vis-weighted L2 norm of keypoints)."""
    n = torch.norm(pred - gt, dim=2)
    if visibility is not None:
        vis = visibility.to(n.dtype)
        return (n * vis).sum() / torch.clamp(vis.sum(), min=1.0)
    return n.sum() / pred.shape[1]


def get_final_preds(hms, use_softmax=True):
    """lib/utils/heatmap_decoding.py:87-107.

    use_softmax=True: kornia spatial_expectation2d(normalized_coordinates=False)
    restated: (sum x*h, sum y*h) with x in [0,W-1], y in [0,H-1]; no softmax inside.
    False: argmax over the flattened map; u = idx % H, v = idx // H (H, as the reference).
    """
    assert isinstance(hms, torch.Tensor) and hms.ndim == 4
    b, k, h, w = hms.shape
    if use_softmax:
        xs = torch.arange(w, dtype=hms.dtype)
        ys = torch.arange(h, dtype=hms.dtype)
        ex = (hms.sum(2) * xs).sum(-1)
        ey = (hms.sum(3) * ys).sum(-1)
        return torch.stack((ex, ey), dim=2)
    idx = torch.argmax(hms.reshape(b, k, -1), dim=2)
    return torch.stack((idx % h, idx // h), dim=2).float()


def get_max_preds(batch_heatmaps):
    """lib/core/inference.py:18-46 (numpy argmax, zeroed where maxval <= 0)."""
    assert isinstance(batch_heatmaps, np.ndarray) and batch_heatmaps.ndim == 4
    b, k, _, w = batch_heatmaps.shape
    flat = batch_heatmaps.reshape(b, k, -1)
    idx = np.argmax(flat, 2).reshape(b, k, 1)
    maxvals = np.amax(flat, 2).reshape(b, k, 1)
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % w
    preds[:, :, 1] = np.floor(preds[:, :, 1] / w)
    preds *= np.tile(maxvals > 0.0, (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def total_loss(cfg_loss, hm_pred, hm_gt, pose2d_pred, pose2d_gt, visibility):
    """AverageMeter.computeLosses generic branch, lib/core/function.py:1334-1344."""
    tot = 0
    if cfg_loss['WITH_HEATMAP_LOSS']:
        tot = tot + cfg_loss['HEATMAP_LOSS_FACTOR'] * heatmap_loss(hm_pred, hm_gt)
    if cfg_loss['WITH_POSE2D_LOSS']:
        tot = tot + cfg_loss['POSE2D_LOSS_FACTOR'] * joints_mse_loss(
            pose2d_pred[:, :, 0:2], pose2d_gt[:, :, 0:2], visibility)
    return tot


W32_EXTRA = {
    'FINAL_CONV_KERNEL': 1,
    'STAGE2': dict(NUM_MODULES=1, NUM_BRANCHES=2, BLOCK='BASIC', NUM_BLOCKS=[4, 4], NUM_CHANNELS=[32, 64], FUSE_METHOD='SUM'),
    'STAGE3': dict(NUM_MODULES=4, NUM_BRANCHES=3, BLOCK='BASIC', NUM_BLOCKS=[4, 4, 4], NUM_CHANNELS=[32, 64, 128], FUSE_METHOD='SUM'),
    'STAGE4': dict(NUM_MODULES=3, NUM_BRANCHES=4, BLOCK='BASIC', NUM_BLOCKS=[4, 4, 4, 4], NUM_CHANNELS=[32, 64, 128, 256], FUSE_METHOD='SUM'),
}


def state_template(extra=W32_EXTRA, num_joints=21):
    """Shapes of every state_dict entry of PoseHighResolutionNet (SURVEY 8b key families)."""
    t = {}

    def conv(k, co, ci, ks, bias=False):
        t[k + '.weight'] = (co, ci, ks, ks)
        if bias:
            t[k + '.bias'] = (co,)

    def bn(k, c):
        for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
            t['{}.{}'.format(k, leaf)] = (c,)
        t[k + '.num_batches_tracked'] = ()

    conv('conv1', 64, 3, 3); bn('bn1', 64)
    conv('conv2', 64, 64, 3); bn('bn2', 64)
    inp = 64
    for k in range(4):
        p = 'layer1.{}'.format(k)
        conv(p + '.conv1', 64, inp, 1); bn(p + '.bn1', 64)
        conv(p + '.conv2', 64, 64, 3); bn(p + '.bn2', 64)
        conv(p + '.conv3', 256, 64, 1); bn(p + '.bn3', 256)
        if k == 0:
            conv(p + '.downsample.0', 256, inp, 1); bn(p + '.downsample.1', 256)
        inp = 256
    pre = [256]
    for s in (2, 3, 4):
        sc = extra['STAGE{}'.format(s)]
        ch = list(sc['NUM_CHANNELS'])
        tp = 'transition{}'.format(s - 1)
        for i in range(len(ch)):
            if i < len(pre):
                if ch[i] != pre[i]:
                    conv('{}.{}.0'.format(tp, i), ch[i], pre[i], 3); bn('{}.{}.1'.format(tp, i), ch[i])
            else:
                for j in range(i + 1 - len(pre)):
                    co = ch[i] if j == i - len(pre) else pre[-1]
                    conv('{}.{}.{}.0'.format(tp, i, j), co, pre[-1], 3); bn('{}.{}.{}.1'.format(tp, i, j), co)
        for m in range(sc['NUM_MODULES']):
            mp = 'stage{}.{}'.format(s, m)
            for i, c in enumerate(ch):
                for k in range(sc['NUM_BLOCKS'][i]):
                    b = '{}.branches.{}.{}'.format(mp, i, k)
                    conv(b + '.conv1', c, c, 3); bn(b + '.bn1', c)
                    conv(b + '.conv2', c, c, 3); bn(b + '.bn2', c)
            for i in range(len(ch)):
                for j in range(len(ch)):
                    f = '{}.fuse_layers.{}.{}'.format(mp, i, j)
                    if j > i:
                        conv(f + '.0', ch[i], ch[j], 1); bn(f + '.1', ch[i])
                    elif j < i:
                        for k in range(i - j):
                            co = ch[i] if k == i - j - 1 else ch[j]
                            conv('{}.{}.0'.format(f, k), co, ch[j], 3); bn('{}.{}.1'.format(f, k), co)
        pre = ch
    tot = int(sum(pre))
    conv('last_layer.0', tot, tot, 1, bias=True); bn('last_layer.1', tot)
    conv('last_layer.3', num_joints, tot, extra.get('FINAL_CONV_KERNEL', 1), bias=True)
    return t


def final_preds_oracle(post_process, batch_heatmaps, center, scale):
    """lib/core/inference.py:49-85 restated in numpy loops; the image-space map is solved from the three
    point pairs of lib/utils/transforms.py:58-90 (what cv2.getAffineTransform computes), rot = 0."""
    import numpy as np
    b, k, h, w = batch_heatmaps.shape
    flat = batch_heatmaps.reshape(b, k, -1)
    idx = flat.argmax(2)
    maxvals = flat.max(2).reshape(b, k, 1)
    coords = np.zeros((b, k, 2), dtype=np.float32)
    coords[..., 0] = idx % w
    coords[..., 1] = np.floor(idx / w)
    coords *= (maxvals > 0.0).astype(np.float32)
    if post_process:
        for n in range(b):
            for p in range(k):
                hm = batch_heatmaps[n][p]
                px = int(np.floor(coords[n][p][0] + 0.5))
                py = int(np.floor(coords[n][p][1] + 0.5))
                if 1 < px < w - 1 and 1 < py < h - 1:
                    diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(b):
        sc = np.asarray(scale[i], dtype=np.float32) * 200.0
        src_w = sc[0]
        src = np.zeros((3, 2), dtype=np.float32)
        dst = np.zeros((3, 2), dtype=np.float32)
        src[0] = center[i]
        src[1] = np.asarray(center[i], dtype=np.float32) + np.array([0, src_w * -0.5], np.float32)
        dst[0] = [w * 0.5, h * 0.5]
        dst[1] = np.array([w * 0.5, h * 0.5], np.float32) + np.array([0, w * -0.5], np.float32)
        for pts in (src, dst):
            d = pts[0] - pts[1]
            pts[2] = pts[1] + np.array([-d[1], d[0]], dtype=np.float32)
        # affine t with t @ [x, y, 1] = src for the three dst points
        a = np.concatenate([dst.astype(np.float64), np.ones((3, 1))], axis=1)
        t = np.linalg.solve(a, src.astype(np.float64)).T          # 2 x 3
        for p in range(k):
            preds[i, p] = t @ np.array([coords[i, p, 0], coords[i, p, 1], 1.0])
    return preds, maxvals
