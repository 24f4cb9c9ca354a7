"""pose_hrnet_PoseAggr — the reference's temporal pose aggregation ("PoseWarper") variant on the HIP path
(reference lib/models/pose_hrnet_PoseAggr.py): inference (`USE_WARPING_TEST`) and training of the
aggregation head (`USE_WARPING_TRAIN`).

Input: 5·B frames ordered [prev2 | prev1 | current | next1 | next2] (reference :593-611). The backbone + head of
pose_hrnet_softmax produce the per-frame heat-map logits; then (reference :612-646)

    diff      = logits[current] (tiled 5x) - logits                       frame differences
    feats     = offset_feats(diff)                                        20 BasicBlocks, 21 -> 128 channels
    off_k     = offsets_k(feats),  k = 1..5                               3x3 convs with dilation 3, 6, 12, 18, 24
    warped    = 0.2 * sum_k deform_conv_k(logits, off_k)                  21 deformable groups, dilation as above
    out       = 0.3 w[cur] + 0.25 w[prev1] + 0.25 w[next1] + 0.1 w[prev2] + 0.1 w[next2]
    heatmaps  = spatial softmax(out * trainable_temp)

every step a HIP launch through the C ABI: hrnet_lincomb_f32, hrnet_conv2d (+ BatchNorm prologues from the running
statistics), hrnet_sum_terms, hrnet_conv2d_dilated3x3, hrnet_deform_conv_forward, hrnet_spatial_softmax_fwd. The head
runs op by op (it is not part of the recorded programs).

TRAINING (`USE_WARPING_TRAIN`, reference :703-733): the reference freezes the backbone and trains offset_feats, the
offset convs and the deformable convs. Here the backbone runs its recorded training-mode forward without a backward
program (batch statistics, running statistics updated, no gradient - as with the reference's frozen parameters,
whose only gradient consumer would be the backbone itself), and the head runs as op-by-op autograd layers
(hipnet/eager.py: ConvBN, AddReLU, DilatedConv, LinComb + the deformable-conv and softmax Functions), each forward
and backward a HIP launch. The head's gradients arrive in `param.grad` like any autograd gradient (use a torch
optimiser over the head's parameters).

state_dict(): the reference's keys in the reference's order - trainable_temp, backbone, offset_feats.{0..19},
offsets1..5, deform_conv1..5 (weight, bias).
"""
import ctypes
import os

import torch
import torch.nn as nn

from deformable_conv import DeformConv
from hipnet import _capi as C
from hipnet import eager as E
from models.pose_hrnet import BN_MOMENTUM, BasicBlock, Bottleneck, HighResolutionModule, _Unit, blocks_dict  # noqa: F401
from models.pose_hrnet_softmax import PoseHighResolutionNet as _SoftmaxNet
from models.pose_hrnet_softmax import _SpatialSoftmax

INNER_CH = 128          # reference :359
CHAIN_BLOCKS = 20       # reference :362
FRAME_WEIGHTS = (0.1, 0.25, 0.3, 0.25, 0.1)      # prev2, prev1, current, next1, next2 (reference :640)


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _lincomb(out, srcs, coefs):
    n = out.numel()
    assert all(s.numel() == n and s.is_contiguous() and s.dtype == torch.float32 for s in srcs) and out.is_contiguous()
    C.call('hrnet_lincomb_f32', out.data_ptr(), n, len(srcs), _ptrs(srcs), (ctypes.c_float * len(coefs))(*coefs),
           C.stream_ptr())


class PoseHighResolutionNet(_SoftmaxNet):
    def __init__(self, cfg, is_train=False, **kwargs):
        nn.Module.__init__(self)
        self.trainable_temp = nn.Parameter(torch.tensor(1.0), requires_grad=bool(cfg.MODEL.TRAINABLE_SOFTMAX))
        self._build(cfg, **kwargs)
        self.use_warping_train = bool(cfg.MODEL.USE_WARPING_TRAIN)
        self.use_warping_test = bool(cfg.MODEL.USE_WARPING_TEST)
        self.is_train = is_train
        self.flag = (is_train and self.use_warping_train) or (not is_train and self.use_warping_test)   # reference :293
        nj = int(cfg.MODEL.NUM_JOINTS)
        self.num_joints = nj
        self.dilation_rates = [int(d) for d in cfg.MODEL.DILATION_RATES]
        if self.flag:
            # reference :441-485: BasicBlock(nj -> 128) with a 1x1 conv + BatchNorm on the identity, then 19 x BasicBlock(128)
            first = _Unit([INNER_CH, INNER_CH], [3, 3], nj, downsample_to=INNER_CH)
            self.offset_feats = nn.Sequential(first, *[BasicBlock(INNER_CH, INNER_CH) for _ in range(CHAIN_BLOCKS - 1)])
            for k, d in enumerate(self.dilation_rates, 1):          # reference :497-506
                setattr(self, 'offsets{}'.format(k), nn.Conv2d(INNER_CH, nj * 2 * 9, kernel_size=3, stride=1, dilation=d,
                                                               padding=d, bias=False))
            for k, d in enumerate(self.dilation_rates, 1):          # reference :508-516
                setattr(self, 'deform_conv{}'.format(k), DeformConv(nj, nj, (3, 3), stride=1, padding=d, dilation=d,
                                                                    deformable_groups=nj))
        self._tap_cache = {}

    # ---- the aggregation head, op by op ------------------------------------------------------------------
    def _dilated_taps(self, name, conv, dt, tdt):
        """nine packed 1x1 matrices of a dilated 3x3 conv (repacked when the weight tensor changes)"""
        key = (name, conv.weight.data_ptr(), conv.weight._version)
        hit = self._tap_cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2], hit[3]
        w = conv.weight.detach().float()
        co, ci = w.shape[0], w.shape[1]
        cop, cip = (co + 15) // 16 * 16, (ci + 7) // 8 * 8
        taps = torch.empty((9, cop * cip), dtype=tdt, device=w.device)
        for t in range(9):
            wt = w[:, :, t // 3, t % 3].contiguous()
            C.call('hrnet_pack_weights', dt, wt.data_ptr(), taps[t].data_ptr(), co, ci, 1, cop, cip, 0, C.stream_ptr())
        self._tap_cache[name] = (key, taps, cop, cip)
        return taps, cop, cip

    def _aggregate(self, x, net, plan):
        """x: (5B, nj, H, W) f32 logits on the device -> (B, nj, H, W) aggregated logits"""
        T, nj, H, W = x.shape
        if T % 5:
            raise ValueError('pose_hrnet_PoseAggr expects 5 frames per sample (batch {} is not a multiple of 5)'.format(T))
        B = T // 5
        dt, tdt, dev = net.dtid, net.compute_dtype, x.device
        st = C.stream_ptr
        per = B * nj * H * W
        xs = [x[g * B:(g + 1) * B] for g in range(5)]
        diff = torch.empty_like(x)
        for g in range(5):
            _lincomb(diff[g * B:(g + 1) * B], [xs[2], xs[g]], [1.0, -1.0])
        convs, bns = net.convs, plan.bns
        cin0 = convs['offset_feats.0.conv1'].Cin_pad
        a = torch.empty((T, H, W, cin0), dtype=tdt, device=dev)
        C.call('hrnet_nchw_to_nhwc', dt, diff.data_ptr(), a.data_ptr(), T, H, W, cin0, nj, st())

        def conv(src, name, bn=None):
            r = convs[name]
            y = torch.empty((T, H, W, r.Cout_pad), dtype=tdt, device=dev)
            C.call('hrnet_conv2d', dt, src.data_ptr(), r.wf.data_ptr(), bn.scale.data_ptr() if bn else None,
                   bn.shift.data_ptr() if bn else None, None, y.data_ptr(), None, T, H, W, r.Cin_pad, H, W, r.Cout_pad,
                   r.ks, 1, 0, 1 if bn else 0, 0, st())
            return y

        def residual(y, bn, other, bn_other):
            out = torch.empty_like(y)
            C.call('hrnet_sum_terms', dt, out.data_ptr(), T, H, W, y.shape[3], 2, _ptrs([y, other]),
                   _ptrs([bn.scale, bn_other.scale if bn_other else None]),
                   _ptrs([bn.shift, bn_other.shift if bn_other else None]), (ctypes.c_int * 2)(0, 0),
                   (ctypes.c_int * 2)(0, 0), 1, st())
            return out

        cur = a
        for k in range(CHAIN_BLOCKS):
            p = 'offset_feats.{}'.format(k)
            y1 = conv(cur, p + '.conv1')
            y2 = conv(y1, p + '.conv2', bns[p + '.bn1'])
            if k == 0:
                yd = conv(cur, p + '.downsample.0')
                cur = residual(y2, bns[p + '.bn2'], yd, bns[p + '.downsample.1'])
            else:
                cur = residual(y2, bns[p + '.bn2'], cur, None)
        warped = []
        noff = nj * 18
        for k, d in enumerate(self.dilation_rates, 1):
            oc = getattr(self, 'offsets{}'.format(k))
            taps, cop, cip = self._dilated_taps('offsets{}'.format(k), oc, dt, tdt)
            off_nhwc = torch.empty((T, H, W, cop), dtype=tdt, device=dev)
            C.call('hrnet_conv2d_dilated3x3', dt, cur.data_ptr(), taps.data_ptr(), taps.stride(0) * taps.element_size(),
                   off_nhwc.data_ptr(), T, H, W, cip, cop, d, st())
            off = torch.empty((T, noff, H, W), dtype=torch.float32, device=dev)
            C.call('hrnet_nhwc_to_nchw', dt, off_nhwc.data_ptr(), off.data_ptr(), T, H, W, cop, noff, st())
            warped.append(getattr(self, 'deform_conv{}'.format(k))(x, off).contiguous())
        w = torch.empty_like(x)
        _lincomb(w, warped, [0.2] * len(warped))
        out = torch.empty((B, nj, H, W), dtype=torch.float32, device=dev)
        _lincomb(out, [w[g * B:(g + 1) * B] for g in range(5)], list(FRAME_WEIGHTS))
        del per
        return out

    def _aggregate_train(self, x, net):
        """the same head as autograd layers over the C ABI (training-mode BatchNorm); x: (5B, nj, H, W) logits, no grad"""
        T, nj, H, W = x.shape
        if T % 5:
            raise ValueError('pose_hrnet_PoseAggr expects 5 frames per sample (batch {} is not a multiple of 5)'.format(T))
        B = T // 5
        tdt = net.compute_dtype
        xs = [x[g * B:(g + 1) * B] for g in range(5)]
        diff = torch.empty_like(x)
        for g in range(5):
            _lincomb(diff[g * B:(g + 1) * B], [xs[2], xs[g]], [1.0, -1.0])
        cur = E.ToNHWC.apply(diff, net.convs['offset_feats.0.conv1'].Cin_pad, tdt)
        for unit in self.offset_feats:
            cur = E.basic_block(cur, unit)
        warped = []
        for k, d in enumerate(self.dilation_rates, 1):
            off = E.ToNCHW.apply(E.DilatedConv.apply(cur, getattr(self, 'offsets{}'.format(k)).weight, d), nj * 18)
            warped.append(getattr(self, 'deform_conv{}'.format(k))(x, off))
        w = E.LinComb.apply([0.2] * len(warped), *warped)
        return E.LinComb.apply(list(FRAME_WEIGHTS), *[w[g * B:(g + 1) * B] for g in range(5)])

    def forward(self, x):
        warping = (self.training and self.use_warping_train) or (not self.training and self.use_warping_test)   # :611
        if not (warping and self.flag):
            return _SoftmaxNet.forward(self, x)
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError('pose_hrnet_PoseAggr: input must be a tensor on the HIP device (.cuda()); no CPU path')
        net = self.hip()
        if self.training and torch.is_grad_enabled():
            with torch.no_grad():
                logits, _, _ = net.forward(x.contiguous().float(), training=True, need_grad=False)
            agg = self._aggregate_train(logits.contiguous(), net)
            return _SpatialSoftmax.apply(agg, self.trainable_temp), self.trainable_temp
        if self.training:
            raise NotImplementedError('pose_hrnet_PoseAggr: a training-mode pass without gradients (batch statistics in '
                                      'the aggregation head, no autograd) is not built; use model.eval() for inference')
        with torch.no_grad():
            logits, _, plan = net.forward(x.contiguous().float(), training=False, need_grad=False)
            agg = self._aggregate(logits.contiguous(), net, plan)
            heat = _SpatialSoftmax.apply(agg, self.trainable_temp)
        return heat, self.trainable_temp


    def init_weights(self, pretrained=''):
        """reference pose_hrnet_PoseAggr.py:647-730 ("PoseWarper initialization"): every conv N(0, 0.001) / bias 0 and
        every BatchNorm (1, 0) - all FROZEN (requires_grad False); the deformable convs start as the identity (centre
        tap of channel k -> k is 1) and train; optional checkpoint; then, with USE_WARPING_TRAIN, the offset-feature
        chain trains again and the five dilated offset convs are zeroed (the warp starts as the identity) and train."""
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.normal_(std=0.001)
                    m.weight.requires_grad = False
                    if m.bias is not None:
                        m.bias.zero_()
                        m.bias.requires_grad = False
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.fill_(1)
                    m.bias.zero_()
                    m.weight.requires_grad = False
                    m.bias.requires_grad = False
                elif isinstance(m, DeformConv) and self.flag:
                    m.weight.zero_()
                    kh, kw = m.weight.shape[2] // 2, m.weight.shape[3] // 2
                    for k in range(m.weight.shape[0]):
                        m.weight[k, k, kh, kw] = 1.0
                    m.weight.requires_grad = True
        if os.path.isfile(pretrained):
            state = torch.load(pretrained, map_location='cpu')
            keep = {k: v for k, v in state.items()
                    if k.split('.')[0] in self.pretrained_layers or self.pretrained_layers[0] == '*'}
            self.load_state_dict(keep, strict=False)
        elif pretrained:
            raise ValueError('{} does not exist!'.format(pretrained))
        if self.use_warping_train and self.flag:
            for m in self.offset_feats.modules():
                if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                    m.weight.requires_grad = True
                    if m.bias is not None:
                        m.bias.requires_grad = True
            with torch.no_grad():
                for k in range(1, len(self.dilation_rates) + 1):
                    oc = getattr(self, 'offsets{}'.format(k))
                    oc.weight.zero_()
                    oc.weight.requires_grad = True
        self._tap_cache = {}
        self.invalidate_weights()


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, is_train, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
