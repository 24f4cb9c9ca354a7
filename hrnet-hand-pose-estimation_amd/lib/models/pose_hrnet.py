"""pose_hrnet — MI355X-native drop-in for the reference's lib/models/pose_hrnet.py.

Same plugin surface (reference lib/models/pose_hrnet.py:603-609, dispatched by
`eval(cfg.MODEL.NAME + '.get_pose_net')(cfg, is_train=...)`, tools/train.py:152):

    get_pose_net(cfg, is_train, **kwargs) -> nn.Module
    module(x: (B,3,H,W) float NCHW on the HIP device) -> (heatmaps (B,K,H/4,W/4), inter_feat (B,C0,H/4,W/4))
    module.state_dict(): the reference's 1839 keys (OIHW f32 conv weights, BatchNorm buffers),
    so reference checkpoints load with strict=True.

The module tree below exists only to own parameters/buffers under the reference's names; no
leaf module's forward ever runs. `forward` hands the input to hipnet (recorded programs of
hand-written HIP kernels, see hipnet/engine.py); there is NO CPU or eager-PyTorch fallback — a
CPU tensor or a missing libhrnet_hip.so raises.

`cfg.MODEL.COMPUTE_DTYPE` ('fp32' | 'bf16', this build's only extra key) selects the device
arithmetic: fp32 = exact f32 MFMA (parity path), bf16 = bf16 MFMA with f32 accumulation and
f32 master weights/statistics.
"""
import logging
import os

import torch
import torch.nn as nn

from hipnet.engine import PlanTicket
from hipnet.net import HipNet

BN_MOMENTUM = 0.1
logger = logging.getLogger(__name__)


def _conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)


class _Unit(nn.Module):
    """Parameter container: convN/bnN children (+ optional `downsample`) with the reference names."""

    def __init__(self, widths, kernels, inplanes, downsample_to=None):
        super(_Unit, self).__init__()
        c = inplanes
        for n, (w, k) in enumerate(zip(widths, kernels), 1):
            setattr(self, 'conv{}'.format(n), _conv(c, w, k))
            setattr(self, 'bn{}'.format(n), nn.BatchNorm2d(w, momentum=BN_MOMENTUM))
            c = w
        if downsample_to is not None:
            self.downsample = nn.Sequential(_conv(inplanes, downsample_to, 1),
                                            nn.BatchNorm2d(downsample_to, momentum=BN_MOMENTUM))


class BasicBlock(_Unit):
    """3x3 -> 3x3 residual unit (reference pose_hrnet.py:28-57); container only."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__([planes, planes], [3, 3], inplanes)


class Bottleneck(_Unit):
    """1x1 -> 3x3 -> 1x1 (x4) residual unit (reference pose_hrnet.py:60-98); container only."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__([planes, planes, planes * 4], [1, 3, 1], inplanes,
                                         downsample_to=planes * 4 if inplanes != planes * 4 else None)


blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


def _conv_bn(cin, cout, k, stride, relu):
    layers = [_conv(cin, cout, k, stride), nn.BatchNorm2d(cout)]
    if relu:
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """Branches + fuse layers of one exchange unit (reference pose_hrnet.py:101-266); container only."""

    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super(HighResolutionModule, self).__init__()
        for what, seq in (('NUM_BLOCKS', num_blocks), ('NUM_CHANNELS', num_channels),
                          ('NUM_INCHANNELS', num_inchannels)):
            if num_branches != len(seq):
                msg = 'NUM_BRANCHES({}) <> {}({})'.format(num_branches, what, len(seq))
                logger.error(msg)
                raise ValueError(msg)
        if blocks is not BasicBlock:
            raise ValueError('hipnet supports BLOCK: BASIC inside HighResolutionModule')
        self.num_branches = num_branches
        self.fuse_method = fuse_method
        self.multi_scale_output = multi_scale_output
        ch = [c * blocks.expansion for c in num_channels]
        if list(num_inchannels) != ch:
            raise ValueError('hipnet expects stage inputs to carry NUM_CHANNELS channels')
        self.num_inchannels = ch
        self.branches = nn.ModuleList(
            nn.Sequential(*[blocks(ch[i], num_channels[i]) for _ in range(num_blocks[i])])
            for i in range(num_branches))
        rows = []
        for i in range(num_branches if multi_scale_output else 1):
            row = []
            for j in range(num_branches):
                if j > i:
                    row.append(nn.Sequential(_conv(ch[j], ch[i], 1), nn.BatchNorm2d(ch[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:
                    row.append(nn.Sequential(*[
                        _conv_bn(ch[j], ch[i] if k == i - j - 1 else ch[j], 3, 2, relu=(k != i - j - 1))
                        for k in range(i - j)]))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows) if num_branches > 1 else None

    def get_num_inchannels(self):
        return self.num_inchannels


class PoseHighResolutionNet(nn.Module):

    def __init__(self, cfg, **kwargs):
        super(PoseHighResolutionNet, self).__init__()
        self._build(cfg, **kwargs)

    def _build(self, cfg, **kwargs):
        extra = cfg.MODEL.EXTRA
        self.conv1 = _conv(3, 64, 3, 2)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = _conv(64, 64, 3, 2)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(*[Bottleneck(64 if k == 0 else 256, 64) for k in range(4)])

        pre = [256]
        self._stage_cfg = {}
        for s in (2, 3, 4):
            sc = cfg['MODEL']['EXTRA']['STAGE{}'.format(s)]
            block = blocks_dict[sc['BLOCK']]
            ch = [c * block.expansion for c in sc['NUM_CHANNELS']]
            setattr(self, 'stage{}_cfg'.format(s), sc)
            self._stage_cfg[s] = dict(NUM_MODULES=sc['NUM_MODULES'], NUM_BRANCHES=sc['NUM_BRANCHES'],
                                      NUM_BLOCKS=list(sc['NUM_BLOCKS']), NUM_CHANNELS=list(sc['NUM_CHANNELS']))
            if sc['NUM_BRANCHES'] != len(pre) + 1 and s > 2:
                raise ValueError('hipnet expects each stage to add exactly one branch')
            setattr(self, 'transition{}'.format(s - 1), self._make_transition_layer(pre, ch))
            mods = [HighResolutionModule(sc['NUM_BRANCHES'], block, sc['NUM_BLOCKS'], ch, sc['NUM_CHANNELS'],
                                         sc['FUSE_METHOD'], True) for _ in range(sc['NUM_MODULES'])]
            setattr(self, 'stage{}'.format(s), nn.Sequential(*mods))
            pre = ch
        tot = int(sum(pre))
        k = extra.FINAL_CONV_KERNEL
        if k != 1:
            raise ValueError('hipnet supports FINAL_CONV_KERNEL: 1')
        self.last_layer = nn.Sequential(
            _conv(tot, tot, 1, bias=True), nn.BatchNorm2d(tot, momentum=BN_MOMENTUM), nn.ReLU(inplace=False),
            _conv(tot, cfg.MODEL.NUM_JOINTS, k, bias=True))
        self.pretrained_layers = cfg['MODEL']['EXTRA']['PRETRAINED_LAYERS']
        cd = str(cfg.MODEL.get('COMPUTE_DTYPE', 'fp32')).lower()
        if cd not in ('fp32', 'bf16'):
            raise ValueError('MODEL.COMPUTE_DTYPE must be fp32 or bf16, got {}'.format(cd))
        self.compute_dtype = torch.float32 if cd == 'fp32' else torch.bfloat16
        self._hip = None
        self._anchor = None

    @staticmethod
    def _make_transition_layer(pre, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(_conv_bn(pre[i], c, 3, 1, True) if c != pre[i] else None)
            else:
                n = i + 1 - len(pre)
                layers.append(nn.Sequential(*[
                    _conv_bn(pre[-1], c if j == n - 1 else pre[-1], 3, 2, True) for j in range(n)]))
        return nn.ModuleList(layers)

    # ---- device state -----------------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        self._hip = None            # storage may move (.cuda(), .to(), .half()): rebuild lazily
        self._anchor = None
        return super(PoseHighResolutionNet, self)._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        r = super(PoseHighResolutionNet, self).load_state_dict(*a, **kw)
        if self._hip is not None:
            self._hip.mark_weights_dirty()
        return r

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('compute dtype must be torch.float32 or torch.bfloat16')
        self.compute_dtype = dtype
        self._hip = None

    def hip(self):
        if self._hip is None:
            self._hip = HipNet(self, self._stage_cfg, self.compute_dtype)
        return self._hip

    def invalidate_weights(self):
        """call after editing parameters through `.data` (in-place optimizers are tracked)"""
        if self._hip is not None:
            self._hip.mark_weights_dirty()

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError('pose_hrnet: input must be a tensor on the HIP device - move the model and the input '
                               'with .cuda() first (callers that run a CPU forward before .cuda(), like the '
                               "reference's get_model_summary at tools/train.py:196-200, must do it after); "
                               'this build has no CPU path')
        net = self.hip()
        x = x.contiguous().float()
        if self.training and torch.is_grad_enabled():
            if self._anchor is None:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            return _HRNetFunction.apply(x, self._anchor, self)
        hm, inter, _ = net.forward(x, training=self.training, need_grad=False)
        return hm, inter

    def init_weights(self, pretrained=''):
        """reference pose_hrnet.py:570-600: conv N(0, 0.001), bias 0, BN (1, 0); optional checkpoint."""
        logger.info('=> init weights from normal distribution')
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.normal_(std=0.001)
                    if m.bias is not None:
                        m.bias.zero_()
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.fill_(1)
                    m.bias.zero_()
        if os.path.isfile(pretrained):
            state = torch.load(pretrained, map_location='cpu')
            logger.info('=> loading pretrained model {}'.format(pretrained))
            keep = {k: v for k, v in state.items()
                    if k.split('.')[0] in self.pretrained_layers or self.pretrained_layers[0] == '*'}
            self.load_state_dict(keep, strict=False)
        elif pretrained:
            logger.error('=> please download pre-trained models first!')
            raise ValueError('{} does not exist!'.format(pretrained))
        self.invalidate_weights()


class _HRNetFunction(torch.autograd.Function):
    """Autograd anchor: forward/backward are the recorded HIP programs; parameter gradients are
    written straight into the flat gradient buffer that every `param.grad` views."""

    @staticmethod
    def forward(ctx, x, anchor, module):
        net = module.hip()
        ctx.set_materialize_grads(False)
        hm, inter, plan = net.forward(x, training=True, need_grad=True)
        ctx.net, ctx.plan = net, plan
        ctx.ticket = PlanTicket(plan)
        ctx.hook = getattr(module, '_segment_hook', None)
        return hm, inter

    @staticmethod
    def backward(ctx, g_hm, g_inter):
        plan = ctx.plan
        if ctx.ticket is None or not ctx.ticket.valid():
            raise RuntimeError('pose_hrnet: the activations of this forward pass are gone (backward already ran '
                               'for it, or another forward reused its buffers); call forward again')
        if g_hm is None:
            g_hm = torch.zeros((plan.N, plan.nj, plan.out_act.H, plan.out_act.W), dtype=torch.float32,
                               device=plan.dev)
        g_hm = g_hm.contiguous().float()
        if g_inter is not None:
            g_inter = g_inter.contiguous().float()
        ctx.net.backward(plan, g_hm, g_inter, ctx.hook)
        ctx.ticket.release()
        ctx.ticket = None
        return None, None, None


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
