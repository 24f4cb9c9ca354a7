"""CPU-only tests: the C-ABI library loads and exports every symbol include/hrnet_hip.h declares,
host-side config / module surface / synthetic data, and the 2-rank gradient exchange on gloo."""
import os
import re
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd')
YAML = os.path.join(PKG, 'experiments', 'RHD', 'RHD_HRNet_w32_max_hmloss_v1.yaml')


def test_library_exports_every_declared_symbol():
    from hipnet import _capi
    header = open(os.path.join(REPO, 'include', 'hrnet_hip.h')).read()
    declared = set(re.findall(r'\b(hrnet_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    lib = _capi.lib()                      # loads without a GPU; no compute call is made here
    for name in sorted(declared):
        assert hasattr(lib, name), 'libhrnet_hip.so lacks ' + name
    assert declared == set(_capi.EXPORTED), declared ^ set(_capi.EXPORTED)
    assert _capi.call('hrnet_abi_version') == _capi.ABI_VERSION == 2


def test_host_side_shape_helpers_are_pure():
    from hipnet import _capi as C
    # statistics rows = workgroups along x: 64 images x 16 tiles of 16x16, a few tiles per workgroup
    rows = C.call('hrnet_conv_tiles', 64, 64, 64, 32, 3, 1)
    assert 128 <= rows <= 64 * 16 and (64 * 16) % rows == 0
    assert C.call('hrnet_conv_tiles', 2, 8, 8, 256, 3, 1) == 2
    assert C.call('hrnet_wgrad_splits', C.HR_BF16, 64, 64, 64, 32, 32, 3, 1) >= 1
    assert 1 <= C.call("hrnet_reduce_blocks", 64, 64, 64, 32) <= 2048


def test_config_merges_reference_style_yaml_and_freezes():
    from config import get_cfg_defaults, update_config

    class A:
        cfg = YAML
        opts = ['TRAIN.LR', '0.01', 'GPUS', '(0,1)']
    cfg = get_cfg_defaults()
    update_config(cfg, A)
    assert cfg.GPUS == (0, 1) and cfg.TRAIN.LR == 0.01
    assert cfg.MODEL.EXTRA.STAGE3.NUM_CHANNELS == [32, 64, 128]
    assert cfg['MODEL']['EXTRA']['STAGE4']['BLOCK'] == 'BASIC'          # item access, as pose_hrnet.py:292
    with pytest.raises(AttributeError):
        cfg.TRAIN.LR = 1.0
    with pytest.raises(KeyError):
        c2 = get_cfg_defaults()
        c2.merge_from_list(['NO.SUCH.KEY', '1'])
    with pytest.raises(ValueError):
        c3 = get_cfg_defaults()
        c3.merge_from_list(['TRAIN.LR', 'fast'])


def test_module_surface_matches_reference_state_dict():
    from config import get_cfg_defaults
    from models import pose_hrnet
    from oracle import hrnet_cpu as O
    cfg = get_cfg_defaults()
    cfg.merge_from_file(YAML)
    model = eval('pose_hrnet.get_pose_net')(cfg, is_train=False)     # the reference's dispatch idiom
    sd = model.state_dict()
    tmpl = O.state_template()
    assert list(sd.keys()) == list(tmpl.keys())
    assert all(tuple(sd[k].shape) == tuple(tmpl[k]) for k in tmpl)
    assert sum(p.numel() for p in model.parameters()) == 29547477
    with pytest.raises(RuntimeError, match='no CPU path'):
        model(torch.zeros(1, 3, 64, 64))
    # reference error conventions (pose_hrnet.py:119-137)
    with pytest.raises(ValueError, match='NUM_BRANCHES'):
        pose_hrnet.HighResolutionModule(2, pose_hrnet.BasicBlock, [4], [32, 64], [32, 64], 'SUM')
    cfg.defrost()
    cfg.MODEL.INIT_WEIGHTS = True
    cfg.MODEL.PRETRAINED = '/nonexistent/checkpoint.pth'
    with pytest.raises(ValueError, match='does not exist'):
        pose_hrnet.get_pose_net(cfg, is_train=True)


def test_softmax_variant_module_surface():
    from config import get_cfg_defaults
    from models import pose_hrnet_softmax
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(PKG, 'experiments', 'RHD', 'RHD_HRNet_w32_trainable_softmax_pose2dloss_v1.yaml'))
    m = eval('pose_hrnet_softmax.get_pose_net')(cfg, is_train=False)
    keys = list(m.state_dict().keys())
    assert keys[0] == 'trainable_temp' and len(keys) == 1840            # reference order: own parameter first
    assert m.trainable_temp.requires_grad and float(m.trainable_temp) == 1.0
    assert m.head_align_corners and m.inter_from_cat
    cfg.MODEL.TRAINABLE_SOFTMAX = False
    assert not pose_hrnet_softmax.get_pose_net(cfg, is_train=False).trainable_temp.requires_grad


def test_poseaggr_variant_module_surface():
    """pose_hrnet_PoseAggr (reference lib/models/pose_hrnet_PoseAggr.py:286-372): the softmax variant's keys, then the
    offset-feature chain (first block 21 -> 128 with a 1x1 conv + BatchNorm on the identity, 19 more of 128), five
    dilated offset convs (128 -> 21*2*9) and five deformable convs (weight, bias) - 1840 + 261 keys"""
    from config import get_cfg_defaults
    from models import pose_hrnet_PoseAggr
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(PKG, 'experiments', 'RHD', 'RHD_HRNet_w32_trainable_softmax_pose2dloss_v1.yaml'))
    m = eval('pose_hrnet_PoseAggr.get_pose_net')(cfg, is_train=False)
    sd = m.state_dict()
    keys = list(sd.keys())
    assert keys[0] == 'trainable_temp' and len(keys) == 1840 + 261
    assert keys[1840] == 'offset_feats.0.conv1.weight' and keys[-1] == 'deform_conv5.bias'
    assert tuple(sd['offset_feats.0.conv1.weight'].shape) == (128, 21, 3, 3)
    assert tuple(sd['offset_feats.0.downsample.0.weight'].shape) == (128, 21, 1, 1)
    assert tuple(sd['offset_feats.19.conv2.weight'].shape) == (128, 128, 3, 3)
    for k, d in enumerate((3, 6, 12, 18, 24), 1):
        conv = getattr(m, 'offsets{}'.format(k))
        assert tuple(conv.weight.shape) == (378, 128, 3, 3) and conv.dilation == (d, d) and conv.padding == (d, d)
        dcn = getattr(m, 'deform_conv{}'.format(k))
        assert tuple(dcn.weight.shape) == (21, 21, 3, 3) and dcn.dilation == (d, d) and dcn.deformable_groups == 21
    # without warping at test time the model is the plain softmax variant (reference :611)
    cfg.MODEL.USE_WARPING_TEST = False
    plain = pose_hrnet_PoseAggr.get_pose_net(cfg, is_train=False)
    assert len(plain.state_dict()) == 1840 and not plain.flag


def test_poseaggr_init_freezes_the_backbone_and_starts_as_the_identity_warp():
    """reference lib/models/pose_hrnet_PoseAggr.py:647-730 (init_weights, "PoseWarper initialization"), read from the
    source (the reference module cannot be imported here: its deformable-conv extension is CUDA-only): every conv /
    BatchNorm frozen, the offset-feature chain re-enabled, offsets1..5 zero and trainable, deform_conv1..5 identity
    centre taps and trainable; get_optimizer then builds a torch optimiser over the trainable parameters only"""
    from config import get_cfg_defaults
    from models import pose_hrnet_PoseAggr
    from utils.utils import get_optimizer
    cfg = get_cfg_defaults()
    # the PoseAggr experiment file (the reference's key names: experiments/MHP/..._PoseAggr_v1.yaml)
    cfg.merge_from_file(os.path.join(PKG, 'experiments', 'MHP', 'MHP_HRNet_w32_trainable_softmax_pose2dloss_PoseAggr_v1.yaml'))
    assert cfg.MODEL.NAME == 'pose_hrnet_PoseAggr' and list(cfg.MODEL.DILATION_RATES) == [3, 6, 12, 18, 24]
    assert cfg.MODEL.USE_WARPING_TRAIN and cfg.MODEL.USE_WARPING_TEST and cfg.MODEL.TRAINABLE_SOFTMAX
    cfg.MODEL.INIT_WEIGHTS = True
    cfg.MODEL.PRETRAINED = ''
    m = eval('pose_hrnet_PoseAggr.get_pose_net')(cfg, is_train=True)       # (tools/train.py:113 dispatches by eval)
    named = dict(m.named_parameters())
    head = ('offset_feats.', 'offsets', 'deform_conv')
    for name, p in named.items():
        if name == 'trainable_temp':
            continue
        if name.startswith(head):
            # (a DeformConv built with bias=True keeps a trainable bias; reference modules/deform_conv.py:38-41)
            assert p.requires_grad, name
        else:
            assert not p.requires_grad, name                      # the whole backbone, convs and BatchNorms
    w = named['stage3.0.branches.1.0.conv1.weight']
    assert abs(float(w.std()) - 1e-3) < 1e-4
    assert float((named['offset_feats.3.bn1.weight'] - 1).abs().max()) == 0.0
    for k in range(1, 6):
        assert float(named['offsets{}.weight'.format(k)].abs().max()) == 0.0
        dw = named['deform_conv{}.weight'.format(k)]
        eye = torch.zeros_like(dw)
        for c in range(dw.shape[0]):
            eye[c, c, 1, 1] = 1.0
        assert torch.equal(dw.detach(), eye)
    opt = get_optimizer(cfg, m)
    assert isinstance(opt, torch.optim.Optimizer) and type(opt).__module__.startswith('torch.optim')
    ids = {id(p) for g in opt.param_groups for p in g['params']}
    assert ids == {id(p) for p in m.parameters() if p.requires_grad}
    # the plain model keeps the fused flat-buffer optimiser
    from models import pose_hrnet
    cfg2 = get_cfg_defaults()
    cfg2.merge_from_file(YAML)
    assert not any(not p.requires_grad for p in pose_hrnet.get_pose_net(cfg2, is_train=False).parameters())


def test_tools_dispatch_every_model_family_by_name():
    """tools/train.py:152 and tools/evaluate_2D.py:92 of the reference resolve eval(cfg.MODEL.NAME + '.get_pose_net'):
    the three families this build has must be importable names in both tools"""
    for tool in ('train.py', 'evaluate_2D.py'):
        src = open(os.path.join(PKG, 'tools', tool)).read()
        line = next(l for l in src.splitlines() if l.startswith('from models import'))
        for name in ('pose_hrnet', 'pose_hrnet_softmax', 'pose_hrnet_PoseAggr'):
            assert name in [t.strip() for t in line.split('import', 1)[1].split('#')[0].split(',')], (tool, name)


def test_init_weights_follows_reference_distribution():
    from config import get_cfg_defaults
    from models import pose_hrnet
    cfg = get_cfg_defaults()
    cfg.merge_from_file(YAML)
    model = pose_hrnet.get_pose_net(cfg, is_train=False)
    model.init_weights('')
    w = model.stage3[0].branches[1][0].conv1.weight
    assert abs(float(w.std()) - 1e-3) < 1e-4 and abs(float(w.mean())) < 1e-4
    assert float(model.last_layer[0].bias.abs().max()) == 0.0
    assert float((model.bn1.weight - 1).abs().max()) == 0.0


def test_synthetic_data_is_portable_and_reference_shaped():
    from hipnet import synth
    a, b = synth.rhd_batch(2, seed=7), synth.rhd_batch(2, seed=7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a['imgs'].shape == (2, 3, 256, 256) and a['heatmaps'].shape == (2, 21, 64, 64)
    assert a['pose2d'].shape == (2, 21, 2) and a['visibility'].shape == (2, 21, 1)
    vis = a['visibility'][..., 0]
    peak = a['heatmaps'].reshape(2, 21, -1).max(-1)
    assert np.all(peak[vis] == 1.0) and np.all(peak[~vis] == 0.0)
    # known-answer pins of the counter-based generator (bit-exact across machines)
    u = synth.uniform01(synth.key_seed('conv1.weight'), 4)
    assert u.dtype == np.float32 and np.all((u >= 0) & (u < 1))
    assert np.array_equal(u, synth.uniform01(synth.key_seed('conv1.weight'), 8)[:4])


def _gradsync_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(PKG, 'lib'))
    from hipnet.optim import GradSync

    class FakeConv:
        pass

    class FakeNet:
        pass

    class FakePlan:
        pass

    class FakeModel:
        pass
    # 6 "layers" of 1000 floats, flat order = forward order; backward finishes them last-to-first
    net = FakeNet()
    net.total_params = 6000
    net.flat_g = torch.full((6000,), float(rank + 1))
    net.convs, net.offsets = {}, {}
    for i in range(6):
        c = FakeConv(); c.mod = FakeConv(); c.mod.weight = object()
        net.convs['l{}'.format(i)] = c
        net.offsets[id(c.mod.weight)] = (i * 1000, 1000)
    plan = FakePlan()
    plan.net = net
    plan.bwd = list(range(60))
    plan.bucket_marks = [((6 - i) * 10, 'l{}'.format(i)) for i in range(5, -1, -1)]   # op index after layer i
    class FakeFlatAdam:
        grad_scale = 1.0
    total = float(sum(range(1, world + 1)))
    # (1) an optimizer that takes the 1/world factor itself (FlatAdam): the buffer keeps the SUM
    opt = FakeFlatAdam()
    sync = GradSync(FakeModel(), optimizer=opt, bucket_bytes=8000, broadcast=False)       # 2 layers per bucket
    sync.begin(plan)
    cuts = list(sync.cuts)
    # engine.Plan._run_segments: hook.after(c) for every cut strictly inside, then after(end)
    for c in [c for c in cuts if 0 < c < len(plan.bwd)] + [len(plan.bwd)]:
        sync.after(c)
    sync.finish()
    ok = bool(torch.all(net.flat_g == total)) and opt.grad_scale == 1.0 / world
    # the plan bench.py prints at N>1: every float of the flat gradient in exactly one bucket, >= 8000 bytes each
    d = sync.describe()
    spans = sorted((b['offset'], b['offset'] + b['floats']) for b in d['buckets'])
    ok = ok and d['payload'] == 'f32' and spans[0][0] == 0 and spans[-1][1] == 6000
    ok = ok and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # (2) any other optimizer (torch.optim.SGD ...): finish() leaves the MEAN, as DDP does
    net.flat_g.fill_(float(rank + 1))
    sync = GradSync(FakeModel(), optimizer=object(), bucket_bytes=8000, broadcast=False)
    sync.begin(plan)
    for c in [c for c in sync.cuts if 0 < c < len(plan.bwd)] + [len(plan.bwd)]:
        sync.after(c)
    sync.finish()
    ok = ok and bool(torch.all(net.flat_g == total / world))
    # (3) a backward that never passed the bucket marks (run in pieces around an external gradient): finish()
    # still exchanges every range exactly once
    net.flat_g.fill_(float(rank + 1))
    sync.begin(plan)
    sync.after(sync.cuts[0])
    sync.finish()
    ok = ok and bool(torch.all(net.flat_g == total / world))
    # (4) a plan recorded for a single process (deferred weight gradients: nothing is final at a mark) - built before
    # GradSync was attached - is exchanged in ONE all-reduce after the pass, never in part
    import warnings
    net.flat_g.fill_(float(rank + 1))
    plan2 = FakePlan()
    plan2.net, plan2.bwd, plan2.bucket_marks, plan2.defer_wgrad = net, plan.bwd, plan.bucket_marks, True
    with warnings.catch_warnings(record=True) as wlog:
        warnings.simplefilter('always')
        sync.begin(plan2)
    ok = ok and sync.cuts == [] and any('deferred weight gradients' in str(w.message) for w in wlog)
    for c in cuts:
        sync.after(c)                                   # marks the backward run passes: nothing may leave here
    ok = ok and bool(torch.all(net.flat_g == float(rank + 1)))
    sync.after(len(plan.bwd))
    sync.finish()
    ok = ok and bool(torch.all(net.flat_g == total / world))
    # (5) a plan recorded FOR data parallelism (engine.Plan.dp_plan): the deferred launches only touch the flat buffer's
    # late region [late_start, total) - the main region goes in buckets at the marks, the late one when the pass ends
    net.flat_g.fill_(float(rank + 1))
    net.late_start = 4000                                # layers 4, 5 sit in the late region; marks cover l0..l3
    plan3 = FakePlan()
    plan3.net, plan3.bwd, plan3.defer_wgrad, plan3.dp_plan = net, plan.bwd, True, True
    plan3.bucket_marks = [((4 - i) * 10, 'l{}'.format(i)) for i in range(3, -1, -1)]
    sync.begin(plan3)
    d = sync.describe()
    ok = ok and len(sync.cuts) >= 1 and [b for b in d['buckets'] if b.get('late_region')][0]['offset'] == 4000
    spans = sorted((b['offset'], b['offset'] + b['floats']) for b in d['buckets'])
    ok = ok and spans[0][0] == 0 and spans[-1][1] == 6000 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sync.after(sync.cuts[0])
    for w_ in sync._works:
        w_.wait()
    ok = ok and bool(torch.all(net.flat_g[4000:] == float(rank + 1)))      # the late region has not left yet
    for c in sync.cuts[1:] + [len(plan3.bwd)]:
        sync.after(c)
    sync.finish()
    ok = ok and bool(torch.all(net.flat_g == total / world))
    # (6) ... and with the late region in GROUPS (engine.Plan.late_cuts: op index, flat range, side lane): every group
    # leaves at its own cut, before the program's last op, and nothing of it is left for the end
    net.flat_g.fill_(float(rank + 1))
    net.trainable_count = 6000
    plan4 = FakePlan()
    plan4.net, plan4.bwd, plan4.defer_wgrad, plan4.dp_plan = net, plan.bwd, True, True
    plan4.bucket_marks = plan3.bucket_marks
    plan4.late_cuts = [(33, 5500, 6000, 1), (35, 5000, 5500, 1), (37, 4000, 5000, 1)]
    sync.begin(plan4)
    d = sync.describe()
    groups = [b for b in d['buckets'] if b.get('late_group')]
    ok = ok and len(groups) == 3 and not [b for b in d['buckets'] if b.get('late_region')]
    ok = ok and d['exposed_mb_after_backward'] <= 4000 * 4 / 1e6 and sync.cuts == sorted(sync.cuts)
    spans = sorted((b['offset'], b['offset'] + b['floats']) for b in d['buckets'])
    ok = ok and spans[0][0] == 0 and spans[-1][1] == 6000 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    issued_before_end = 0
    for c in [c for c in sync.cuts if c < len(plan4.bwd)]:
        n0 = len(sync._works)
        sync.after(c)
        issued_before_end += len(sync._works) - n0
    for w_ in sync._works:
        w_.wait()
    ok = ok and issued_before_end >= 3 + 1 and bool(torch.all(net.flat_g[4000:] == total))    # the late region is done
    sync.after(len(plan4.bwd))
    sync.finish()
    ok = ok and bool(torch.all(net.flat_g == total / world))
    # groups that do not tile the late region are ignored (one piece at the end, as in (5))
    plan4.late_cuts = [(33, 5500, 6000, 1)]
    sync._plan = None
    net.flat_g.fill_(float(rank + 1))
    sync.begin(plan4)
    ok = ok and not sync._late_ranges and [b for b in sync.describe()['buckets'] if b.get('late_region')][0]['offset'] == 4000
    for c in sync.cuts + [len(plan4.bwd)]:
        sync.after(c)
    sync.finish()
    ok = ok and bool(torch.all(net.flat_g == total / world))
    q.put((rank, ok, cuts))
    dist.destroy_process_group()


def test_gradient_exchange_two_ranks_gloo():
    """N>1 path: bucketed sum-all-reduce of the flat gradient covers every element exactly once."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_gradsync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2] and len(res[0][2]) >= 2     # same bucket cuts on both ranks


def test_flat_adam_moments_are_remapped_by_parameter_name_across_flat_layouts():
    """FlatAdam checkpoints carry [(name, offset, numel)]: moments written under the module-order flat buffer (round 2)
    or under another permutation land on the same PARAMETERS after load, and a foreign checkpoint is refused."""
    from hipnet.optim import legacy_layout, remap_flat
    order = ['a.weight', 'stage2.0.fuse_layers.0.1.0.weight', 'b.weight', 'stage3.0.branches.2.0.conv1.weight', 'c.bias']
    size = {'a.weight': 6, 'stage2.0.fuse_layers.0.1.0.weight': 4, 'b.weight': 3,
            'stage3.0.branches.2.0.conv1.weight': 5, 'c.bias': 2}
    # current layout: main region in module order, then the late region (fuse layers / wide branches)
    cur_names = ['a.weight', 'b.weight', 'c.bias', 'stage2.0.fuse_layers.0.1.0.weight', 'stage3.0.branches.2.0.conv1.weight']
    cur, off = [], 0
    for n in cur_names:
        cur.append((n, off, size[n]))
        off += size[n]
    total = off
    old = legacy_layout(cur, order + ['frozen.temp'])
    assert [n for n, _o, _k in old] == order and old[1] == ('stage2.0.fuse_layers.0.1.0.weight', 6, 4)
    saved = torch.zeros(total)
    val = {n: float(i + 1) for i, n in enumerate(order)}
    for n, o, k in old:
        saved[o:o + k] = val[n]
    got = remap_flat(saved, old, cur, total)
    for n, o, k in cur:
        assert torch.all(got[o:o + k] == val[n]), n
    # same layout: the tensor passes through untouched; round trip old -> cur -> old restores the original
    assert remap_flat(got, cur, cur, total) is got
    assert torch.equal(remap_flat(got, cur, old, total), saved)
    # (layouts come back from torch.load as lists of lists)
    assert torch.equal(remap_flat(saved, [list(t) for t in old], cur, total), got)
    with pytest.raises(ValueError):
        remap_flat(saved, old[:-1], cur, total)
    bad = [(n, o, k + (1 if n == 'b.weight' else 0)) for n, o, k in old]
    with pytest.raises(ValueError):
        remap_flat(saved, bad, cur, total)
