"""Data-parallel gradient exchange on the real backward program: 2 ranks (both on cuda:0, gloo carrying
the CUDA tensors) - installing GradSync broadcasts rank 0's parameters / buffers to ranks that started
from different ones, and the bucketed, overlapped all-reduce must leave exactly the mean of the ranks' local
gradients in the flat gradient buffer (torch.optim.SGD has no gradient scale to fold 1/world into), also when
a gradient on inter_feat splits the backward run in two (the RCCL run at N>1 is the driver's)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd')


def _worker(rank, world, port, q, atomic='0'):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ['HRNET_DETERMINISTIC'] = '1'      # ordered batch statistics: the two runs see the same activations
        # atomic '0': weight gradients through slabs + ordered sums, the two runs are compared bit for bit;
        # '1': by float atomics into the flat gradient (the default of a training run), compared to f32 summation noise
        os.environ['HRNET_WGRAD_ATOMIC'] = atomic
        for p in (REPO, os.path.join(PKG, 'lib')):
            if p not in sys.path:
                sys.path.insert(0, p)
        import numpy as np
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from config import get_cfg_defaults
        from core.loss import HeatmapLoss
        from hipnet import synth
        from hipnet.optim import GradSync
        from models import pose_hrnet
        cfg = get_cfg_defaults()
        cfg.merge_from_file(os.path.join(PKG, 'experiments', 'RHD', 'RHD_HRNet_w32_max_hmloss_v1.yaml'))
        cfg.MODEL.COMPUTE_DTYPE = 'fp32'
        model = pose_hrnet.get_pose_net(cfg, is_train=False)
        # every rank starts from DIFFERENT weights and BatchNorm buffers: installing GradSync must make them rank 0's
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(model.state_dict(), 2 + rank).items()}
        model.load_state_dict(sd)
        model = model.cuda().train()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)              # no grad_scale: finish() must average
        sync = GradSync(model, optimizer=opt, bucket_bytes=2 << 20)    # small buckets: several overlapped exchanges over the main region
        net = model.hip()
        ps = [torch.empty_like(net.flat_p) for _ in range(world)]
        dist.all_gather(ps, net.flat_p)
        rv = [torch.empty_like(model.bn1.running_var) for _ in range(world)]
        dist.all_gather(rv, model.bn1.running_var)
        ref0 = torch.from_numpy(np.asarray(synth.fill_for_key('conv2.weight', tuple(model.conv2.weight.shape), 2))).cuda()
        same = bool(torch.equal(ps[0], ps[1]) and torch.equal(rv[0], rv[1]) and torch.equal(model.conv2.weight.detach(), ref0))
        b = synth.rhd_batch(2, seed=40 + rank, img_h=128, img_w=128)
        x, gt = torch.from_numpy(b['imgs']).cuda(), torch.from_numpy(b['heatmaps']).cuda()
        crit = HeatmapLoss()
        errs = []
        for with_inter in (False, True):
            def run():
                model.zero_grad()
                hm, inter = model(x)
                loss = crit(hm, gt)
                if with_inter:
                    loss = loss + inter.square().mean()     # a gradient on inter_feat: backward runs in two pieces
                loss.backward()
            model._segment_hook = None
            run()                                             # local gradient, no exchange
            local = net.flat_g.clone()
            parts = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(parts, local)
            want = (parts[0] + parts[1]) * (1.0 / world)
            model._segment_hook = sync
            run()
            sync.finish()
            torch.cuda.synchronize()
            got = net.flat_g
            errs.append(float((got - want).abs().max()))
            if atomic == '0':
                same = same and bool(torch.equal(got, want))
            else:
                rel = float((got - want).norm() / want.norm())
                errs.append(rel)
                same = same and rel <= 2e-5
        # the backward program was recorded FOR data parallelism (the process group was up when the plan was built):
        # weight-gradient launches are deferred to the single-lane tail only in the flat buffer's late region, which
        # leaves in its own exchange when the pass ends
        plan = sync._plan
        d = sync.describe()
        late = [b for b in d['buckets'] if b.get('late_region')]
        groups = sorted((b for b in d['buckets'] if b.get('late_group')), key=lambda b: b['offset'])
        same = same and plan.wgrad_atomic == (atomic == '1') and plan.dp_plan and plan.defer_wgrad and not plan.offload_wgrad and plan.n_deferred_wgrads > 20
        # the late region leaves in groups while the single-lane tail runs (engine.Plan.late_cuts): >= 6 exchanges issued
        # at cuts BEFORE the program's last op, tiling [late_start, trainable_count), nothing of it left for the end
        same = same and not late and len(groups) >= 6 and groups[0]['offset'] == net.late_start
        same = same and groups[-1]['offset'] + groups[-1]['floats'] == net.trainable_count
        same = same and all(a['offset'] + a['floats'] == b['offset'] for a, b in zip(groups, groups[1:]))
        same = same and all(g['after_op'] < d['backward_ops'] for g in groups) and d['exposed_mb_after_backward'] <= 30.0
        same = same and sum(g['floats'] for g in groups) > 10_000_000
        q.put((rank, same, len(sync.cuts), errs + [len(groups), d['exposed_mb_after_backward']]))
        dist.destroy_process_group()
    except Exception as e:   # surface the failure in the parent
        import traceback
        q.put((rank, False, -1, traceback.format_exc()))


@pytest.mark.parametrize('atomic', ['0', '1'])
def test_two_rank_gradient_exchange_on_the_recorded_backward(atomic):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port + int(atomic), q, atomic)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _, _ in res), res
    assert all(n >= 2 for _, _, n, _ in res), res          # the exchange really was bucketed
