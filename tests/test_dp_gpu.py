"""Data-parallel gradient exchange on the real backward program: 2 ranks (both on cuda:0, gloo carrying
the CUDA tensors) - the bucketed, overlapped all-reduce must leave exactly the sum of the ranks' local
gradients in the flat gradient buffer (the RCCL run at N>1 is the driver's)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd')


def _worker(rank, world, port, q):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
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
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(model.state_dict(), 2).items()}
        model.load_state_dict(sd)
        model = model.cuda().train()
        b = synth.rhd_batch(2, seed=40 + rank, img_h=128, img_w=128)
        x, gt = torch.from_numpy(b['imgs']).cuda(), torch.from_numpy(b['heatmaps']).cuda()
        crit = HeatmapLoss()

        def run():
            model.zero_grad()
            crit(model(x)[0], gt).backward()
        run()                                             # local gradient, no exchange
        net = model.hip()
        local = net.flat_g.clone()
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        want = parts[0] + parts[1]
        sync = GradSync(model, bucket_bytes=8 << 20)      # small buckets: several overlapped exchanges
        run()
        sync.finish()
        torch.cuda.synchronize()
        got = net.flat_g
        ok = bool(torch.equal(got, want))
        q.put((rank, ok, len(sync.cuts), float((got - want).abs().max())))
        dist.destroy_process_group()
    except Exception as e:   # surface the failure in the parent
        import traceback
        q.put((rank, False, -1, traceback.format_exc()))


def test_two_rank_gradient_exchange_on_the_recorded_backward():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _, _ in res), res
    assert all(n >= 2 for _, _, n, _ in res), res          # the exchange really was bucketed
