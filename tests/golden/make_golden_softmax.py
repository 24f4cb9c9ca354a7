"""Golden fixture of the reference's pose_hrnet_softmax variant (build container only).

    python tests/golden/make_golden_softmax.py        # writes tests/golden/w32_softmax_train_b2.npz

Runs /root/reference/lib/models/pose_hrnet_softmax.py (file-path import, same shims as make_golden.py)
on a 128x128 batch of 2 in train mode with the portable synthetic weights and stores the outputs:
soft-max heat maps, inter_feat checksum, the loss (reference HeatmapLoss on the soft-max maps plus a
small quadratic on inter_feat so that its gradient path is exercised), every parameter's gradient
checksum and the temperature's gradient.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if not hasattr(np, 'int'):
        np.int = int
    ref = G._load('ref_pose_hrnet_softmax', 'lib/models/pose_hrnet_softmax.py')
    loss_mod = G._load('ref_loss', 'lib/core/loss.py')
    cfg = G.get_cfg_defaults()
    cfg.merge_from_file(os.path.join(G.REF, 'experiments/RHD/RHD_HRNet_w32_trainable_softmax_pose2dloss_v1.yaml'))
    model = ref.get_pose_net(cfg, is_train=False)
    G.load_synth_weights(model, salt=5)
    with torch.no_grad():
        model.trainable_temp.fill_(1.5)
    model.train()
    b = G.synth.rhd_batch(2, seed=321, img_h=128, img_w=128)
    hm, inter, temp = model(torch.from_numpy(b['imgs']))
    gt = torch.from_numpy(b['heatmaps'])
    gt = gt / gt.sum((2, 3), keepdim=True).clamp_min(1e-6)          # soft-max maps sum to one
    loss = loss_mod.HeatmapLoss()(hm, gt) * 1e4 + 1e-3 * inter.square().mean()
    loss.backward()
    named = dict(model.named_parameters())
    keys = list(model.state_dict().keys())
    out = {'state_keys_head': np.array(keys[:3]), 'n_state_entries': np.int64(len(keys)),
           'heatmaps': hm.detach().numpy(), 'inter_feat_checksum': G.checksum(inter),
           'inter_shape': np.array(inter.shape), 'loss': np.float64(loss.item()),
           'temp_grad': np.float64(model.trainable_temp.grad.item()),
           'grad_keys': np.array([k for k in keys if k in named]),
           'grad_checksums': np.array([[named[k].grad.double().sum().item(), named[k].grad.double().abs().sum().item()]
                                       for k in keys if k in named])}
    for k in ('last_layer.3.weight', 'stage4.2.fuse_layers.0.3.0.weight', 'conv1.weight'):
        out['grad.' + k] = named[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'w32_softmax_train_b2.npz'), **out)
    print('w32_softmax_train_b2.npz', os.path.getsize(os.path.join(HERE, 'w32_softmax_train_b2.npz')))
    print('loss', loss.item(), 'temp grad', model.trainable_temp.grad.item(), keys[:3])


if __name__ == '__main__':
    main()
