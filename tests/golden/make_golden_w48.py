"""Golden fixture for the w48 geometry (BASELINE config 4), from the reference's own modules (build container only).

    python tests/golden/make_golden_w48.py            # writes tests/golden/w48_eval_b1.npz

Runs the reference's pose_hrnet.py AND pose_hrnet_softmax.py built from its w48 yaml
(experiments/RHD/RHD_HRNet_w48_softmax_hm-pose2dloss_v1.yaml: channels 48/96/192/384) in eval mode on one
384x288 synthetic crop with BN statistics calibrated on a second one; stores checksums + a slice of the heat
maps of both heads and the calibrated statistics of a few layers (weights come from the portable PRNG).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

YAML = 'experiments/RHD/RHD_HRNet_w48_softmax_hm-pose2dloss_v1.yaml'


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if not hasattr(np, 'int'):
        np.int = int
    out = {}
    for tag, rel in (('plain', 'lib/models/pose_hrnet.py'), ('softmax', 'lib/models/pose_hrnet_softmax.py')):
        ref = G._load('ref_w48_' + tag, rel)
        cfg = G.get_cfg_defaults()
        cfg.merge_from_file(os.path.join(G.REF, YAML))
        model = ref.get_pose_net(cfg, is_train=False)
        G.load_synth_weights(model, salt=6)
        calib = torch.from_numpy(G.synth.rhd_batch(1, seed=78, img_h=384, img_w=288)['imgs'])
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for m in bns:
            m.momentum = 1.0
        model.train()
        with torch.no_grad():
            model(calib)
        model.eval()
        x = torch.from_numpy(G.synth.rhd_batch(1, seed=2, img_h=384, img_w=288)['imgs'])
        with torch.no_grad():
            res = model(x)
        hm = res[0]
        out[tag + '.heatmaps_checksum'] = G.checksum(hm)
        out[tag + '.heatmaps_slice'] = hm[0, :, 40, 20:44].numpy().copy()
        out[tag + '.shape'] = np.array(hm.shape)
        if tag == 'softmax':
            sd = model.state_dict()
            for k in ('last_layer.1.running_mean', 'last_layer.1.running_var'):
                out['softmax.stat.' + k] = sd[k].numpy().copy()
        if tag == 'plain':
            sd = model.state_dict()
            out['n_params'] = np.int64(sum(p.numel() for p in model.parameters()))
            for k, v in sd.items():
                if k.endswith(('running_mean', 'running_var')):
                    out['stat.' + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'w48_eval_b1.npz'), **out)
    print('w48_eval_b1.npz', os.path.getsize(os.path.join(HERE, 'w48_eval_b1.npz')), out['plain.shape'], out['n_params'])


if __name__ == '__main__':
    main()
