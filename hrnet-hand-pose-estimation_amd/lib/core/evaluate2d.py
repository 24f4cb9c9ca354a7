"""2-D evaluation on the output side of the hot path (reference tools/evaluate_2D.py:118-131 checkpoint
loading, :165-169,:235-245,:268-294 metrics): per-joint end-point error weighted by visibility, PCK for
thresholds 1..49 px, and the two result files in the reference's format so that its committed
tools/eval2D_results_*/ files are regression targets once a dataset is present."""
import os

import numpy as np
import torch


def load_checkpoint_state(model, path, map_location='cpu'):
    """torch.load(path) -> optional 'state_dict' entry -> strip the `module.` prefix DataParallel / DDP
    checkpoints carry (tools/evaluate_2D.py:127-129, tools/train.py:166-168) -> load_state_dict(strict=True)."""
    sd = torch.load(path, map_location=map_location)
    if isinstance(sd, dict) and 'state_dict' in sd and not torch.is_tensor(sd['state_dict']):
        sd = sd['state_dict']
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    return model


class Eval2DAccumulator(object):
    """running sums of tools/evaluate_2D.py:165-169,268-274"""

    def __init__(self, n_joints, hm_size):
        self.K, self.hm_size = n_joints, float(hm_size)
        self.th = np.arange(1, 50)
        self.pck = np.zeros(len(self.th))
        self.mse = np.zeros(n_joints)
        self.vis = np.zeros(n_joints)

    def add(self, pred, gt, visibility, crop_size=None, corner=None, orig_size=None):
        """pred / gt (B,K,2) in heat-map pixels; visibility (B,K[,1]); RHD: crop_size (B,), corner (B,2);
        otherwise orig_size = (width, height) of the image the coordinates are scaled to"""
        pred = np.asarray(pred, dtype=np.float64)
        gt = np.asarray(gt, dtype=np.float64)
        vis = np.asarray(visibility, dtype=np.float64).reshape(pred.shape[0], self.K)
        if crop_size is not None:
            cs = np.asarray(crop_size, dtype=np.float64).reshape(-1, 1, 1) / self.hm_size
            co = np.asarray(corner, dtype=np.float64).reshape(-1, 1, 2)
            pred, gt = pred * cs + co, gt * cs + co
        else:
            s = np.array([orig_size[0] / self.hm_size, orig_size[1] / self.hm_size])
            pred, gt = pred * s, gt * s
        each = np.linalg.norm(pred - gt, axis=2) * vis
        self.mse += each.sum(0)
        self.vis += vis.sum(0)
        self.pck += ((each[None] < self.th[:, None, None]) * vis[None]).sum((1, 2))

    def result(self):
        with np.errstate(invalid='ignore', divide='ignore'):
            mse = self.mse / self.vis                # a joint that was never visible gives nan, as the reference
            pck = self.pck / self.vis.sum()
        return mse, np.stack((self.th, pck))

    def save(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        mse, pck = self.result()
        np.savetxt(os.path.join(out_dir, 'mse2d_each_joint.txt'), mse, fmt='%.4f')
        np.savetxt(os.path.join(out_dir, 'PCK2d.txt'), pck)
        return mse, pck
