"""End-to-end plumbing on the GPU box: tools/train.py for 2 tiny epochs (checkpoint + resume), then
tools/evaluate_2D.py on the produced state (strict load) - config 1 of BASELINE.json, B=4."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd')


def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout + r.stderr


def test_train_then_evaluate_cli(tmp_path):
    cfg = os.path.join(PKG, 'experiments', 'RHD', 'RHD_HRNet_w32_max_hmloss_v1.yaml')
    out = str(tmp_path / 'out')
    common = ['OUTPUT_DIR', out, 'LOG_DIR', str(tmp_path / 'log'), 'TRAIN.IMAGES_PER_GPU', '4', 'TEST.IMAGES_PER_GPU', '4',
              'PRINT_FREQ', '1']
    log = _run([sys.executable, 'tools/train.py', '--cfg', cfg, '--batches-per-epoch', '3', 'TRAIN.BEGIN_EPOCH', '0',
                'TRAIN.END_EPOCH', '1'] + common, PKG)
    assert 'Speed' in log and 'samples/s' in log and 'HeatmapLoss' in log
    exp = os.path.join(out, 'RHD', 'RHD_HRNet_w32_max_hmloss_v1')
    assert os.path.exists(os.path.join(exp, 'checkpoint.pth.tar')) and os.path.exists(os.path.join(exp, 'final_state.pth.tar'))
    # AUTO_RESUME picks the checkpoint up and continues with epoch 1
    log2 = _run([sys.executable, 'tools/train.py', '--cfg', cfg, '--batches-per-epoch', '2', 'TRAIN.BEGIN_EPOCH', '0',
                 'TRAIN.END_EPOCH', '2'] + common, PKG)
    assert 'resumed from' in log2 and 'Epoch: [1]' in log2
    ev = _run([sys.executable, 'tools/evaluate_2D.py', '--cfg', cfg, '--model_path', os.path.join(exp, 'final_state.pth.tar'),
               '--batch_size', '4', '--num_batches', '4', '--gpu', '0', 'OUTPUT_DIR', out], PKG)
    assert 'fps:' in ev and 'PCK@20px' in ev
    import numpy as np
    res = os.path.join(out, 'eval2D_results_RHD_HRNet_w32_max_hmloss_v1')
    pck = np.loadtxt(os.path.join(res, 'PCK2d.txt'))          # the reference's format: thresholds row, PCK row
    mse = np.loadtxt(os.path.join(res, 'mse2d_each_joint.txt'))
    assert pck.shape == (2, 49) and np.array_equal(pck[0], np.arange(1, 50)) and mse.shape == (21,)
    assert np.all(np.diff(pck[1]) >= 0) and 0.0 <= pck[1, 0] and pck[1, -1] <= 1.0


def test_train_cli_with_the_trainable_softmax_variant(tmp_path):
    """pose_hrnet_softmax through tools/train.py - expectation decode + key-point loss, the
    temperature learns (reference experiments/RHD/RHD_HRNet_w32_trainable_softmax_pose2dloss_v1.yaml)."""
    cfg = os.path.join(PKG, 'experiments', 'RHD', 'RHD_HRNet_w32_trainable_softmax_pose2dloss_v1.yaml')
    out = str(tmp_path / 'out')
    log = _run([sys.executable, 'tools/train.py', '--cfg', cfg, '--batches-per-epoch', '3', 'TRAIN.BEGIN_EPOCH', '0',
                'TRAIN.END_EPOCH', '1', 'OUTPUT_DIR', out, 'LOG_DIR', str(tmp_path / 'log'), 'TRAIN.IMAGES_PER_GPU', '4',
                'TEST.IMAGES_PER_GPU', '4', 'PRINT_FREQ', '1'], PKG)
    assert 'Speed' in log and 'Pose2DLoss' in log
    exp = os.path.join(out, 'RHD', 'RHD_HRNet_w32_trainable_softmax_pose2dloss_v1')
    import torch
    sd = torch.load(os.path.join(exp, 'final_state.pth.tar'), map_location='cpu')
    sd = sd.get('state_dict', sd)
    key = [k for k in sd if k.endswith('trainable_temp')][0]
    assert float(sd[key]) != 1.0          # the temperature moved
