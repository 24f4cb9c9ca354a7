"""Host helpers on the boundary of the hot path (reference lib/utils/utils.py:37-115):
create_logger, get_optimizer, save_checkpoint."""
import logging
import os
import time
from pathlib import Path

import torch
import torch.optim as optim


def create_logger(cfg, cfg_name, phase='train'):
    """<OUTPUT_DIR>/<yaml's directory name>/<EXP_NAME>/ + a log file; returns (logger, output dir, tb dir)."""
    out_root = Path(cfg.OUTPUT_DIR or 'output')
    dataset = os.path.basename(os.path.dirname(cfg_name)) or 'RHD'
    final_output_dir = out_root / dataset / cfg.EXP_NAME
    final_output_dir.mkdir(parents=True, exist_ok=True)
    time_str = time.strftime('%Y-%m-%d-%H-%M')
    stem = os.path.basename(cfg_name).split('.')[0]
    logging.basicConfig(filename=str(final_output_dir / '{}_{}_{}.log'.format(stem, time_str, phase)),
                        format='%(asctime)-15s %(message)s')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    if not any(isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler)
               for h in logger.handlers):
        logger.addHandler(logging.StreamHandler())
    tb_dir = Path(cfg.LOG_DIR or 'log') / dataset / cfg.EXP_NAME / (stem + '_' + time_str)
    tb_dir.mkdir(parents=True, exist_ok=True)
    return logger, str(final_output_dir), str(tb_dir)


def get_optimizer(cfg, model):
    """cfg.TRAIN.OPTIMIZER in {sgd, adam, adamw}; one parameter group carrying `initial_lr`.

    'adam' on a hipnet model returns hipnet.optim.FlatAdam (one fused HIP kernel over the flat
    parameter buffer, torch.optim.Adam arithmetic); HRNET_TORCH_OPTIM=1 forces torch.optim."""
    name = cfg.TRAIN.OPTIMIZER.lower()
    params = [{'params': [p for p in model.parameters() if p.requires_grad], 'initial_lr': cfg.TRAIN.LR}]
    inner = model.module if hasattr(model, 'module') else model
    # FlatAdam steps the whole flat parameter buffer from the flat gradient buffer: right only when every parameter
    # trains through the recorded programs. A model with frozen parameters (pose_hrnet_PoseAggr freezes its backbone,
    # reference pose_hrnet_PoseAggr.py:647-730, and its head's gradients arrive as autograd .grad tensors) gets the
    # torch optimiser over the parameters that require gradients, as the reference builds it (utils.py:83).
    frozen = any(not p.requires_grad for n, p in inner.named_parameters() if n != 'trainable_temp')
    if name == 'adam' and hasattr(inner, 'hip') and not frozen and os.environ.get('HRNET_TORCH_OPTIM', '0') != '1':
        from hipnet.optim import FlatAdam
        return FlatAdam(inner, lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    if name == 'sgd':
        return optim.SGD(params, lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD,
                         nesterov=cfg.TRAIN.NESTEROV)
    if name == 'adam':
        return optim.Adam(params, lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    if name == 'adamw':
        return optim.AdamW(params, lr=cfg.TRAIN.LR, weight_decay=cfg.TRAIN.WD)
    return None


def save_checkpoint(states, is_best, output_dir, filename='checkpoint.pth.tar'):
    torch.save(states, os.path.join(output_dir, filename))
    if is_best and 'state_dict' in states:
        torch.save(states['state_dict'], os.path.join(output_dir, 'model_best.pth.tar'))
