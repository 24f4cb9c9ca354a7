"""Training entry point with the reference's flags (tools/train.py:57-92):

    python tools/train.py --cfg experiments/RHD/RHD_HRNet_w32_max_hmloss_v1.yaml [KEY value ...]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train.py --cfg ...

cfg -> eval(cfg.MODEL.NAME + '.get_pose_net') -> criterion dict -> optimizer -> MultiStepLR ->
core.function.train / validate per epoch -> checkpoint.pth.tar / model_best.pth.tar /
final_state.pth.tar (tools/train.py:126-405). One process per GPU; multi-GPU = RCCL all-reduce of the
flat gradient overlapped with backward (hipnet.optim.GradSync), not DataParallel. Data is the
synthetic RHD-shaped loader (dataset/build.py) since no dataset ships with the repository.
"""
import argparse
import os
import pprint

import _init_paths  # noqa: F401
import torch

from config import cfg, update_config
from core.function import train, validate
from core.loss import HeatmapLoss, JointsMSELoss
from dataset.build import make_dataloader
from models import pose_hrnet, pose_hrnet_PoseAggr, pose_hrnet_softmax  # noqa: F401  (dispatched by name below)
from utils.utils import create_logger, get_optimizer, save_checkpoint


def parse_args():
    p = argparse.ArgumentParser(description='Train keypoints network')
    p.add_argument('--cfg', help='experiment configure file name', required=True, type=str)
    p.add_argument('opts', help='Modify config options using the command-line', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--gpus', help='gpus id for multiprocessing training', type=str)
    p.add_argument('--world-size', default=1, type=int)
    p.add_argument('--dist-url', default='tcp://127.0.0.1:23456', type=str)
    p.add_argument('--rank', default=0, type=int)
    p.add_argument('--local_rank', default=0, type=int)
    p.add_argument('--batches-per-epoch', default=8, type=int, help='synthetic loader length')
    return p.parse_args()


def main():
    args = parse_args()
    update_config(cfg, args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(args.local_rank)))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        torch.distributed.init_process_group(backend=cfg.DIST_BACKEND, device_id=device)
    master = rank == 0
    logger, final_output_dir, tb_log_dir = create_logger(cfg, args.cfg, 'train')
    if master:
        logger.info(pprint.pformat(vars(args)))

    model = eval(cfg.MODEL.NAME + '.get_pose_net')(cfg, is_train=True)
    best_perf, begin_epoch = float('inf'), cfg.TRAIN.BEGIN_EPOCH
    ckpt_file = os.path.join(final_output_dir, 'checkpoint.pth.tar')
    ckpt = None
    if cfg.AUTO_RESUME and os.path.exists(ckpt_file):
        ckpt = torch.load(ckpt_file, map_location='cpu')
        sd = {k[7:] if k.startswith('module.') else k: v for k, v in ckpt['state_dict'].items()}
        model.load_state_dict(sd, strict=True)
        begin_epoch, best_perf = ckpt['epoch'], ckpt.get('loss', best_perf)
        logger.info('=> resumed from {} (epoch {})'.format(ckpt_file, begin_epoch))
    model = model.to(device)
    sync = None
    if world > 1:
        from hipnet.optim import GradSync
        sync = GradSync(model)        # broadcasts rank 0's parameters / buffers, as DDP's constructor does

    criterion = {}
    if cfg.LOSS.WITH_HEATMAP_LOSS:
        criterion['heatmap_loss'] = HeatmapLoss().to(device)
    if cfg.LOSS.WITH_POSE2D_LOSS:
        criterion['pose2d_loss'] = JointsMSELoss().to(device)
    optimizer = get_optimizer(cfg, model)
    if sync is not None:
        sync.attach(optimizer)        # 1/world: folded into FlatAdam, applied in finish() for torch optimizers
    if ckpt is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    writer_dict = {'writer': None, 'train_global_steps': 0, 'valid_global_steps': 0}
    if ckpt is not None:
        writer_dict['train_global_steps'] = ckpt.get('train_global_steps', 0)
        writer_dict['valid_global_steps'] = ckpt.get('valid_global_steps', 0)

    def lr_at(epoch):
        return cfg.TRAIN.LR * (cfg.TRAIN.LR_FACTOR ** sum(epoch >= s for s in cfg.TRAIN.LR_STEP))

    train_loader = make_dataloader(cfg, True, world > 1, args.batches_per_epoch, rank, world)
    valid_loader = make_dataloader(cfg, False, world > 1, max(1, args.batches_per_epoch // 4), rank, world)
    for epoch in range(begin_epoch, cfg.TRAIN.END_EPOCH):
        for g in optimizer.param_groups:          # MultiStepLR(LR_STEP, LR_FACTOR)
            g['lr'] = lr_at(epoch)
        for loader in train_loader.values():
            loader.sampler.set_epoch(epoch)
        train(cfg, args, master, train_loader, model, criterion, optimizer, epoch, final_output_dir, tb_log_dir,
              writer_dict, logger, fp16=cfg.FP16.ENABLED, device=device)
        perf = best_perf
        if not cfg.WITHOUT_EVAL:
            recorder = validate(cfg, args, master, valid_loader, model, criterion, final_output_dir, tb_log_dir,
                                writer_dict, logger, device=device)
            perf = recorder.avg_total_loss
        is_best = perf < best_perf
        best_perf = min(best_perf, perf)
        if master:
            logger.info('=> saving checkpoint to {} (best: {})'.format(final_output_dir, is_best))
            save_checkpoint({'epoch': epoch + 1, 'model': cfg.MODEL.NAME, 'state_dict': model.state_dict(),
                             'loss': perf, 'optimizer': optimizer.state_dict(),
                             'train_global_steps': writer_dict['train_global_steps'],
                             'valid_global_steps': writer_dict['valid_global_steps']}, is_best, final_output_dir)
    if master:
        final = os.path.join(final_output_dir, 'final_state.pth.tar')
        logger.info('saving final model state to {}'.format(final))
        torch.save(model.state_dict(), final)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
