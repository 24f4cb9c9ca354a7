"""Per-batch training / validation loops of the 2-D hot path (reference lib/core/function.py:
train_helper :24-162 generic branch :67-69,71-73,85-106; train :164-193; val_helper :635-774 generic
branch; validate :790-828; AverageMeter :1272-1378).

Same call signatures, loss-dict keys, log line format and tensorboard scalar names; only the
generic 2-D dataset branch (HandGraph_kpt / RHD_kpt / FreiHand_kpt / MHP_kpt ...) exists here -
the multi-view / temporal / CPM branches belong to other model families.

`debug`: the reference hard-codes a module-level `debug = True` that stops every epoch after 5
iterations (function.py:22,193,812). Here it defaults to False; set `core.function.debug = True`
to reproduce that behaviour.
"""
import time

import torch

from utils.heatmap_decoding import get_final_preds

debug = False

GENERIC_DATASETS = ('HandGraph_kpt', 'RHD_kpt', 'FreiHand_kpt', 'MHP_kpt', 'MHP_CPM_kpt', 'MHP_seq', 'synthetic_kpt')

_LOSS_NAMES = (('heatmap_loss', 'WITH_HEATMAP_LOSS', 'HeatmapLoss', 'heatmap_loss'),
               ('pose2d_loss', 'WITH_POSE2D_LOSS', 'Pose2DLoss', 'pose2d_loss'))


class AverageMeter(object):
    """Running sums of the loss terms; `computeLosses` also builds the total the step back-propagates."""

    def __init__(self, config, criterion):
        self.config = config
        self.criterion = criterion
        L = config.LOSS
        # Running sums stay ON THE DEVICE: the reference adds `loss.item()` three times per step
        # (lib/core/function.py:1334-1344), i.e. three host syncs that empty the launch queue - +0.9 ms on a 15.5 ms
        # step here (bench.py --pose2d-loss). The attributes below read the same numbers, syncing only when they are
        # read (every PRINT_FREQ steps and at the end of an epoch).
        self._sums = {'total_loss': 0.}
        if L.WITH_HEATMAP_LOSS:
            self._sums['heatmap_loss'] = 0.
        if L.WITH_POSE2D_LOSS:
            self._sums['pose2d_loss'] = 0.
        self.pose3d_loss = 0. if L.WITH_POSE3D_LOSS else None
        self.time_consistency_loss = 0. if L.WITH_TIME_CONSISTENCY_LOSS else None
        self.bone_loss = 0. if L.WITH_BONE_LOSS else None
        self.jointangle_loss = 0. if L.WITH_JOINTANGLE_LOSS else None
        self.n = 0

    def _read(self, key):
        v = self._sums.get(key)
        return None if v is None else (float(v.item()) if hasattr(v, 'item') else float(v))

    total_loss = property(lambda self: self._read('total_loss'))
    heatmap_loss = property(lambda self: self._read('heatmap_loss'))
    pose2d_loss = property(lambda self: self._read('pose2d_loss'))

    def computeAvgLosses(self):
        n = max(self.n, 1)
        self.avg_total_loss = self.total_loss / n
        out = {'total_loss': self.avg_total_loss}
        if self.config.LOSS.WITH_HEATMAP_LOSS:
            self.avg_heatmap_loss = out['heatmap_loss'] = self.heatmap_loss / n
        if self.config.LOSS.WITH_POSE2D_LOSS:
            self.avg_pose2d_loss = out['pose2d_loss'] = self.pose2d_loss / n
        return out

    def computeLosses(self, heatmaps_pred=None, heatmaps_gt=None, pose2d_pred=None, pose2d_gt=None,
                      visibility=None, pose3d_pred=None, pose3d_gt=None, n=1):
        self.n += n
        out = dict.fromkeys(('heatmap_loss', 'pose2d_loss', 'pose3d_loss', 'TC_loss', 'jointangle_loss', 'bone_loss'))
        total = 0
        names = self.criterion.keys()
        if 'heatmap_loss' in names:
            l = self.criterion['heatmap_loss'](heatmaps_pred, heatmaps_gt)
            self._sums['heatmap_loss'] = self._sums['heatmap_loss'] + l.detach()
            total = total + self.config.LOSS.HEATMAP_LOSS_FACTOR * l
            out['heatmap_loss'] = l
        if 'pose2d_loss' in names:
            l = self.criterion['pose2d_loss'](pose2d_pred[:, :, 0:2], pose2d_gt[:, :, 0:2], visibility=visibility)
            self._sums['pose2d_loss'] = self._sums['pose2d_loss'] + l.detach()
            total = total + self.config.LOSS.POSE2D_LOSS_FACTOR * l
            out['pose2d_loss'] = l
        for other in ('pose3d_loss', 'bone_loss', 'jointangle_loss'):
            if other in names:
                raise NotImplementedError('{} belongs to model families outside the HRNet 2-D hot path'.format(other))
        self._sums['total_loss'] = self._sums['total_loss'] + total.detach()
        out['total_loss'] = total
        return out


def _to_device(t, device):
    return t.cuda(device, non_blocking=True) if device is not None else t.cuda(non_blocking=True)


def _forward_and_losses(config, ret, model, recorder, device):
    imgs, heatmaps_gt, pose2d_gt, visibility = ret['imgs'], ret['heatmaps'], ret['pose2d'], ret['visibility']
    # pose_hrnet returns (heatmaps, inter_feat); pose_hrnet_softmax adds the temperature as a third item
    # (the reference's 2-tuple unpack at lib/core/function.py:68 cannot take that model)
    outputs = model(_to_device(imgs, device))
    heatmaps_pred = outputs[0]
    pose2d_pred = get_final_preds(heatmaps_pred, use_softmax=config.MODEL.HEATMAP_SOFTMAX)
    if config.LOSS.WITH_HEATMAP_LOSS:
        heatmaps_gt = _to_device(heatmaps_gt, device)
    if config.LOSS.WITH_POSE2D_LOSS:
        pose2d_gt = _to_device(pose2d_gt, device)
        visibility = _to_device(visibility, device)
    visibility = visibility.reshape(visibility.shape[0], -1)     # B x K (the reference squeezes)
    return imgs, recorder.computeLosses(heatmaps_pred, heatmaps_gt, pose2d_pred, pose2d_gt, visibility=visibility)


def _message(head, batch_time, nimg, loss_dict, recorder, always):
    msg = head + 'Time {:.3f}s\tSpeed {:.1f} samples/s\tTotalLoss {:.5f} ({:.5f})'.format(
        batch_time, nimg / batch_time, loss_dict['total_loss'].item(), recorder.avg_total_loss)
    for key, _flag, label, attr in _LOSS_NAMES:
        l = loss_dict[key]
        if l is not None and (always or l):
            msg += '\t{} {:.5f} ({:.5f})'.format(label, l.item(), getattr(recorder, 'avg_' + attr))
    return msg


def train_helper(epoch, i, args, config, master, ret, model, optimizer, dataset_name, train_loader, writer_dict,
                 logger, output_dir, tb_log_dir, pose3d_gt=None, recorder=None, fp16=False, device=None):
    end = time.time()
    imgs, loss_dict = _forward_and_losses(config, ret, model, recorder, device)
    total_loss = loss_dict['total_loss']
    optimizer.zero_grad()
    if fp16 and hasattr(optimizer, 'backward'):
        optimizer.backward(total_loss)          # an apex-style FP16_Optimizer wrapper (reference function.py:101-104)
    else:
        if fp16 and not getattr(train_helper, '_fp16_warned', False):
            train_helper._fp16_warned = True
            logger.warning('FP16.ENABLED is accepted but has no effect here: there is no loss scaling on this path; '
                           'reduced precision is MODEL.COMPUTE_DTYPE: bf16 (bf16 MFMA, f32 accumulation and '
                           'master weights)')
        total_loss.backward()
    sync = getattr(model, '_segment_hook', None)
    if sync is not None:
        sync.finish()              # gradient all-reduce issued during backward (hipnet.optim.GradSync)
    optimizer.step()
    batch_time = time.time() - end
    if i % config.PRINT_FREQ == 0 and master:
        recorder.computeAvgLosses()
        head = 'Dataset: {0} Epoch: [{1}][{2}/{3}]\t'.format(dataset_name, epoch, i, len(train_loader))
        logger.info(_message(head, batch_time, imgs.size(0), loss_dict, recorder, always=False))
        writer = writer_dict['writer']
        if writer is not None:
            steps = writer_dict['train_global_steps']
            for key, _flag, _label, _attr in _LOSS_NAMES:
                if loss_dict[key] is not None:
                    writer.add_scalar('train_loss/' + key, loss_dict[key], steps)
            writer.add_scalar('train_loss/total_loss', total_loss, steps)
    writer_dict['train_global_steps'] += 1


def train(config, args, master, train_loader_dict, model, criterion, optimizer, epoch, output_dir, tb_log_dir,
          writer_dict, logger, fp16=False, device=None):
    recorder = AverageMeter(config, criterion)
    model.train()
    for dataset_name, train_loader in train_loader_dict.items():
        logger.info('Training on {} dataset [Batch size: {}]\n'.format(dataset_name, train_loader.batch_size))
        if dataset_name not in GENERIC_DATASETS:
            raise NotImplementedError('dataset branch {} is outside the HRNet 2-D hot path'.format(dataset_name))
        for i, ret in enumerate(train_loader):
            if getattr(getattr(train_loader, 'dataset', None), 'exception', False):
                continue
            train_helper(epoch, i, args, config, master, ret, model, optimizer, dataset_name, train_loader,
                         writer_dict, logger, output_dir, tb_log_dir, recorder=recorder, fp16=fp16, device=device)
            if debug and i == 4:
                break
    recorder.computeAvgLosses()
    return recorder


def val_helper(i, config, args, master, ret, model, dataset_name, val_loader, recorder, logger, output_dir,
               tb_log_dir, device=None):
    end = time.time()
    if config.TEST.FLIP_TEST:
        raise NotImplementedError('TEST.FLIP_TEST references an undefined dataset in the reference '
                                  '(function.py:692) and is off in every hot-path yaml')
    imgs, loss_dict = _forward_and_losses(config, ret, model, recorder, device)
    if master and i % config.PRINT_FREQ == 0:
        batch_time = time.time() - end
        recorder.computeAvgLosses()
        head = 'Dataset: {0} Test: [{1}/{2}]\t'.format(dataset_name, i, len(val_loader))
        logger.info(_message(head, batch_time, imgs.size(0), loss_dict, recorder, always=True))


def validate(config, args, master, val_loader_dict, model, criterion, output_dir, tb_log_dir, writer_dict, logger,
             device=None):
    recorder = AverageMeter(config, criterion)
    writer = writer_dict['writer']
    model.eval()
    for dataset_name, val_loader in val_loader_dict.items():
        logger.info('Validating on {} dataset [Batch size: {}]\n'.format(dataset_name, val_loader.batch_size))
        if dataset_name not in GENERIC_DATASETS:
            raise NotImplementedError('dataset branch {} is outside the HRNet 2-D hot path'.format(dataset_name))
        with torch.no_grad():
            for i, ret in enumerate(val_loader):
                if getattr(getattr(val_loader, 'dataset', None), 'exception', False):
                    continue
                val_helper(i, config, args, master, ret, model, dataset_name, val_loader, recorder, logger,
                           output_dir, tb_log_dir, device=device)
                if debug and i == 4:
                    break
        recorder.computeAvgLosses()
        steps = writer_dict['valid_global_steps']
        if master and writer is not None:
            writer.add_scalar('val_loss/total_loss', recorder.avg_total_loss, steps)
            if config.LOSS.WITH_HEATMAP_LOSS:
                writer.add_scalar('val_loss/heatmap_loss', recorder.avg_heatmap_loss, steps)
            if config.LOSS.WITH_POSE2D_LOSS:
                writer.add_scalar('val_loss/pose2d_loss', recorder.avg_pose2d_loss, steps)
        writer_dict['valid_global_steps'] = steps + 1
    return recorder
