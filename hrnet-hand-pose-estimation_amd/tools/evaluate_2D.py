"""2-D evaluation entry point with the reference's flags (tools/evaluate_2D.py:29-59):

    python tools/evaluate_2D.py --cfg <yaml> --model_path <state_dict.pth.tar> --batch_size 32 --gpu 0

cfg -> get_pose_net(is_train=False) -> strict load ('module.' prefix stripped) -> eval loop:
model(imgs) + get_final_preds -> per-joint end-point error (pixels of the input crop) weighted by
visibility, PCK for thresholds 1..49 px, fps after 20 warm-up iterations; writes
mse2d_each_joint.txt and PCK2d.txt (tools/evaluate_2D.py:172-294). Data: synthetic RHD-shaped loader.
"""
import argparse
import os
import time

import _init_paths  # noqa: F401
import numpy as np
import torch

from config import cfg, update_config
from dataset.build import make_dataloader
from models import pose_hrnet, pose_hrnet_softmax  # noqa: F401
from utils.heatmap_decoding import get_final_preds


def parse_args():
    p = argparse.ArgumentParser(description='Please specify the mode [training/assessment/predicting]')
    p.add_argument('--cfg', required=True, type=str)
    p.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--gpu', default=-1, type=int)
    p.add_argument('--world-size', default=1, type=int)
    p.add_argument('--is_vis', default=0, type=int)
    p.add_argument('--batch_size', default=32, type=int)
    p.add_argument('--model_path', default='', type=str)
    p.add_argument('--num_batches', default=24, type=int, help='synthetic loader length')
    return p.parse_args()


def main():
    args = parse_args()
    update_config(cfg, args)
    device = torch.device('cuda', max(args.gpu, 0))
    torch.cuda.set_device(device)
    model = eval(cfg.MODEL.NAME + '.get_pose_net')(cfg, is_train=False)
    if args.model_path:
        sd = torch.load(args.model_path, map_location='cpu')
        sd = sd.get('state_dict', sd)
        model.load_state_dict({k[7:] if k.startswith('module.') else k: v for k, v in sd.items()}, strict=True)
    model = model.to(device).eval()
    c = cfg.clone()
    c.defrost()
    c.TEST.IMAGES_PER_GPU = args.batch_size
    loader = list(make_dataloader(c, False, num_batches=args.num_batches).values())[0]
    K = cfg.MODEL.NUM_JOINTS
    scale = cfg.MODEL.IMAGE_SIZE[0] / cfg.MODEL.HEATMAP_SIZE[0]
    thresholds = np.arange(1, 50)
    # accumulators and file formats of the reference (tools/evaluate_2D.py:166-169,270-294): per-joint error
    # sums, PCK counted with a strict `<` over all visible joints, PCK2d.txt = two rows (thresholds, PCK)
    err_sum, vis_sum, pck_hits = np.zeros(K), np.zeros(K), np.zeros(len(thresholds))
    timed, t_total = 0, 0.0
    with torch.no_grad():
        for i, ret in enumerate(loader):
            imgs = ret['imgs'].to(device)
            torch.cuda.synchronize()
            t0 = time.time()
            hm = model(imgs)[0]      # (heatmaps, inter_feat[, temperature])
            pred = get_final_preds(hm, cfg.MODEL.HEATMAP_SOFTMAX)
            torch.cuda.synchronize()
            if i >= 20 or i >= len(loader) // 2:
                t_total += time.time() - t0
                timed += imgs.shape[0]
            pred = pred.cpu().numpy() * scale
            gt = ret['pose2d'].numpy() * scale
            vis = ret['visibility'].numpy().reshape(imgs.shape[0], K).astype(np.float64)
            epe = np.linalg.norm(pred - gt, axis=2)
            err_sum += (epe * vis).sum(0)
            vis_sum += vis.sum(0)
            pck_hits += ((epe[None] * vis[None] < thresholds[:, None, None]) * vis[None]).sum((1, 2))
    out_dir = os.path.join(cfg.OUTPUT_DIR or 'output', 'eval2D_results_' + cfg.EXP_NAME)
    os.makedirs(out_dir, exist_ok=True)
    mse_each = err_sum / np.maximum(vis_sum, 1)
    pck = pck_hits / max(vis_sum.sum(), 1)
    np.savetxt(os.path.join(out_dir, 'mse2d_each_joint.txt'), mse_each, fmt='%.4f')
    np.savetxt(os.path.join(out_dir, 'PCK2d.txt'), np.stack((thresholds, pck)))
    print('fps: {:.1f}'.format(timed / max(t_total, 1e-9)))
    print('mean EPE {:.3f} px  PCK@20px {:.4f}  AUC(1-49px) {:.4f}'.format(mse_each.mean(), pck[19], pck.mean()))


if __name__ == '__main__':
    main()
