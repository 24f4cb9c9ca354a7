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
from core.evaluate2d import Eval2DAccumulator, load_checkpoint_state
from dataset.build import make_dataloader
from models import pose_hrnet, pose_hrnet_PoseAggr, pose_hrnet_softmax  # noqa: F401
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
        load_checkpoint_state(model, args.model_path)
    model = model.to(device).eval()
    c = cfg.clone()
    c.defrost()
    c.TEST.IMAGES_PER_GPU = args.batch_size
    loader = list(make_dataloader(c, False, num_batches=args.num_batches).values())[0]
    K = cfg.MODEL.NUM_JOINTS
    # accumulators and file formats of the reference (tools/evaluate_2D.py:166-169,270-294): per-joint error
    # sums, PCK counted with a strict `<` over all visible joints, PCK2d.txt = two rows (thresholds, PCK);
    # coordinates are scaled from heat-map pixels to the input crop (:241-245 with orig size = crop size)
    acc = Eval2DAccumulator(K, cfg.MODEL.HEATMAP_SIZE[0])
    crop = (cfg.MODEL.IMAGE_SIZE[0], cfg.MODEL.IMAGE_SIZE[1])
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
            acc.add(pred.cpu().numpy(), ret['pose2d'].numpy(), ret['visibility'].numpy(), orig_size=crop)
    out_dir = os.path.join(cfg.OUTPUT_DIR or 'output', 'eval2D_results_' + cfg.EXP_NAME)
    mse_each, pck = acc.save(out_dir)
    print('fps: {:.1f}'.format(timed / max(t_total, 1e-9)))
    print('mean EPE {:.3f} px  PCK@20px {:.4f}  AUC(1-49px) {:.4f}'.format(np.nanmean(mse_each), pck[1, 19], pck[1].mean()))


if __name__ == '__main__':
    main()
