"""ORACLE — test infrastructure only (never imported by the product path).

Restatement of the accumulation loop of the reference's 2-D evaluator, /root/reference/tools/evaluate_2D.py:
    :165-169   th2d_lst = 1..49, PCK2d_lst / mse2d_lst / visibility_lst zeroed
    :235-239   RHD: pred * crop_size / hm_size + corner (the same affine on the ground truth)
    :241-245   other datasets: x * orig_width / hm_size, y * orig_height / hm_size
    :268-274   mse_each_joint = ||pred - gt||_2 * visibility;  mse2d_lst += its column sums;
               visibility_lst += visibility column sums;  PCK2d_lst[t] += sum((mse_each_joint < th[t]) * visibility)
    :282-283   mse2d_lst /= visibility_lst;  PCK2d_lst /= visibility_lst.sum()
    :293-294   np.savetxt(mse_file, mse2d_lst, fmt='%.4f');  np.savetxt(pck_file, np.stack((th2d_lst, PCK2d_lst)))

Parity pin: the loop is inline in the reference's main() (needs cv2 / kornia / datasets to run), so it cannot be
executed here; it is restated line by line in plain python loops and pinned by the property the reference's
committed result files hold (tools/eval2D_results_*/PCK2d.txt: first row 1..49, second row non-decreasing in
[0,1]) in tests/test_eval2d_cpu.py.
"""
import numpy as np


def evaluate_batches(batches, n_joints, hm_size):
    """batches: iterable of dicts {pred (B,K,2) heat-map px, gt (B,K,2), visibility (B,K,1) 0/1 and either
    crop_size (B,) + corner (B,2) [RHD] or orig_size (w, h)}. Returns (mse2d_each_joint (K,), PCK (2,49))."""
    th = np.array([i for i in range(1, 50)])
    pck = np.zeros((len(th),))
    mse = np.zeros((n_joints,))
    vis_sum = np.zeros((n_joints,))
    for b in batches:
        pred = np.array(b['pred'], dtype=np.float64)
        gt = np.array(b['gt'], dtype=np.float64)
        vis = np.array(b['visibility'], dtype=np.float64)[:, :, 0]
        if 'crop_size' in b:
            cs = np.asarray(b['crop_size'], dtype=np.float64).reshape(-1, 1, 1)
            corner = np.asarray(b['corner'], dtype=np.float64)[:, None, :]
            pred = pred * cs / hm_size + corner
            gt = gt * cs / hm_size + corner
        else:
            ow, oh = b['orig_size']
            pred = pred * np.array([ow / hm_size, oh / hm_size])
            gt = gt * np.array([ow / hm_size, oh / hm_size])
        B = pred.shape[0]
        each = np.zeros((B, n_joints))
        for i in range(B):
            for k in range(n_joints):
                d = pred[i, k] - gt[i, k]
                each[i, k] = np.sqrt(d[0] * d[0] + d[1] * d[1]) * vis[i, k]
        mse += each.sum(axis=0)
        vis_sum += vis.sum(axis=0)
        for t in range(len(th)):
            pck[t] += np.sum((each < th[t]) * vis)
    mse = mse / vis_sum
    pck = pck / vis_sum.sum()
    return mse, np.stack((th, pck))
