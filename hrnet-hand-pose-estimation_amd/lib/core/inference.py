"""get_max_preds (reference lib/core/inference.py:18-46): arg-max decode with max values, numpy in /
numpy out like the reference, computed by the HIP arg-max kernel."""
import numpy as np
import torch

from hipnet import _capi as C


def get_max_preds(batch_heatmaps):
    assert isinstance(batch_heatmaps, np.ndarray), 'batch_heatmaps should be numpy.ndarray'
    assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
    b, k, h, w = batch_heatmaps.shape
    hms = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).cuda()
    preds = torch.empty((b, k, 2), dtype=torch.float32, device=hms.device)
    maxvals = torch.empty((b, k), dtype=torch.float32, device=hms.device)
    C.call('hrnet_decode_argmax', hms.data_ptr(), preds.data_ptr(), maxvals.data_ptr(), b * k, h, w, 1, C.stream_ptr())
    return preds.cpu().numpy(), maxvals.cpu().numpy().reshape(b, k, 1)


def transform_preds(coords, center, scale, output_size):
    """Heat-map pixel coordinates -> original image coordinates (reference lib/utils/transforms.py:50-55
    with rot=0, inv=1). The reference builds the map from three point pairs with cv2.getAffineTransform
    (transforms.py:58-90); without rotation that map is the similarity
        p_img = center + (p_hm - output_size / 2) * (200 * scale[0] / output_size[0])
    (the x extent sets the factor for both axes, as src_w / dst_w does there). float64 out, like np.zeros()."""
    coords = np.asarray(coords)
    scale = np.asarray(scale, dtype=np.float64).reshape(-1)
    if scale.size == 1:
        scale = np.repeat(scale, 2)
    f = 200.0 * float(scale[0]) / float(output_size[0])
    out = np.zeros(coords.shape)
    out[:, 0] = float(center[0]) + (coords[:, 0].astype(np.float64) - 0.5 * float(output_size[0])) * f
    out[:, 1] = float(center[1]) + (coords[:, 1].astype(np.float64) - 0.5 * float(output_size[1])) * f
    return out


def get_final_preds(config, batch_heatmaps, center, scale):
    """reference lib/core/inference.py:49-85: arg-max, optional quarter-pixel shift towards the higher
    neighbour (TEST.POST_PROCESS, interior maxima only), then back to image coordinates."""
    coords, maxvals = get_max_preds(batch_heatmaps)
    h, w = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    if config.TEST.POST_PROCESS:
        px = np.floor(coords[..., 0] + 0.5).astype(np.int64)
        py = np.floor(coords[..., 1] + 0.5).astype(np.int64)
        inner = (px > 1) & (px < w - 1) & (py > 1) & (py < h - 1)
        pxc, pyc = np.clip(px, 1, w - 2), np.clip(py, 1, h - 2)
        bi = np.arange(coords.shape[0])[:, None]
        ki = np.arange(coords.shape[1])[None, :]
        dx = batch_heatmaps[bi, ki, pyc, pxc + 1] - batch_heatmaps[bi, ki, pyc, pxc - 1]
        dy = batch_heatmaps[bi, ki, pyc + 1, pxc] - batch_heatmaps[bi, ki, pyc - 1, pxc]
        coords[..., 0] += np.where(inner, np.sign(dx) * 0.25, 0.0).astype(coords.dtype)
        coords[..., 1] += np.where(inner, np.sign(dy) * 0.25, 0.0).astype(coords.dtype)
    preds = coords.copy()
    for i in range(coords.shape[0]):
        preds[i] = transform_preds(coords[i], center[i], scale[i], [w, h])
    return preds, maxvals
