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
