"""Device-side data step: the reference builds Gaussian targets and normalises images
on the host, one sample at a time (lib/dataset/target_generators/target_generators.py:14-53,
lib/dataset/transforms/build.py:84-85); here a whole batch is one HIP launch each, so the loader
only has to deliver key-point coordinates and u8 crops."""
import ctypes

import torch

from hipnet import _capi as C

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class HeatmapGenerator(object):
    """same constructor as the reference's (output_res, num_joints, sigma=-1 -> output_res/64); called on a
    batch of joints (B,K,3) [x, y, visible] or (K,3), returns float32 maps on the device."""

    def __init__(self, output_res, num_joints, sigma=-1):
        self.output_res, self.num_joints = output_res, num_joints
        self.sigma = float(output_res) / 64 if sigma < 0 else float(sigma)

    def __call__(self, joints, device=None):
        j = torch.as_tensor(joints, dtype=torch.float32)
        single = j.dim() == 2
        if single:
            j = j[None]
        assert j.shape[1] == self.num_joints and j.shape[2] >= 2
        j = j.to(device or ('cuda' if not j.is_cuda else j.device))
        vis = j[..., 2].contiguous() if j.shape[2] > 2 else None
        hms = gaussian_targets(j[..., :2], vis, self.output_res, self.output_res, self.sigma)
        return hms[0] if single else hms


def gaussian_targets(pose2d, visibility, height, width, sigma):
    """pose2d (B,K,2) heat-map pixel coordinates, visibility (B,K[,1]) or None -> (B,K,H,W) f32"""
    if not pose2d.is_cuda:
        raise RuntimeError('gaussian_targets: tensors must be on the HIP device (no CPU path)')
    b, k = pose2d.shape[:2]
    p = pose2d.contiguous().float()
    v = None if visibility is None else visibility.reshape(b, k).contiguous().float()
    out = torch.empty((b, k, height, width), dtype=torch.float32, device=p.device)
    C.call('hrnet_gaussian_targets', p.data_ptr(), C.ptr(v), out.data_ptr(), b * k, height, width, float(sigma),
           C.stream_ptr())
    return out


def normalize_u8(images_nhwc, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """(B,H,W,3) uint8 RGB crops on the device -> (B,3,H,W) f32 normalised (ToTensor + Normalize)"""
    if not images_nhwc.is_cuda or images_nhwc.dtype != torch.uint8 or images_nhwc.shape[-1] != 3:
        raise RuntimeError('normalize_u8 expects a (B,H,W,3) uint8 tensor on the HIP device')
    x = images_nhwc.contiguous()
    b, h, w, _ = x.shape
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=x.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    C.call('hrnet_normalize_u8', x.data_ptr(), out.data_ptr(), b, h, w, m, s, C.stream_ptr())
    return out
