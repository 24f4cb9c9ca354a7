"""Portable synthetic data: counter-based PRNG, weight fill, RHD-shaped batches.

Everything here is integer -> float arithmetic on numpy uint64 (splitmix64), so
the container that generates the golden fixtures and the GPU box regenerate
bit-identical weights and inputs without shipping them.

The batch layout mirrors the sample dict of the reference loader
(lib/dataset/RHDDatasetKeypoints.py:126-134): 'imgs' (B,3,256,256) ImageNet-
normalised (lib/dataset/transforms/build.py:84-85), 'heatmaps' (B,21,64,64)
un-normalised Gaussian sigma=2 in a 6*sigma+3 window centred at int(coord)
(lib/dataset/target_generators/target_generators.py:15-53), 'pose2d' (B,21,2),
'visibility' (B,21,1).
"""
import zlib

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def key_seed(name, salt=0):
    """64-bit stream id for a tensor name (crc32 of the utf-8 name, salted)."""
    return (np.uint64(zlib.crc32(name.encode('utf-8'))) << np.uint64(20)) ^ np.uint64(salt)


def uniform01(seed, n):
    """n floats in [0,1) with 24 random bits each (exact in fp32)."""
    with np.errstate(over='ignore'):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = _mix(np.uint64(seed) * _M2 + idx * _GOLD)
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def fill_for_key(name, shape, salt=0):
    """Deterministic fp32 tensor for a state_dict key.

    conv / linear weights: uniform with the std of He-normal (fan_in);
    biases: small uniform; BN gamma in [0.5,1.5), beta in [-0.25,0.25);
    running_mean in [-0.25,0.25), running_var in [0.5,1.5).
    """
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(key_seed(name, salt), n)
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'weight' and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        a = np.float32(np.sqrt(3.0) * np.sqrt(2.0 / fan_in))
        v = (u - np.float32(0.5)) * np.float32(2.0) * a
    elif leaf == 'weight' or leaf == 'running_var':
        v = np.float32(0.5) + u
    elif leaf in ('bias', 'running_mean'):
        v = (u - np.float32(0.5)) * np.float32(0.5)
    else:
        v = u
    return v.astype(np.float32).reshape(shape)


def fill_state_dict(template, salt=0):
    """template: {key: shape-like or tensor}; returns {key: np.ndarray}."""
    out = {}
    for k, v in template.items():
        shape = tuple(v.shape) if hasattr(v, 'shape') else tuple(v)
        out[k] = fill_for_key(k, shape, salt)
    return out


IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def gaussian_heatmaps(pose2d, visibility, hm_h=64, hm_w=64, sigma=2):
    """Un-normalised Gaussians, peak 1 at int(coord), window 6*sigma+3."""
    b, k = pose2d.shape[:2]
    size = 6 * sigma + 3
    g1 = np.arange(size, dtype=np.float32)
    c = 3 * sigma + 1
    g = np.exp(-((g1[None, :] - c) ** 2 + (g1[:, None] - c) ** 2) / (2.0 * sigma * sigma)).astype(np.float32)
    hms = np.zeros((b, k, hm_h, hm_w), dtype=np.float32)
    for i in range(b):
        for j in range(k):
            if not visibility[i, j, 0]:
                continue
            x, y = int(pose2d[i, j, 0]), int(pose2d[i, j, 1])
            if x < 0 or y < 0 or x >= hm_w or y >= hm_h:
                continue
            ulx, uly = x - 3 * sigma - 1, y - 3 * sigma - 1
            brx, bry = x + 3 * sigma + 2, y + 3 * sigma + 2
            cx0, cx1 = max(0, -ulx), min(brx, hm_w) - ulx
            cy0, cy1 = max(0, -uly), min(bry, hm_h) - uly
            ax0, ax1 = max(0, ulx), min(brx, hm_w)
            ay0, ay1 = max(0, uly), min(bry, hm_h)
            hms[i, j, ay0:ay1, ax0:ax1] = np.maximum(hms[i, j, ay0:ay1, ax0:ax1], g[cy0:cy1, cx0:cx1])
    return hms


def rhd_batch(batch, seed=1234, img_h=256, img_w=256, num_joints=21, sigma=2):
    """One synthetic RHD-shaped batch (numpy, fp32 NCHW images)."""
    hm_h, hm_w = img_h // 4, img_w // 4
    u8 = np.floor(uniform01(key_seed('imgs', seed), batch * 3 * img_h * img_w) * 256.0)
    imgs = (u8.reshape(batch, 3, img_h, img_w) / np.float32(255.0)
            - IMAGENET_MEAN[None, :, None, None]) / IMAGENET_STD[None, :, None, None]
    up = uniform01(key_seed('pose2d', seed), batch * num_joints * 2).reshape(batch, num_joints, 2)
    lo = np.float32(4.0)
    pose2d = lo + up * np.array([hm_w - 8.0, hm_h - 8.0], dtype=np.float32)
    vis = (uniform01(key_seed('visibility', seed), batch * num_joints) < 0.9).reshape(batch, num_joints, 1)
    hms = gaussian_heatmaps(pose2d, vis, hm_h, hm_w, sigma)
    return {
        'imgs': imgs.astype(np.float32),
        'heatmaps': hms,
        'pose2d': pose2d.astype(np.float32),
        'visibility': vis,
    }
