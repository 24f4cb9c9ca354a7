"""Loader on the input side of the hot path. No dataset ships with the repository, so the loader is
synthetic and RHD-shaped: it yields the sample dict of the reference's RHD key-point dataset
(lib/dataset/RHDDatasetKeypoints.py:126-134; batch = IMAGES_PER_GPU * number of GPUs of this
process, lib/dataset/build.py:66-97) from the portable generator in hipnet/synth.py."""
import torch

from hipnet import synth


class SyntheticRHD(torch.utils.data.Dataset):
    exception = False

    def __init__(self, cfg, length, seed):
        self.cfg, self.length, self.seed = cfg, length, seed
        self.img_w, self.img_h = cfg.MODEL.IMAGE_SIZE
        self._cache = {}

    def __len__(self):
        return self.length

    def batch(self, index, batch_size):
        key = (index, batch_size)
        if key not in self._cache:
            b = synth.rhd_batch(batch_size, seed=self.seed + index, img_h=self.img_h, img_w=self.img_w,
                                num_joints=self.cfg.MODEL.NUM_JOINTS, sigma=self.cfg.MODEL.SIGMA)
            self._cache = {key: {k: torch.from_numpy(v) for k, v in b.items()}}
        return self._cache[key]


class SyntheticLoader(object):
    """iterable of batches with the attributes the loops use (`batch_size`, `dataset`, `len`)"""

    def __init__(self, dataset, batch_size, num_batches, rank=0, world=1):
        self.dataset, self.batch_size, self.num_batches = dataset, batch_size, num_batches
        self.rank, self.world = rank, world
        self.sampler = self

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        for i in range(self.num_batches):
            yield self.dataset.batch(i * self.world + self.rank, self.batch_size)


def make_dataloader(cfg, is_train=True, distributed=False, num_batches=8, rank=0, world=1):
    per_gpu = cfg.TRAIN.IMAGES_PER_GPU if is_train else cfg.TEST.IMAGES_PER_GPU
    ds = SyntheticRHD(cfg, length=per_gpu * num_batches * world, seed=1234 if is_train else 4321)
    return {'synthetic_kpt': SyntheticLoader(ds, per_gpu, num_batches, rank, world)}
