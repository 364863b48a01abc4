"""Synthetic stand-in for ``COCO24PDataset`` (datasets/coco24p.py): items are ``(image [3,S,S] fp32 0..255, labels
[50,51], img_info, img_id)`` exactly as the trainer unpacks them (train_24p.py:83).  The real dataset (cv2.imread of
hard-coded paths) is outside the hot path; its ``TrainTransform`` / ``preproc`` half is ``ep24.input`` (GPU)
and is re-exported here under the reference's names."""
import _path  # noqa: F401
import torch
from ep24 import synth


class SyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, length=64, size=640, num_gt=10, num_classes=80, seed=0):
        self.length, self.size, self.num_gt, self.num_classes, self.seed = length, size, num_gt, num_classes, seed
        self._cache = {}

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        if idx in self._cache:
            return self._cache[idx]
        img = synth.make_images(1, self.size, seed=self.seed * 100003 + idx)[0]
        lab = synth.make_labels(1, self.num_gt, size=self.size, seed=self.seed * 100003 + 7919 + idx, num_classes=self.num_classes)[0]
        hw = (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
        self._cache[idx] = (img, lab, hw, idx)
        return self._cache[idx]


COCO24PDataset = SyntheticDataset        # name the reference's Exp imports (exp/yolox_base.py:76)


from ep24.input import TrainTransform, preproc  # noqa: E402,F401   (datasets/data_augment.py:109-174, on the GPU)
