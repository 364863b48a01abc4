"""Synthetic stand-in for ``COCO24PDataset`` (datasets/coco24p.py): items are ``(image [3,S,S] fp32 0..255, labels
[50,51], img_info, img_id)`` exactly as the trainer unpacks them (train_24p.py:83).  The real dataset (cv2.imread of
hard-coded paths) is outside the hot path; its ``TrainTransform`` / ``preproc`` half is ``ep24.input`` (GPU)
and is re-exported here under the reference's names."""
import _path  # noqa: F401
import torch
from ep24 import synth


class SyntheticDataset(torch.utils.data.Dataset):
    """``raw=True``: items are what a decoder + txt reader hand over BEFORE ``TrainTransform`` - a uint8 HWC image and the label rows
    ``[k,51]`` with coordinates normalised by width / height (datasets/coco24p.py + 2+24_labels_create.py's txt format) - so the
    letterbox and the label scaling run on the GPU (``ep24.input.TrainTransform.batch`` behind the ``DataPrefetcher``): 1.2 MB per
    640 x 640 image over PCIe instead of the reference's 4.9 MB fp32 canvas."""

    def __init__(self, length=64, size=640, num_gt=10, num_classes=80, seed=0, raw=False):
        self.length, self.size, self.num_gt, self.num_classes, self.seed = length, size, num_gt, num_classes, seed
        self.raw = raw
        self._cache = {}

    def __len__(self):
        return self.length

    DISTINCT = 64                         # items beyond this repeat (a long synthetic epoch must not hold 5 MB per item on the host)

    def __getitem__(self, idx):
        ident = idx
        idx = idx % self.DISTINCT
        if idx in self._cache:
            c = self._cache[idx]
            return (c[0], c[1], c[2], ident)
        img = synth.make_images(1, self.size, seed=self.seed * 100003 + idx)[0]
        lab = synth.make_labels(1, self.num_gt, size=self.size, seed=self.seed * 100003 + 7919 + idx, num_classes=self.num_classes)[0]
        hw = (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
        if self.raw:
            img = img.permute(1, 2, 0).to(torch.uint8).contiguous()             # HWC bytes, as cv2.imread returns them
            rows = lab[: self.num_gt].double().clone()
            rows[:, 1::2] /= float(hw[1])                                        # x / width
            rows[:, 2::2] /= float(hw[0])                                        # y / height
            lab = rows.numpy()
        self._cache[idx] = (img, lab, hw, idx)
        return (img, lab, hw, ident)


class ResumableSampler(torch.utils.data.Sampler):
    """The epoch's index order of ``base`` (a SequentialSampler, or a DistributedSampler shard), with the first ``start`` indices
    left out on the NEXT pass only: a run resumed inside an epoch continues with the batch it stopped in front of instead of
    training the epoch's first batches twice and never reaching its tail.  ``len()`` stays the whole epoch's."""

    def __init__(self, base):
        self.base, self.start = base, 0

    def set_epoch(self, epoch):
        if hasattr(self.base, "set_epoch"):
            self.base.set_epoch(epoch)

    def __len__(self):
        return len(self.base)

    def __iter__(self):
        start, self.start = int(self.start), 0
        for i, idx in enumerate(self.base):
            if i >= start:
                yield idx


def raw_collate(items):
    """Batches of the raw source stay lists (images differ in size in a real dataset): ``DataPrefetcher`` hands them to
    ``TrainTransform.batch``."""
    return [it[0] for it in items], [it[1] for it in items], [it[2] for it in items], [it[3] for it in items]


COCO24PDataset = SyntheticDataset        # name the reference's Exp imports (exp/yolox_base.py:76)


from ep24.input import TrainTransform, preproc  # noqa: E402,F401   (datasets/data_augment.py:109-174, on the GPU)
