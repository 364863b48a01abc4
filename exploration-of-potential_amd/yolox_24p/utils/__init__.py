"""The slice of the reference's ``utils`` package the 24p path uses: ``bboxes_iou`` (utils/boxes.py:166-243),
``postprocess`` (utils/boxes.py:29-99),
``save_checkpoint`` / ``load_ckpt`` (utils/checkpoint.py:11-43) and the ``yoloxwarmcos`` schedule
(utils/lr_scheduler.py:121-148)."""
import math
import os
import shutil

import _path  # noqa: F401
import torch
from ep24.loss import bboxes_iou  # noqa: F401
from ep24.infer import postprocess  # noqa: F401      (utils/boxes.py:29-99)


def save_checkpoint(state, is_best, save_dir, model_name=""):
    if not os.path.exists(save_dir):
        os.makedirs(save_dir)
    filename = os.path.join(save_dir, model_name + "_ckpt.pth")
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, os.path.join(save_dir, "best_ckpt.pth"))


def load_ckpt(model, ckpt):
    """Shape-tolerant load: keys that are missing or whose shapes differ are skipped (with a note)."""
    own = model.state_dict()
    keep = {}
    for k, v in own.items():
        if k not in ckpt:
            print("{} is not in the ckpt. Please double check and see if this is desired.".format(k))
            continue
        if v.shape != ckpt[k].shape:
            print("Shape of {} in checkpoint is {}, while shape of {} in model is {}.".format(k, ckpt[k].shape, k, v.shape))
            continue
        keep[k] = ckpt[k]
    model.load_state_dict(keep, strict=False)
    return model


class LRScheduler:
    """``yoloxwarmcos``: quadratic warm-up, cosine decay, constant floor for the last no-aug epochs."""

    def __init__(self, name, lr, iters_per_epoch, total_epochs, warmup_epochs=0, warmup_lr_start=0, no_aug_epochs=0,
                 min_lr_ratio=0.05):
        if name != "yoloxwarmcos":
            raise ValueError("Scheduler version {} not supported.".format(name))
        self.lr = lr
        self.total_iters = iters_per_epoch * total_epochs
        self.warmup_iters = iters_per_epoch * warmup_epochs
        self.no_aug_iters = iters_per_epoch * no_aug_epochs
        self.warmup_lr_start = warmup_lr_start
        self.min_lr = lr * min_lr_ratio

    def update_lr(self, iters):
        if iters <= self.warmup_iters:
            return (self.lr - self.warmup_lr_start) * pow(iters / float(max(self.warmup_iters, 1)), 2) + self.warmup_lr_start
        if iters >= self.total_iters - self.no_aug_iters:
            return self.min_lr
        span = self.total_iters - self.warmup_iters - self.no_aug_iters
        return self.min_lr + 0.5 * (self.lr - self.min_lr) * (1.0 + math.cos(math.pi * (iters - self.warmup_iters) / span))
