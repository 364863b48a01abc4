"""The slice of the reference's ``utils`` package the 24p path uses: ``bboxes_iou`` / ``circle_inter`` (utils/boxes.py:102-243),
``postprocess`` (utils/boxes.py:29-99),
``save_checkpoint`` / ``load_ckpt`` (utils/checkpoint.py:11-43), the learning-rate schedules
(utils/lr_scheduler.py:9-205) and ``ModelEMA`` (utils/ema.py:22-60)."""
import os
import shutil

import _path  # noqa: F401
import torch
from ep24.loss import bboxes_iou, circle_inter  # noqa: F401   (utils/boxes.py:102-243)
from . import boxes  # noqa: F401
from ep24.infer import postprocess  # noqa: F401      (utils/boxes.py:29-99)
from ep24.schedule import LRScheduler  # noqa: F401   (utils/lr_scheduler.py:9-92)
from ep24.ema import ModelEMA, is_parallel  # noqa: F401   (utils/ema.py:13-60)


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
