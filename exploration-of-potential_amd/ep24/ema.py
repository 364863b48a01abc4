"""``ModelEMA`` of the reference (yolox_24p/utils/ema.py:13-60) on the flat device buffers.

The reference walks the EMA model's ``state_dict`` and does ``v *= d; v += (1 - d) * model_v`` tensor by tensor
(~600 small launches per update).  Here the parameters and the BatchNorm running statistics of an ep24 model live in
two flat fp32 buffers (ep24.engine.ParamHome), so an update is two launches of ``ep24_ema_update`` - or none at all
when the captured training step fuses it into the SGD kernel (``TrainStep(ema=...)``).  The arithmetic is the
reference's: ``d`` and ``1 - d`` are computed in double precision on the host and rounded to fp32, the two products
and the sum are each rounded to fp32.  Integer entries (``num_batches_tracked``) keep the values of the copy, as in
the reference (only floating-point entries are averaged, ema.py:58).
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr, stream_ptr

__all__ = ["ModelEMA", "is_parallel"]


def is_parallel(model):
    """True for the torch wrappers that keep the real model in ``.module`` (ema.py:13-19)."""
    return isinstance(model, (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel))


def _bare(model):
    return model.module if is_parallel(model) else model


def _copy_model(model):
    """deepcopy without the launch plans and flat buffers hanging off the model (the copy builds its own on first use)."""
    held = {k: model.__dict__.pop(k) for k in ("_ep24_home", "_engines") if k in model.__dict__}
    try:
        twin = deepcopy(model)
    finally:
        model.__dict__.update(held)
    if "_engines" in held:
        twin._engines = {}
    return twin


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = _copy_model(_bare(model)).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))          # ramp that helps the early epochs (ema.py:41)
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def next_decay(self):
        """Advance the update counter and return ``(d, 1 - d)`` of this update as python floats."""
        self.updates += 1
        d = self.decay(self.updates)
        return d, 1.0 - d

    def homes(self, model):
        from .engine import param_home
        src, dst = param_home(_bare(model)), param_home(self.ema)
        if (src.numel, src.bnumel) != (dst.numel, dst.bnumel):
            raise _lib.Ep24Error("ep24: the EMA model's parameter layout differs from the trained model's")
        return src, dst

    def update(self, model):
        _lib.require_gpu()
        with torch.no_grad():
            d, omd = self.next_decay()
            src, dst = self.homes(model)
            s = stream_ptr()
            call("ema_update", ptr(dst.flat), ptr(src.flat), src.numel, d, omd, None, s)
            if src.bnumel:
                call("ema_update", ptr(dst.bflat), ptr(src.bflat), src.bnumel, d, omd, None, s)
