"""Oracle: ModelEMA.update (yolox_24p/utils/ema.py:34-60).  Test infrastructure only.

``decay(x) = decay0 * (1 - exp(-x / 2000))`` with x the 1-based update count, then for every floating-point entry of
the EMA state: ``v *= d`` followed by ``v += (1 - d) * model_v`` - fp32 tensor arithmetic with the python-float
scalars d and 1 - d, i.e. two rounded products and a rounded sum per element.  Pinned by tests/golden/g11_ema.npz,
which was produced by the reference class itself.
"""
import math

import torch


def decay_at(updates, decay=0.9999):
    return decay * (1 - math.exp(-updates / 2000))


def update(ema_state, model_state, updates, decay=0.9999):
    """In place on the tensors of ``ema_state`` (dict name -> tensor); returns the new update count."""
    updates += 1
    d = decay_at(updates, decay)
    with torch.no_grad():
        for k, v in ema_state.items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1.0 - d) * model_state[k].detach()
    return updates
