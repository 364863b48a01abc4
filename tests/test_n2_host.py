"""SURVEY 8f N2 on the CPU: the learning-rate schedules (host logic) and the EMA / L1 oracles against values produced by
the reference itself (tests/golden/g10_lr, g11_ema_*, g12_loss_l1; generator make_golden.py gen_n2)."""
import numpy as np
import pytest
import torch

from conftest import t
from ep24 import synth
from ep24.schedule import LRScheduler

LR_CASES = [          # (name, kwargs, lr, iters_per_epoch, epochs) - the list make_golden.py sampled
    ("cos", dict(), 0.01, 37, 6),
    ("warmcos", dict(warmup_epochs=2), 0.02, 25, 8),
    ("warmcos", dict(warmup_epochs=1, warmup_lr_start=1e-4), 0.02, 25, 8),
    ("yoloxwarmcos", dict(warmup_epochs=5, warmup_lr_start=0, no_aug_epochs=100, min_lr_ratio=0.05), 0.01 / 64.0 * 20, 50, 300),
    ("yoloxwarmcos", dict(warmup_epochs=1, no_aug_epochs=2), 0.005, 40, 10),
    ("yoloxsemiwarmcos", dict(warmup_epochs=1, no_aug_epochs=2, semi_epoch=5, iters_per_epoch_semi=17), 0.01, 30, 12),
    ("multistep", dict(milestones=[3, 7]), 0.1, 20, 10),
    ("multistep", dict(milestones=[2, 4, 6], gamma=0.5), 0.1, 20, 8),
]


def test_lr_schedules_equal_the_reference_bit_for_bit(golden):
    z = golden("g10_lr")
    assert int(z["n_cases"]) == len(LR_CASES)
    for i, (name, kw, lr, ipe, epochs) in enumerate(LR_CASES):
        sch = LRScheduler(name, lr, ipe, epochs, **kw)
        got = np.array([sch.update_lr(int(it)) for it in z["c%d_iters" % i]], dtype=np.float64)
        assert np.array_equal(got, z["c%d_lr" % i]), (name, kw)


def test_lr_scheduler_interface():
    with pytest.raises(ValueError, match="Scheduler version step not supported."):
        LRScheduler("step", 0.1, 10, 10)
    sch = LRScheduler("yoloxwarmcos", 0.01, 10, 10, warmup_epochs=1, no_aug_epochs=1, min_lr_ratio=0.05)
    assert sch.total_iters == 100 and sch.min_lr_ratio == 0.05           # options become attributes, lr_scheduler.py:29
    assert sch.update_lr(0) == 0.0 and sch.update_lr(10) == 0.01 and sch.update_lr(95) == 0.01 * 0.05
    assert sch.lr_func(50) == sch.update_lr(50)
    with pytest.raises(AttributeError):                                   # warmcos without warmup_epochs, as the reference
        LRScheduler("warmcos", 0.1, 10, 10)
    # the experiment file builds it the reference's way (exp/yolox_base.py:155-167)
    import os
    import sys
    from conftest import PKG
    sys.path.insert(0, os.path.join(PKG, "yolox_24p"))
    try:
        from exp import Exp
        exp = Exp()
        s2 = exp.get_lr_scheduler(0.003, 100)
        assert s2.update_lr(100 * exp.warmup_epochs) == 0.003 and s2.update_lr(100 * exp.max_epoch) == 0.003 * exp.min_lr_ratio
    finally:
        sys.path.remove(os.path.join(PKG, "yolox_24p"))


@pytest.mark.parametrize("start", [0, 1500])
def test_ema_oracle_vs_reference(golden, start):
    from oracle import ema as oema
    z = golden("g11_ema_%d" % start)
    state = {"w": t(z["w0"]).clone(), "stat": t(z["stat0"]).clone(), "count": torch.tensor(7)}
    updates = int(z["start"])
    for step in range(4):
        model = {"w": t(z["w_model%d" % step]), "stat": t(z["stat_model%d" % step]), "count": torch.tensor(8 + step)}
        updates = oema.update(state, model, updates, float(z["decay"]))
        assert torch.equal(state["w"], t(z["w_ema%d" % step]))
        assert torch.equal(state["stat"], t(z["stat_ema%d" % step]))
    assert updates == int(z["updates"]) and int(state["count"]) == int(z["count_ema"]) == 7


def l1_case(z):
    B = int(z["B"])
    labels = synth.make_labels(B, [int(c) for c in z["counts"]], seed=int(z["label_seed"]))
    raw = synth.make_raw_head(B, seed=int(z["head_seed"]))
    origin, a0 = [], 0
    for s in synth.STRIDES:
        n = (640 // s) ** 2
        origin.append(raw[:, a0:a0 + n, :26].clone())
        a0 += n
    return labels, raw, origin


def test_l1_oracle_vs_reference(golden):
    from oracle.loss import LossOracle
    z = golden("g12_loss_l1")
    labels, raw, origin = l1_case(z)
    outputs = synth.decode_head(raw).requires_grad_(True)
    origin = [o.requires_grad_(True) for o in origin]
    tup5 = list(synth.outputs_train_tuple(outputs))
    tup5[4] = origin
    tup = LossOracle(80, use_l1=True)(tuple(tup5), labels)
    tup[0].backward()
    for k, v in (("loss", tup[0]), ("loss_iou_w", tup[1]), ("loss_obj", tup[2]), ("loss_cls", tup[3]), ("loss_l1", tup[4])):
        torch.testing.assert_close(v.detach(), t(z[k]), rtol=1e-5, atol=1e-7)
    g = torch.cat([o.grad for o in origin], 1).reshape(-1, 26)
    rows = t(z["d_origin_rows"])
    assert torch.equal(g.abs().sum(-1).nonzero().reshape(-1), rows)
    assert torch.equal(g[rows], t(z["d_origin_vals"]))
    assert abs(float(outputs.grad.double().abs().sum()) - float(z["grad_abs_sum"])) < 1e-5 * float(z["grad_abs_sum"])
