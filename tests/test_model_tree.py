"""CPU-only: the product's parameter tree keeps the reference's state-dict names, shapes and size."""
import torch

from conftest import t
from ep24 import nn as enn


def test_l_keys_and_param_count(golden):
    z = golden("g7_model_l_keys")
    m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
    sd = m.state_dict()
    assert {k: str(tuple(v.shape)) for k, v in sd.items()} == {str(k): str(s) for k, s in zip(z["keys"], z["shapes"])}
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"]) == 54225857
    m.head.initialize_biases(1e-2)
    assert abs(float(m.head.cls_preds[0].bias[0]) + 4.59512) < 1e-4 and float(m.head.reg_preds[0].bias.abs().max()) < 1.0


def test_bn_hyperparameters_and_no_cpu_path():
    import pytest
    from ep24 import _lib
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125))
    bns = [x for x in m.modules() if isinstance(x, torch.nn.BatchNorm2d)]
    assert bns and all(b.eps == 1e-3 and b.momentum == 0.03 for b in bns)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.Ep24Error):
            m(torch.zeros(1, 3, 64, 64), train=True)


def test_exec_order_covers_every_parameter():
    from ep24.engine import exec_order
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125))
    seen = set()
    for mod in exec_order(m):
        if isinstance(mod, enn.BaseConv):
            seen |= {mod.conv.weight, mod.bn.weight, mod.bn.bias}
        elif isinstance(mod, tuple) and mod[0] in ("csp_merged", "pair_merged"):   # two units over one input run as one
            for c in ((mod[1].conv1, mod[1].conv2) if mod[0] == "csp_merged" else mod[1:]):
                seen |= {c.conv.weight, c.bn.weight, c.bn.bias}
        elif isinstance(mod, enn.YOLOXHead):
            for k in range(3):
                for c in (mod.reg_preds[k], mod.obj_preds[k], mod.cls_preds[k]):
                    seen |= {c.weight, c.bias}
    assert seen == set(m.parameters())


def test_grouped_weight_gradient_split_plan():
    """Round 5: pixel splits of a grouped weight-gradient launch (ep24.engine.wgrad_group_splits) - every workgroup keeps its minimum
    number of 64-pixel steps, the (tile, split) grid fits the resident slots unless one split each already exceeds them, and the cap is
    the largest that does."""
    from ep24.engine import wgrad_group_splits
    # fifteen 1x1 256 -> 256 layers of the 40 x 40 level at B = 20: 16 tiles of 64 x 64, 500 steps, 1024 slots
    sp, wg, slots = wgrad_group_splits([16] * 15, [500] * 15, 3, 120)
    assert slots == 1024 and sp == [4] * 15 and wg == 960
    # seven 3x3 256 -> 256 layers: 36 tiles of 128 x 128 each -> two splits fill 504 of 512 slots; an eighth pushes it to one split
    assert wgrad_group_splits([36] * 7, [500] * 7, 0, 120)[:2] == ([2] * 7, 504)
    assert wgrad_group_splits([36] * 8, [500] * 8, 0, 120)[:2] == ([1] * 8, 288)
    # mixed shapes share the cap but never drop below the minimum steps: a 125-step problem is never split at 120
    sp, wg, _ = wgrad_group_splits([9, 36, 4], [2000, 500, 125], 0, 120)
    assert sp[2] == 1 and all(st // q >= 120 for st, q in zip([2000, 500, 125], sp)) and wg <= 512
    assert sp == [16, 4, 1] and wg == 292                            # the common cap K = 16 is clipped per problem by its own step count
    # more tiles than slots: one split each, several rounds
    assert wgrad_group_splits([300, 300], [500, 500], 0, 120) == ([1, 1], 600, 512)
