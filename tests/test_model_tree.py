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
