"""CPU side of the depthwise variants (DWConv, /root/reference/yolox_24p/models/network_blocks.py:57-76): the parameter tree that
``depthwise=True`` builds carries the reference's state-dict names and shapes (G19, written from the reference's own constructors),
the plan's execution order reaches every parameter, and what the path does not take is refused with a message."""
import pytest
import torch

from ep24 import nn as enn
from ep24.engine import exec_order, head_is_merged
from ep24.options import DEFAULT


def test_depthwise_state_dict_is_the_references(golden):
    z = golden("g19_model_dw_tiny")
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125, depthwise=True), enn.YOLOXHead(80, 0.125, depthwise=True))
    ref = {str(k): str(s) for k, s in zip(z["keys"], z["shapes"])}
    assert {k: str(tuple(v.shape)) for k, v in m.state_dict().items()} == ref
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])
    # the dense model (the BASELINE configurations) is untouched by the switch
    d = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125))
    assert not any("dconv" in k for k in d.state_dict()) and any("dconv" in k for k in m.state_dict())


def test_execution_order_reaches_every_parameter_of_the_depthwise_tree():
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.25, depthwise=True), enn.YOLOXHead(80, 0.25, depthwise=True))
    assert not head_is_merged(m.head, DEFAULT)                  # depthwise first convs of the two branches do not merge
    seen = set()
    for item in exec_order(m, DEFAULT):
        if isinstance(item, tuple) and item[0] == "csp_merged":
            mods = [item[1].conv1, item[1].conv2]
        elif isinstance(item, tuple):
            raise AssertionError(item[0])
        elif isinstance(item, enn.YOLOXHead):
            mods = list(item.cls_preds) + list(item.reg_preds) + list(item.obj_preds)
        else:
            assert isinstance(item, enn.BaseConv), type(item)
            mods = [item]
        for mod in mods:
            for p in mod.parameters():
                assert id(p) not in seen
                seen.add(id(p))
    assert seen == {id(p) for p in m.parameters()}
    # every DWConv contributes its depthwise unit directly in front of its 1x1 unit
    order = [it for it in exec_order(m, DEFAULT) if isinstance(it, enn.BaseConv)]
    for mod in m.modules():
        if isinstance(mod, enn.DWConv):
            i = order.index(mod.dconv)
            assert order[i + 1] is mod.pconv


def test_unsupported_groupings_are_refused():
    with pytest.raises(NotImplementedError, match="depthwise"):
        enn.BaseConv(16, 16, 3, 1, groups=4)                    # grouped but not depthwise: not in the reference
    with pytest.raises(NotImplementedError, match="3x3"):
        enn.BaseConv(16, 16, 1, 1, groups=16)
    with pytest.raises(NotImplementedError, match="multiple of 8"):
        enn.BaseConv(12, 12, 3, 1, groups=12)
    with pytest.raises(NotImplementedError, match="CSPDarknet"):
        enn.YOLOPAFPN(1.0, 1.0, depthwise=True, backbone_type="resnet")
    blk = enn.DWConv(16, 24, 3, 2)
    assert blk.dconv.conv.groups == 16 and tuple(blk.dconv.conv.weight.shape) == (16, 1, 3, 3) and blk.pconv.conv.kernel_size == (1, 1)
    assert isinstance(enn.Bottleneck(16, 16, depthwise=True).conv2, enn.DWConv) and isinstance(enn.Bottleneck(16, 16).conv2, enn.BaseConv)
