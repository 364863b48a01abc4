"""BASELINE config 4 on the CPU: the oracle's ResNet-backbone network against the reference (G15), and the parameter
tree of the product's mirror (names, shapes, unused entries)."""
import numpy as np
import torch

from conftest import t
from ep24 import synth
from oracle import model as omodel

GRADS = ("backbone.backbone.conv1.weight", "backbone.backbone.bn1.weight", "backbone.backbone.layer1.0.downsample.0.weight",
         "backbone.backbone.layer1.0.conv2.weight", "backbone.backbone.layer2.0.downsample.0.weight",
         "backbone.backbone.layer2.3.bn3.bias", "backbone.backbone.layer3.5.conv1.weight", "backbone.backbone.layer4.0.conv2.weight",
         "backbone.backbone.layer4.2.bn3.weight", "backbone.lateral_conv0.conv.weight", "head.stems.0.conv.weight")


def sub(g):
    return g if g.numel() <= 40000 else g.reshape(-1)[:: g.numel() // 20000 + 1]


def cotangent(shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(152)) * torch.tensor([0.05] * 26 + [1.0] * 81)


def test_oracle_resnet_network_vs_reference(golden):
    z = golden("g15_resnet")
    net = omodel.Net(0.33, 1.0, 80, backbone_type="resnet")
    assert sorted(net.state_dict().keys()) == [str(k) for k in z["keys"]]
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"])
    synth.fill_state(net, seed=15)
    net.train()
    x = synth.make_images(int(z["B"]), int(z["S"]), seed=151)
    out = net(x, train=True)[3]
    torch.testing.assert_close(out.detach()[:, ::3], t(z["out"]), rtol=1e-4, atol=2e-3)
    (out * cotangent(out.shape)).sum().backward()
    sd = dict(net.named_parameters())
    for name in GRADS:
        g = sd[name].grad
        want = t(z["g:" + name])
        assert float((sub(g) - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 1e-6, name
    assert sd["backbone.backbone.fc.weight"].grad is None and int(z["unused_grad_is_none"]) == 1
    msd = net.state_dict()
    for k in z.files:
        if k.startswith("b:"):
            torch.testing.assert_close(msd[k[2:]], t(z[k]), rtol=1e-5, atol=1e-6)
    net.eval()
    with torch.no_grad():
        torch.testing.assert_close(net(x, train=False)[:, ::3], t(z["out_eval"]), rtol=1e-4, atol=2e-3)


def test_mirror_parameter_tree(golden):
    from ep24 import nn as enn
    z = golden("g15_resnet")
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 1.0, backbone_type="resnet"), enn.YOLOXHead(80, 1.0))
    assert sorted(m.state_dict().keys()) == [str(k) for k in z["keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])
    ref = omodel.Net(0.33, 1.0, 80, backbone_type="resnet").state_dict()
    assert all(tuple(v.shape) == tuple(ref[k].shape) for k, v in m.state_dict().items())
    units = list(m.backbone.backbone.used_units())
    assert len(units) == 1 + 16 * 3 + 4 and units[0][0].kernel_size == (7, 7)
    assert isinstance(enn.YOLOPAFPN(1.0, 1.0, backbone_type="resnet").backbone, enn.ResNet)
