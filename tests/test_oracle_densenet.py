"""BASELINE config 4, densenet121(): the oracle's network against the reference (G16, with the reference's recorded
Dropout2d draws replayed) and the parameter tree of the product's mirror."""
import torch

from conftest import t
from ep24 import synth
from oracle import model as omodel
from test_oracle_resnet import cotangent, sub

DENSE_GRADS = ("backbone.backbone.stem.0.conv.weight", "backbone.backbone.D1.denseblock.0.conv_block.0.bn.weight",
               "backbone.backbone.D1.denseblock.5.conv_block.1.conv.weight", "backbone.backbone.T1.trans.0.conv.weight",
               "backbone.backbone.D2.denseblock.3.conv_block.0.conv.weight", "backbone.backbone.baseconv1.conv.weight",
               "backbone.backbone.D3.denseblock.23.conv_block.1.bn.bias", "backbone.backbone.T3.trans.0.bn.weight",
               "backbone.backbone.D4.denseblock.15.conv_block.1.conv.weight", "backbone.lateral_conv0.conv.weight",
               "head.stems.0.conv.weight")


def test_oracle_densenet_network_vs_reference(golden):
    z = golden("g16_densenet")
    net = omodel.Net(0.33, 1.0, 80, backbone_type="densenet")
    assert sorted(net.state_dict().keys()) == [str(k) for k in z["keys"]]
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"])
    synth.fill_state(net, seed=16)
    net.train()
    net.backbone.backbone.keep = t(z["keep"])
    x = synth.make_images(int(z["B"]), int(z["S"]), seed=162)
    out = net(x, train=True)[3]
    torch.testing.assert_close(out.detach()[:, ::3], t(z["out"]), rtol=1e-4, atol=2e-3)
    (out * cotangent(out.shape)).sum().backward()
    sd = dict(net.named_parameters())
    for name in DENSE_GRADS:
        want = t(z["g:" + name])
        assert float((sub(sd[name].grad) - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 1e-6, name
    msd = net.state_dict()
    for k in z.files:
        if k.startswith("b:"):
            torch.testing.assert_close(msd[k[2:]], t(z[k]), rtol=1e-5, atol=1e-6)
    net.eval()
    with torch.no_grad():
        torch.testing.assert_close(net(x, train=False)[:, ::3], t(z["out_eval"]), rtol=1e-4, atol=2e-3)


def test_mirror_parameter_tree(golden):
    from ep24 import nn as enn
    z = golden("g16_densenet")
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 1.0, backbone_type="densenet"), enn.YOLOXHead(80, 1.0))
    assert sorted(m.state_dict().keys()) == [str(k) for k in z["keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])
    drops = [d for d in m.modules() if isinstance(d, torch.nn.Dropout2d)]
    assert len(drops) == 58 and all(d.p == 0.3 for d in drops)
