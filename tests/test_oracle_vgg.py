"""The fourth backbone of the reference's switch, vgg19(): the oracle's network against the reference (G17) and the parameter
tree of the product's mirror."""
import torch

from conftest import t
from ep24 import synth
from oracle import model as omodel
from test_oracle_resnet import cotangent, sub

VGG_GRADS = ("backbone.backbone.conv_pool1.0.conv.weight", "backbone.backbone.conv_pool1.1.bn.weight", "backbone.backbone.conv_pool3.2.conv.weight",
             "backbone.backbone.conv_pool5.3.bn.bias", "backbone.backbone.conv_add.conv.weight", "backbone.lateral_conv0.conv.weight",
             "head.stems.0.conv.weight")


def test_oracle_vgg_network_vs_reference(golden):
    z = golden("g17_vgg")
    net = omodel.Net(0.33, 1.0, 80, backbone_type="vgg")
    assert sorted(net.state_dict().keys()) == [str(k) for k in z["keys"]]
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"])
    synth.fill_state(net, seed=17)
    net.train()
    x = synth.make_images(int(z["B"]), int(z["S"]), seed=171)
    out = net(x, train=True)[3]
    torch.testing.assert_close(out.detach(), t(z["out"]), rtol=1e-4, atol=2e-3)
    (out * cotangent(out.shape)).sum().backward()
    sd = dict(net.named_parameters())
    for name in VGG_GRADS:
        want = t(z["g:" + name])
        assert float((sub(sd[name].grad) - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 1e-6, name
    msd = net.state_dict()
    for k in z.files:
        if k.startswith("b:"):
            torch.testing.assert_close(msd[k[2:]], t(z[k]), rtol=1e-5, atol=1e-6)
    net.eval()
    with torch.no_grad():
        torch.testing.assert_close(net(x, train=False), t(z["out_eval"]), rtol=1e-4, atol=2e-3)


def test_mirror_parameter_tree(golden):
    from ep24 import nn as enn
    z = golden("g17_vgg")
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 1.0, backbone_type="vgg"), enn.YOLOXHead(80, 1.0))
    assert sorted(m.state_dict().keys()) == [str(k) for k in z["keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])
    try:
        enn.YOLOPAFPN(1.0, 1.0, backbone_type="mobilenet")
        raise AssertionError("unknown backbone accepted")
    except NotImplementedError:
        pass
