"""Oracle network vs golden vectors generated from the reference model code (G7)."""
import numpy as np
import pytest
import torch

from conftest import t
from oracle import model as om

BLOCKS = {
    "baseconv3": lambda: om.Unit(16, 24, 3, 1),
    "baseconv3s2": lambda: om.Unit(16, 32, 3, 2),
    "baseconv1": lambda: om.Unit(16, 8, 1, 1),
    "bottleneck": lambda: om.Res(16, True),
    "csp": lambda: om.CSP(16, 16, 2),
    "csp_noshort": lambda: om.CSP(32, 16, 1, add=False),
    "spp": lambda: om.SPP(16, 16),
    "focus": lambda: om.Stem(3, 8, 3),
}


def load_weights(mod, z):
    sd = {k[2:]: t(z[k]) for k in z.files if k.startswith("w:")}
    missing = mod.load_state_dict(sd, strict=True)
    return missing


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_g7_block(golden, name):
    z = golden("g7_" + name)
    mod = BLOCKS[name]()
    load_weights(mod, z)
    mod.train()
    x = t(z["x"]).clone().requires_grad_(True)
    y = mod(x)
    torch.testing.assert_close(y.detach(), t(z["y"]), rtol=1e-5, atol=1e-5)
    y.backward(t(z["gy"]))
    torch.testing.assert_close(x.grad, t(z["gx"]), rtol=1e-4, atol=1e-5)
    for k, p in mod.named_parameters():
        torch.testing.assert_close(p.grad, t(z["g:" + k]), rtol=1e-4, atol=1e-4)
    for k, v in mod.state_dict().items():
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(v, t(z["after:" + k]), rtol=1e-5, atol=1e-6)


def test_g7_tiny_model(golden):
    z = golden("g7_model_tiny")
    net = om.Net(0.33, 0.125)
    load_weights(net, z)
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"])
    net.train()
    xs, ys, ss, out, extra = net(t(z["x"]), train=True)
    assert extra == []
    torch.testing.assert_close(out.detach(), t(z["out"]), rtol=1e-4, atol=1e-4)
    assert torch.equal(xs[0], t(z["x_shift0"])) and torch.equal(ys[1], t(z["y_shift1"])) and torch.equal(ss[2], t(z["stride2"]))
    out.backward(t(z["gy"]))
    params = dict(net.named_parameters())
    for k in z.files:
        if k.startswith("g:"):
            torch.testing.assert_close(params[k[2:]].grad, t(z[k]), rtol=2e-3, atol=1e-5)
    sd = net.state_dict()
    torch.testing.assert_close(sd["backbone.backbone.stem.conv.bn.running_mean"], t(z["after:stem_rm"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sd["backbone.backbone.stem.conv.bn.running_var"], t(z["after:stem_rv"]), rtol=1e-5, atol=1e-5)
    net.eval()
    with torch.no_grad():
        torch.testing.assert_close(net(t(z["x"]), train=False), t(z["out_eval"]), rtol=1e-4, atol=1e-4)


def test_l_state_dict_keys_match_reference(golden):
    z = golden("g7_model_l_keys")
    net = om.Net(1.0, 1.0)
    sd = net.state_dict()
    ref = {str(k): str(s) for k, s in zip(z["keys"], z["shapes"])}
    assert {k: str(tuple(v.shape)) for k, v in sd.items()} == ref        # same names and shapes (order-free)
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"]) == 54225857


# ---- G19: the depthwise variants (DWConv, network_blocks.py:57-76) from the reference's own modules
DW_BLOCKS = {
    "dwconv3": lambda: om.DW(16, 24, 3, 1),
    "dwconv3s2": lambda: om.DW(16, 32, 3, 2),
    "bottleneck_dw": lambda: om.Res(16, True, depthwise=True),
    "csp_dw": lambda: om.CSP(16, 16, 2, depthwise=True),
}


@pytest.mark.parametrize("name", sorted(DW_BLOCKS))
def test_g19_depthwise_block(golden, name):
    z = golden("g19_" + name)
    mod = DW_BLOCKS[name]()
    load_weights(mod, z)
    mod.train()
    x = t(z["x"]).clone().requires_grad_(True)
    y = mod(x)
    torch.testing.assert_close(y.detach(), t(z["y"]), rtol=1e-5, atol=1e-5)
    y.backward(t(z["gy"]))
    torch.testing.assert_close(x.grad, t(z["gx"]), rtol=1e-4, atol=1e-5)
    for k, p in mod.named_parameters():
        torch.testing.assert_close(p.grad, t(z["g:" + k]), rtol=1e-4, atol=1e-4)
    for k, v in mod.state_dict().items():
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(v, t(z["after:" + k]), rtol=1e-5, atol=1e-6)
    mod.eval()
    with torch.no_grad():
        torch.testing.assert_close(mod(t(z["x"])), t(z["y_eval"]), rtol=1e-5, atol=1e-5)


def test_g19_depthwise_model(golden):
    z = golden("g19_model_dw_tiny")
    net = om.Net(0.33, 0.125, depthwise=True)
    ref = {str(k): str(s) for k, s in zip(z["keys"], z["shapes"])}          # (registration order inside the head differs: compare by name)
    assert {k: str(tuple(v.shape)) for k, v in net.state_dict().items()} == ref
    load_weights(net, z)
    assert sum(p.numel() for p in net.parameters()) == int(z["n_params"])
    net.train()
    out = net(t(z["x"]), train=True)[3]
    torch.testing.assert_close(out.detach(), t(z["out"]), rtol=1e-4, atol=1e-4)
    out.backward(t(z["gy"]))
    params = dict(net.named_parameters())
    for k in z.files:
        if k.startswith("g:"):
            torch.testing.assert_close(params[k[2:]].grad, t(z[k]), rtol=2e-3, atol=1e-5)
    net.eval()
    with torch.no_grad():
        torch.testing.assert_close(net(t(z["x"]), train=False), t(z["out_eval"]), rtol=1e-4, atol=1e-4)
