"""GPU parity of the fourth backbone of the reference's switch, vgg19(): the 2x2 max pool kernel against ATen, every unit of the
backbone on the plan's own inputs against the bf16-emulating oracle, the whole network against the reference-generated G17,
and a captured training step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import t
from ep24 import synth
from test_gpu_engine import _act, cos, rel_err
from test_oracle_resnet import cotangent
from test_oracle_vgg import VGG_GRADS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def test_maxpool2_kernel_vs_aten():
    from ep24._lib import call, ptr, stream_ptr as sp
    g = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn(2, 16, 12, 10, generator=g)).to(BF).float().requires_grad_(True)      # ties at zero
    y = F.max_pool2d(x, 2, 2)
    gy = torch.randn(y.shape, generator=g).to(BF).float()
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).reshape(-1, 16).to(BF).to(DEV)
    yd = torch.zeros(2 * 6 * 5, 16, dtype=BF, device=DEV)
    idx = torch.zeros(2 * 6 * 5 * 16, dtype=torch.uint8, device=DEV)
    call("maxpool2_fwd", ptr(xd), 16, ptr(yd), 16, ptr(idx), 2, 12, 10, 16, sp())
    assert torch.equal(yd.float().cpu(), y.detach().permute(0, 2, 3, 1).reshape(-1, 16))
    gd = gy.permute(0, 2, 3, 1).reshape(-1, 16).to(BF).to(DEV)
    dx = torch.full((2 * 12 * 10, 16), 5.0, dtype=BF, device=DEV)
    call("maxpool2_bwd", ptr(gd), 16, ptr(idx), ptr(dx), 16, 0, 2, 12, 10, 16, sp())
    want = x.grad.permute(0, 2, 3, 1).reshape(-1, 16)
    assert torch.equal(dx.float().cpu(), want)                    # one contribution per element: exact, routing included
    call("maxpool2_bwd", ptr(gd), 16, ptr(idx), ptr(dx), 16, 1, 2, 12, 10, 16, sp())
    assert rel_err(dx, 2 * want) < 1e-2


def vgg_model():
    from ep24 import nn as enn
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 1.0, backbone_type="vgg"), enn.YOLOXHead(80, 1.0))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return synth.fill_state(m, seed=17).to(DEV)


def test_vgg_network_vs_reference_golden_and_per_unit(golden):
    from oracle import model as om
    z = golden("g17_vgg")
    m = vgg_model()
    B, S = int(z["B"]), int(z["S"])
    x = synth.make_images(B, S, seed=171).to(DEV)
    out = m(x, train=True)[3]
    want = t(z["out"])
    assert out.shape == want.shape
    eng = m.engine(B, S)
    bb = m.backbone.backbone
    ref = synth.fill_state(om.Net(0.33, 1.0, 80, backbone_type="vgg"), seed=17).train()
    rb = ref.backbone.backbone
    om.EMULATE_BF16 = True
    worst, checked = 0.0, 0
    try:
        emu = ref(x.cpu(), train=True)[3].detach()
        with torch.no_grad():
            prev = None
            for stage, rstage in zip(bb.stages() + (torch.nn.Sequential(bb.conv_add),), (rb.conv_pool1, rb.conv_pool2, rb.conv_pool3, rb.conv_pool4,
                                                                                           rb.conv_pool5, torch.nn.Sequential(rb.conv_add))):
                for mod, rmod in zip(stage, rstage):
                    if isinstance(mod, torch.nn.MaxPool2d):
                        continue
                    xin, _, y = eng.unit_acts[mod.conv]
                    src = x.cpu() if prev is None else _act(xin)          # the first unit reads im2col rows: feed the image
                    e = rel_err(_act(y), om.conv_bn_act(src, rmod.conv, rmod.bn, "relu", True))
                    worst = max(worst, e)
                    assert e < 1.2e-2, (mod, e)
                    prev = y
                    checked += 1
            # the pools: stage input of stage 2 = pooled output of stage 1, exactly
            s1_last = eng.unit_acts[bb.conv_pool1[1].conv][2]
            s2_first_in = eng.unit_acts[bb.conv_pool2[0].conv][0]
            assert torch.equal(_act(s2_first_in), F.max_pool2d(_act(s1_last), 2, 2))
    finally:
        om.EMULATE_BF16 = False
    assert checked == 17

    def cs(a, b):
        return (cos(a[..., :2], b[..., :2]), cos(a[..., 26:], b[..., 26:]), cos(torch.log(a[..., 2:26]), torch.log(b[..., 2:26])))
    c_ref, c_emu, c_base = cs(out.detach(), want), cs(out.detach(), emu), cs(emu, want)
    print("worst per-unit rel err", worst, "| plan vs fp32 reference", c_ref, "| vs bf16-emulating oracle", c_emu, "| oracle bf16 vs fp32", c_base)
    assert min(c_emu) > 0.9 and c_ref[0] > 0.97 and c_ref[2] > 0.9 and c_ref[1] > c_base[1] - 0.1
    out.backward(cotangent(out.shape).to(DEV))
    params = dict(m.named_parameters())
    for name in VGG_GRADS:
        g = params[name].grad
        ratio = float(g.double().norm()) / float(z["gn:" + name])
        print(name, "grad norm ratio %.3f" % ratio)
        assert torch.isfinite(g).all() and 0.7 < ratio < 1.4, (name, ratio)
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("b:"):
            assert rel_err(sd[k[2:]], t(z[k])) < 2e-2, k
    m.eval()
    oe, we = m(x, train=False), t(z["out_eval"])
    print("eval", cs(oe, we))
    assert oe.shape == we.shape and bool(torch.isfinite(oe[..., 26:]).all())


def test_vgg_training_step_runs_captured():
    from ep24 import loss as eloss, train as etrain
    m = vgg_model()
    B, S = 2, 128
    ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=B, size=S)
    images = synth.make_images(B, S, seed=1).to(DEV)
    labels = synth.make_labels(B, [3, 2], size=S, seed=2).to(DEV)
    losses = [float(ts.step(images, labels)[0]) for _ in range(4)]
    print(losses)
    assert all(np.isfinite(losses)) and losses[0] != losses[1]
