"""GPU parity of the swapped backbone (BASELINE config 4: resnet50() behind the 24p PAFPN + head): the pieces (im2col stem,
max pool, ReLU, bottleneck blocks) against the oracle, the whole network against the reference-generated G15."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import t
from ep24 import synth
from test_gpu_engine import _act, cos, rel_err
from test_oracle_resnet import GRADS, cotangent, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def test_im2col_relu_maxpool_kernels():
    from ep24._lib import call, ptr, stream_ptr as sp
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 20, 24, generator=g) * 255
    OH, OW = 10, 12
    rows = torch.full((2 * OH * OW, 152), 7.0, dtype=BF, device=DEV)
    imgd = img.to(DEV)
    call("im2col_bf16", ptr(imgd), ptr(rows), 152, 2, 3, 20, 24, 7, 2, 3, sp())
    cols = F.unfold(img, 7, padding=3, stride=2).reshape(2, 3, 49, OH * OW).permute(0, 3, 2, 1).reshape(-1, 147)   # (kh,kw) major, c minor
    assert torch.equal(rows[:, :147].float().cpu(), cols.to(BF).float()) and float(rows[:, 147:].abs().sum()) == 0
    # ReLU in place and its mask
    y = torch.randn(50, 40, generator=g).to(BF).to(DEV)
    y0 = y.clone()
    call("relu_fwd", ptr(y, 8), 40, 50, 24, sp())              # a channel slice [8, 32) of rows with stride 40
    want = y0.clone()
    want[:, 8:32] = torch.relu(y0[:, 8:32])
    assert torch.equal(y, want)
    dy = torch.randn(50, 24, generator=g).to(BF).to(DEV)
    d0 = dy.clone()
    call("relu_bwd", ptr(dy), 24, ptr(y, 8), 40, 50, 24, sp())
    assert torch.equal(dy, torch.where(y[:, 8:32] > 0, d0, torch.zeros_like(d0)))
    # MaxPool2d(3, 2, 1) on a ReLU output (ties at zero): values and gradient routing equal ATen's
    x = torch.relu(torch.randn(2, 16, 14, 18, generator=g)).to(BF).float().requires_grad_(True)
    yp = F.max_pool2d(x, 3, 2, 1)
    gy = torch.randn(yp.shape, generator=g).to(BF).float()
    yp.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).reshape(-1, 16).to(BF).to(DEV)
    yd = torch.zeros(2 * 7 * 9, 16, dtype=BF, device=DEV)
    idx = torch.zeros(2 * 7 * 9 * 16, dtype=torch.uint8, device=DEV)
    call("maxpool3s2_fwd", ptr(xd), 16, ptr(yd), 16, ptr(idx), 2, 14, 18, 16, sp())
    assert torch.equal(yd.float().cpu(), yp.detach().permute(0, 2, 3, 1).reshape(-1, 16))
    gd = gy.permute(0, 2, 3, 1).reshape(-1, 16).to(BF).to(DEV)
    dx = torch.full((2 * 14 * 18, 16), 3.0, dtype=BF, device=DEV)
    call("maxpool3s2_bwd", ptr(gd), 16, ptr(idx), ptr(dx), 16, 0, 2, 14, 18, 16, sp())
    want = x.grad.permute(0, 2, 3, 1).reshape(-1, 16)
    assert rel_err(dx, want) < 1e-2 and torch.equal(dx.float().cpu() != 0, want != 0)
    call("maxpool3s2_bwd", ptr(gd), 16, ptr(idx), ptr(dx), 16, 1, 2, 14, 18, 16, sp())
    assert rel_err(dx, 2 * want) < 1e-2


def test_bottleneck_stage_vs_oracle():
    """A strided bottleneck with a downsample branch followed by an identity one: forward, input gradient, every
    parameter gradient and the running statistics against the oracle with the product's bf16 storage points emulated."""
    from ep24 import nn as enn
    from ep24.engine import Engine
    from oracle import model as om
    torch.manual_seed(3)
    down = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 1, 2, bias=False), torch.nn.BatchNorm2d(64))
    stage = torch.nn.Sequential(enn.ResBottleneck(32, 16, 2, down), enn.ResBottleneck(64, 16))
    ref = torch.nn.Sequential(om.ResBlock(32, 16, 2, torch.nn.Sequential(torch.nn.Conv2d(32, 64, 1, 2, bias=False),
                                                                          torch.nn.BatchNorm2d(64))), om.ResBlock(64, 16))
    synth.fill_state(ref, seed=4)
    stage.load_state_dict(ref.state_dict(), strict=True)
    for net in (stage, ref):
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03
    stage.to(DEV)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(4, 32, 16, 16, generator=g).to(BF).float()
    gy = torch.randn(4, 64, 8, 8, generator=g).to(BF).float()

    class StageEngine(Engine):
        def _build(self):
            xin = self.new_act(32, 16, 16)
            xin.buf.t.copy_(x.permute(0, 2, 3, 1).reshape(-1).to(BF))
            self.xin = xin
            cat = self.new_act(128, 8, 8)               # the last block writes into a concat slot, like dark3 / dark4 do
            out = self.res_block(stage[0], xin)
            out = self.res_block(stage[1], out, out=cat.slice(64, 64))
            out.gwrite()
            self.out = out
            self._finalize()

    eng = StageEngine(stage, 4, 16)
    eng.forward()
    om.EMULATE_BF16 = True
    try:
        ref.train()
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(gy)
    finally:
        om.EMULATE_BF16 = False
    assert rel_err(_act(eng.out), yr.detach()) < 1.2e-2
    o = eng.out
    o.buf.grad().view(o.buf.rows, o.buf.ld)[:, o.c0:o.c0 + o.C] = gy.permute(0, 2, 3, 1).reshape(-1, o.C).to(DEV).to(BF)
    eng.home.zero_grad()
    eng.backward(torch.zeros(1, device=DEV))
    r = eng.xin._groot()
    gx = r.buf.grad().view(r.buf.rows, r.buf.ld)[:, r.c0:r.c0 + r.C].reshape(4, 16, 16, 32).permute(0, 3, 1, 2)
    assert rel_err(gx, xr.grad) < 4e-2, rel_err(gx, xr.grad)
    rp = dict(ref.named_parameters())
    for k, p in stage.named_parameters():
        assert cos(p.grad, rp[k].grad) > 0.999 and rel_err(p.grad, rp[k].grad) < 5e-2, (k, cos(p.grad, rp[k].grad), rel_err(p.grad, rp[k].grad))
    rs = ref.state_dict()
    for k, v in stage.state_dict().items():
        if "running" in k:
            assert rel_err(v, rs[k]) < 2e-2, k


def resnet_model():
    from ep24 import nn as enn
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 1.0, backbone_type="resnet"), enn.YOLOXHead(80, 1.0))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return synth.fill_state(m, seed=15).to(DEV)


def test_resnet_network_vs_reference_golden(golden):
    """Whole network against the reference's fp32 run (G15) and against the oracle with bf16 storage emulated.  A 60-layer
    random-weight BN network amplifies rounding (the fp32 and the bf16-emulating oracle already differ at the head by
    about as much as the plan does), so the whole-net assertions are direction-level; the tight ones are per unit, below."""
    from oracle import model as om
    z = golden("g15_resnet")
    m = resnet_model()
    B, S = int(z["B"]), int(z["S"])
    x = synth.make_images(B, S, seed=151).to(DEV)
    out = m(x, train=True)[3]
    want = t(z["out"])
    out_full, out = out, out[:, ::3]
    assert out.shape == want.shape
    ref = synth.fill_state(om.Net(0.33, 1.0, 80, backbone_type="resnet"), seed=15).train()
    om.EMULATE_BF16 = True
    try:
        emu = ref(x.cpu(), train=True)[3]
        (emu * cotangent(emu.shape)).sum().backward()
        emu = emu.detach()
    finally:
        om.EMULATE_BF16 = False
    rparams = dict(ref.named_parameters())

    def cs(a, b):
        return (cos(a[..., :2], b[..., :2]), cos(a[..., 26:], b[..., 26:]), cos(torch.log(a[..., 2:26]), torch.log(b[..., 2:26])))
    c_ref, c_emu, c_base = cs(out.detach(), want), cs(out_full.detach(), emu), cs(emu[:, ::3], want)
    print("plan vs fp32 reference", c_ref, "| plan vs bf16-emulating oracle", c_emu, "| oracle bf16 vs fp32", c_base)
    assert min(c_emu) > 0.9 and c_ref[0] > 0.97 and c_ref[2] > 0.9 and c_ref[1] > c_base[1] - 0.1
    out_full.backward(cotangent(out_full.shape).to(DEV))
    params = dict(m.named_parameters())
    for name in GRADS:
        g = params[name].grad
        c = cos(sub(g.cpu()), t(z["g:" + name]))
        ce = cos(g.cpu(), rparams[name].grad)
        print(name, "grad cos vs fp32 %.4f, vs bf16-emulating oracle %.4f  norm %.4g vs %.4g" % (c, ce, float(g.double().norm()), float(z["gn:" + name])))
        # direction is not reproducible through ~60 random-weight BN layers in bf16 (even lateral_conv0, on the darknet-tested
        # neck, decorrelates); magnitudes are: a missing or doubled branch of the backward graph would move them
        assert torch.isfinite(g).all() and g.shape == params[name].shape
        assert 0.7 < float(g.double().norm()) / float(z["gn:" + name]) < 1.4, name
    # parameters the reference never runs keep a zero gradient
    assert float(params["backbone.backbone.fc.weight"].grad.abs().sum()) == 0
    assert float(params["backbone.backbone.baseconv2.0.weight"].grad.abs().sum()) == 0
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("b:") and "bn1" in k:                      # the stem's statistics see no amplified noise yet
            assert rel_err(sd[k[2:]], t(z[k])) < 2e-2, k
    m.eval()
    oe, we = m(x, train=False)[:, ::3], t(z["out_eval"])
    print("eval", cs(oe, we))
    assert oe.shape == we.shape and bool(torch.isfinite(oe[..., 26:]).all())


def test_resnet_units_vs_oracle_on_the_plans_own_inputs():
    """Every conv unit of the swapped backbone, fed the plan's own input activation, against the oracle's
    conv -> BN -> (+identity) -> ReLU with bf16 storage emulated; the max pool exactly."""
    from oracle import model as om
    m = resnet_model()
    B, S = 2, 128
    x = synth.make_images(B, S, seed=9).to(DEV)
    m(x, train=True)
    eng = m.engine(B, S)
    bb = m.backbone.backbone
    ref = om.Net(0.33, 1.0, 80, backbone_type="resnet")
    synth.fill_state(ref, seed=15)
    rb = ref.backbone.backbone
    rb.train()
    om.EMULATE_BF16 = True
    worst = 0.0
    try:
        with torch.no_grad():
            stem_out = _act(eng.unit_acts[bb.conv1][2])
            want = om.conv_bn_act(x.cpu(), rb.conv1, rb.bn1, "relu", True)
            assert rel_err(stem_out, want) < 1.2e-2, rel_err(stem_out, want)
            first_in = _act(eng.unit_acts[bb.layer1[0].conv1][0])
            assert torch.equal(first_in, F.max_pool2d(stem_out, 3, 2, 1))
            for lname in ("layer1", "layer2", "layer3", "layer4"):
                for blk, rblk in zip(getattr(bb, lname), getattr(rb, lname)):
                    xin = _act(eng.unit_acts[blk.conv1][0])
                    idn = xin
                    if blk.downsample is not None:
                        idn = _act(eng.unit_acts[blk.downsample[0]][2])
                        e = rel_err(idn, om.conv_bn_act(xin, rblk.downsample[0], rblk.downsample[1], None, True))
                        worst = max(worst, e)
                        assert e < 1.2e-2, (lname, "downsample", e)
                    t1 = _act(eng.unit_acts[blk.conv1][2])
                    t2 = _act(eng.unit_acts[blk.conv2][2])
                    y = _act(eng.unit_acts[blk.conv3][2])
                    for got, wnt, tag in ((t1, om.conv_bn_act(xin, rblk.conv1, rblk.bn1, "relu", True), "conv1"),
                                          (t2, om.conv_bn_act(t1, rblk.conv2, rblk.bn2, "relu", True), "conv2"),
                                          (y, om.conv_bn_act(t2, rblk.conv3, rblk.bn3, "relu", True, residual=idn), "conv3")):
                        e = rel_err(got, wnt)
                        worst = max(worst, e)
                        assert e < 1.2e-2, (lname, tag, e)
    finally:
        om.EMULATE_BF16 = False
    print("worst per-unit rel err", worst)


def test_resnet_training_step_runs_captured():
    """The captured step (two-lane backward, slab weight gradients, fused SGD) over the swapped network: finite, moving,
    and the replay equals the eager launch lists."""
    from ep24 import loss as eloss, train as etrain
    m = resnet_model()
    B, S = 2, 128
    lf = eloss.Loss_Function(80)
    ts = etrain.TrainStep(m, lf, lr=0.001, momentum=0.9, batch=B, size=S)
    images = synth.make_images(B, S, seed=1).to(DEV)
    labels = synth.make_labels(B, [3, 2], size=S, seed=2).to(DEV)
    losses = [float(ts.step(images, labels)[0]) for _ in range(4)]
    print(losses)
    assert all(np.isfinite(losses)) and losses[0] != losses[1]
    assert float(dict(m.named_parameters())["backbone.backbone.fc.weight"].grad.abs().sum()) == 0
