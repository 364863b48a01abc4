"""GPU parity of the second swapped backbone of BASELINE config 4, densenet121(): the extra kernels, a dense block +
transition against the oracle (forward and backward, Dropout2d factors fixed), every layer of the full network on the
plan's own tensors, the whole network against the reference-generated G16 (recorded dropout draws replayed)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import t
from ep24 import synth
from test_gpu_engine import _act, _gact, cos, rel_err
from test_oracle_densenet import DENSE_GRADS
from test_oracle_resnet import cotangent, sub

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def test_colstats_gather_avgpool_chanscale_kernels():
    from ep24._lib import call, ptr, stream_ptr as sp
    g = torch.Generator().manual_seed(8)
    x = torch.randn(300, 48, generator=g).to(BF)
    xd = x.to(DEV)
    stats = torch.zeros(8, 2, 64, dtype=torch.int64, device=DEV)
    call("colstats", ptr(xd, 8), 48, ptr(stats, 16), 64, 300, 32, sp())          # columns [8, 40) -> block channels [16, 48)
    xf = x.float()[:, 8:40]
    got = stats[0].double().cpu() / 1048576.0
    assert torch.allclose(got[0, 16:48], xf.double().sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(got[1, 16:48], (xf.double() ** 2).sum(0), rtol=1e-5, atol=1e-3)
    assert float(got[:, :16].abs().sum() + got[:, 48:].abs().sum()) == 0 and float(stats[1:].abs().sum()) == 0
    dst = torch.full((8, 2, 40), -1, dtype=torch.int64, device=DEV)
    call("stats_gather", ptr(stats), 64, ptr(dst), 40, 8, sp())
    assert torch.equal(dst, stats[:, :, :40])
    # AvgPool2d(2, 2) forward / backward
    a = torch.randn(2, 16, 6, 8, generator=g).to(BF).float().requires_grad_(True)
    y = F.avg_pool2d(a, 2, 2)
    gy = torch.randn(y.shape, generator=g).to(BF).float()
    y.backward(gy)
    ad = a.detach().permute(0, 2, 3, 1).reshape(-1, 16).to(BF).to(DEV)
    yd = torch.zeros(2 * 3 * 4, 16, dtype=BF, device=DEV)
    call("avgpool2_fwd", ptr(ad), 16, ptr(yd), 16, 2, 6, 8, 16, sp())
    assert rel_err(yd, y.detach().permute(0, 2, 3, 1).reshape(-1, 16)) < 5e-3
    gd = gy.permute(0, 2, 3, 1).reshape(-1, 16).to(BF).to(DEV)
    dx = torch.full((2 * 6 * 8, 16), 1.0, dtype=BF, device=DEV)
    call("avgpool2_bwd", ptr(gd), 16, ptr(dx), 16, 1, 2, 6, 8, 16, sp())
    assert rel_err(dx, a.grad.permute(0, 2, 3, 1).reshape(-1, 16) + 1.0) < 1e-2
    # Dropout2d as per-(sample, channel) factors
    v = torch.randn(2 * 5, 16, generator=g).to(BF).to(DEV)
    keep = ((torch.rand(2, 16, generator=g) > 0.3).float() / 0.7).to(DEV)
    v0 = v.clone()
    call("chanscale", ptr(v), 16, ptr(keep), 2, 5, 16, sp())
    want = (v0.float().reshape(2, 5, 16) * keep[:, None, :]).reshape(-1, 16)
    assert rel_err(v, want) < 5e-3 and torch.equal(v.float() == 0, want == 0)


def test_bn_apply_accumulating_form_is_apply_plus_add():
    from ep24._lib import call, ptr, stream_ptr as sp
    g = torch.Generator().manual_seed(12)
    M, C = 700, 96
    dy, z = [torch.randn(M, C, generator=g).to(BF).to(DEV) for _ in range(2)]
    save = torch.cat([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).to(DEV)
    gam, bet = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV)
    sg, sb = torch.zeros(C, dtype=torch.int64, device=DEV), torch.zeros(C, dtype=torch.int64, device=DEV)
    call("bn_act_bwd_reduce", ptr(dy), C, ptr(z), C, ptr(save), ptr(gam), ptr(bet), ptr(sg), ptr(sb), M, C, 2, 1, sp())
    old = torch.randn(M, 128, generator=g).to(BF).to(DEV)
    plain = torch.zeros(M, C, dtype=BF, device=DEV)
    gg, gb = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    call("bn_act_bwd_apply", ptr(dy), C, ptr(z), C, ptr(save), ptr(gam), ptr(bet), ptr(sg), ptr(sb), ptr(gg), ptr(gb), ptr(plain), C, M, C, 2, 1, sp())
    acc = old.clone()
    call("bn_act_bwd_apply_acc", ptr(dy), C, ptr(z), C, ptr(save), ptr(gam), ptr(bet), ptr(sg), ptr(sb), ptr(gg), ptr(gb), ptr(acc, 16), 128, M, C, 2, 1, sp())
    want = old.clone()
    want[:, 16:16 + C] = (old[:, 16:16 + C].float() + plain.float()).to(BF)
    assert torch.equal(acc, want) and float(plain.float().abs().sum()) > 0


@pytest.mark.parametrize("nl,drop", [(1, False), (3, False), (3, True)])
def test_dense_block_and_transition_vs_oracle(nl, drop):
    from ep24 import nn as enn
    from ep24.engine import Engine
    from oracle import model as om
    ref = om.DenseNetBackbone(blocks=(nl, 1, 1, 1))
    full = enn.DenseNet(32, (nl, 1, 1, 1))
    synth.fill_state(ref, seed=5)
    full.load_state_dict(ref.state_dict(), strict=True)
    for net in (full, ref):
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03
    stage = torch.nn.ModuleList([full.D1, full.T1]).to(DEV)
    g = torch.Generator().manual_seed(6)
    B, H = 4, 16
    x = torch.randn(B, 64, H, H, generator=g).to(BF).float()
    ct = 64 + 32 * nl
    gy = torch.randn(B, ct // 2, H // 2, H // 2, generator=g).to(BF).float()
    keep = (torch.rand(nl, B, 32, generator=g) > 0.3).float() / 0.7 if drop else torch.ones(nl, B, 32)

    class StageEngine(Engine):
        def _build(self):
            cat = self.new_act(ct, H, H)
            head = cat.slice(0, 64)
            head.buf.t.view(head.buf.rows, head.buf.ld)[:, :64] = x.permute(0, 2, 3, 1).reshape(-1, 64).to(BF).to(self.dev)
            self.xin = head
            bstats = self._stats_slot(cat.C)
            self.drop_keep = keep.to(self.dev)
            self.drop_p, self.fixed_dropout = 0.3, True
            self._f("colstats", head.ptr(), head.ld, bstats, cat.C, head.M, 64, ev=False)
            self.dense_block(full.D1, cat, 64, bstats, 0)
            tconv, tbn = full.T1.trans[0].conv, full.T1.trans[0].bn
            tt = self.new_act(ct // 2, H, H)
            self.conv_raw(tconv, self.pre_bn(tbn, cat, bstats, cat.C), tt)
            out = self.new_act(ct // 2, H // 2, H // 2)
            self.avgpool2(tt, out)
            out.gwrite()
            self.out, self.cat = out, cat
            self._finalize()

    eng = StageEngine(stage, B, H)
    eng.forward()
    om.EMULATE_BF16 = True
    try:
        ref.train()
        ref.keep = keep
        xr = x.clone().requires_grad_(True)
        catr, _ = ref._block(ref.D1, xr, 0)
        yr = ref._trans(ref.T1, catr)
        yr.backward(gy)
    finally:
        om.EMULATE_BF16 = False
    assert rel_err(_act(eng.cat), catr.detach()) < 1.2e-2
    assert rel_err(_act(eng.out), yr.detach()) < 1.2e-2
    o = eng.out
    o.buf.grad().view(o.buf.rows, o.buf.ld)[:, o.c0:o.c0 + o.C] = gy.permute(0, 2, 3, 1).reshape(-1, o.C).to(DEV).to(BF)
    eng.home.zero_grad()
    eng.backward(torch.zeros(1, device=DEV))
    r = eng.xin._groot()
    gx = r.buf.grad().view(r.buf.rows, r.buf.ld)[:, r.c0:r.c0 + r.C].reshape(B, H, H, 64).permute(0, 3, 1, 2)
    rp = dict(ref.named_parameters())
    rs = {k: v.clone() for k, v in ref.state_dict().items()}          # before the teacher-forced calls advance them again
    names = {id(p): n for n, p in full.named_parameters()}
    report = [("gx", cos(gx, xr.grad), rel_err(gx, xr.grad))]
    for p in stage.parameters():
        k = names[id(p)]
        report.append((k, cos(p.grad, rp[k].grad), rel_err(p.grad, rp[k].grad)))
    for r in report:
        print("%-44s cos %.5f rel %.4f" % r)
    assert len(report) == 1 + nl * 6 + 3
    # end to end the gradient passes through two bf16-stored BatchNorm backward stages per dense layer (and one bf16
    # accumulation per consumer of the concatenation): direction within 0.5 %; the exact bookkeeping is teacher-forced below
    assert all(c > 0.995 for _, c, e in report), min(c for _, c, e in report)
    # teacher-forced per layer: the oracle op gets the plan's own input and the plan's own output gradient and must give
    # the plan's parameter gradients (each written by exactly one layer)
    pg = {names[id(p)]: p.grad.float().cpu() for p in stage.parameters()}
    tf = []

    def grads(fn, inputs, params, gout):
        for q in params:
            q.grad = None
        fn(*inputs).backward(gout)
        return [q.grad for q in params]

    om.EMULATE_BF16 = True
    try:
        for i, (lay, rlay) in enumerate(zip(full.D1.denseblock, ref.D1.denseblock)):
            (c1, c2), (r1, r2) = lay.conv_block, rlay.conv_block
            xin = _act(eng.pre_bn_inputs[c1.bn])
            a, z1, b = eng.unit_acts[c1.conv]
            pre = "D1.denseblock.%d.conv_block." % i
            gw, gb = grads(lambda v: om._q(F.relu(F.batch_norm(v, None, None, r1.bn.weight, r1.bn.bias, True, 0.0, 1e-3))), [xin],
                           [r1.bn.weight, r1.bn.bias], _gact(a))
            tf += [(pre + "0.bn.weight", gw), (pre + "0.bn.bias", gb)]
            gc, gw, gb = grads(lambda v: om.conv_bn_act(v, r1.conv, r2.bn, "relu", True), [_act(a)], [r1.conv.weight, r2.bn.weight, r2.bn.bias],
                               _gact(b))
            tf += [(pre + "0.conv.weight", gc), (pre + "1.bn.weight", gw), (pre + "1.bn.bias", gb)]
            chunk = eng.unit_acts[c2.conv][2]
            (gc,) = grads(lambda v: F.conv2d(om._q(v), om._q(r2.conv.weight), None, 1, 1), [_act(b)], [r2.conv.weight], _gact(chunk))
            tf += [(pre + "1.conv.weight", gc)]
    finally:
        om.EMULATE_BF16 = False
    for k, want in tf:
        c, e = cos(pg[k], want), rel_err(pg[k], want)
        print("teacher-forced %-40s cos %.5f rel %.4f" % (k, c, e))
        assert c > 0.9995 and e < 3e-2, (k, c, e)
    for k, v in full.state_dict().items():
        if "running" in k and (k.startswith("D1") or k.startswith("T1")):
            assert rel_err(v, rs[k]) < 2e-2, k


def dense_model():
    from ep24 import nn as enn
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 1.0, backbone_type="densenet"), enn.YOLOXHead(80, 1.0))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return synth.fill_state(m, seed=16).to(DEV)


def test_densenet_network_vs_reference_golden_and_per_layer(golden):
    from oracle import model as om
    z = golden("g16_densenet")
    m = dense_model()
    B, S = int(z["B"]), int(z["S"])
    eng = m.engine(B, S)
    eng.fixed_dropout = True
    eng.drop_keep.copy_(t(z["keep"]).to(DEV))                     # the reference's own Dropout2d draws
    x = synth.make_images(B, S, seed=162).to(DEV)
    out_full = m(x, train=True)[3]
    out, want = out_full[:, ::3], t(z["out"])
    ref = synth.fill_state(om.Net(0.33, 1.0, 80, backbone_type="densenet"), seed=16).train()
    rb = ref.backbone.backbone
    rb.keep = t(z["keep"])
    bb = m.backbone.backbone
    om.EMULATE_BF16 = True
    worst = 0.0
    try:
        emu = ref(x.cpu(), train=True)[3].detach()
        # every dense layer and transition on the plan's own tensors
        with torch.no_grad():
            li = 0
            for bname in ("D1", "D2", "D3", "D4"):
                for lay, rlay in zip(getattr(bb, bname).denseblock, getattr(rb, bname).denseblock):
                    (c1, c2), (r1, r2) = lay.conv_block, rlay.conv_block
                    xin, z1, _ = eng.unit_acts[c1.conv]
                    want_z1 = om.bn_act_conv(_act(eng_input(eng, c1)), r1.bn, r1.conv, True)
                    e1 = rel_err(_act(z1), want_z1)
                    b_in, _, chunk = eng.unit_acts[c2.conv]
                    want_z2 = om._q(om.bn_act_conv(_act(z1), r2.bn, r2.conv, True) * rb.keep[li].view(B, 32, 1, 1))
                    e2 = rel_err(_act(chunk), want_z2)
                    worst = max(worst, e1, e2)
                    assert e1 < 1.2e-2 and e2 < 1.2e-2, (bname, li, e1, e2)
                    li += 1
    finally:
        om.EMULATE_BF16 = False

    def cs(a, b):
        return (cos(a[..., :2], b[..., :2]), cos(a[..., 26:], b[..., 26:]), cos(torch.log(a[..., 2:26]), torch.log(b[..., 2:26])))
    c_ref, c_emu, c_base = cs(out.detach(), want), cs(out_full.detach(), emu), cs(emu[:, ::3], want)
    print("worst per-layer rel err", worst, "| plan vs fp32 reference", c_ref, "| vs bf16-emulating oracle", c_emu, "| oracle bf16 vs fp32", c_base)
    assert min(c_emu) > 0.9 and c_ref[0] > 0.97 and c_ref[2] > 0.9 and c_ref[1] > c_base[1] - 0.1
    out_full.backward(cotangent(out_full.shape).to(DEV))
    params = dict(m.named_parameters())
    for name in DENSE_GRADS:
        g = params[name].grad
        ratio = float(g.double().norm()) / float(z["gn:" + name])
        print(name, "grad cos vs fp32 %.3f  norm ratio %.3f" % (cos(sub(g.cpu()), t(z["g:" + name])), ratio))
        assert torch.isfinite(g).all() and 0.7 < ratio < 1.4, (name, ratio)
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("b:") and "stem" in k:
            assert rel_err(sd[k[2:]], t(z[k])) < 2e-2, k
    m.eval()
    oe, we = m(x, train=False)[:, ::3], t(z["out_eval"])
    print("eval", cs(oe, we))
    assert cos(oe[..., 26:], we[..., 26:]) > 0.97 and cos(oe[..., :2], we[..., :2]) > 0.99


def eng_input(eng, cb):
    """The concat prefix a ConvBlock's pre-activation BN reads = the input of its BN+ReLU output's producer."""
    a, _, _ = eng.unit_acts[cb.conv]          # (a = relu(bn(x)), z, out) of the conv1 unit: a's source prefix has a.C channels
    return eng.pre_bn_inputs[cb.bn]


def test_densenet_training_step_runs_captured():
    from ep24 import loss as eloss, train as etrain
    m = dense_model()
    B, S = 2, 128
    lf = eloss.Loss_Function(80)
    ts = etrain.TrainStep(m, lf, lr=0.001, momentum=0.9, batch=B, size=S)
    images = synth.make_images(B, S, seed=1).to(DEV)
    labels = synth.make_labels(B, [3, 2], size=S, seed=2).to(DEV)
    k0 = ts.eng.drop_keep.clone()
    losses = [float(ts.step(images, labels)[0]) for _ in range(4)]
    print(losses)
    assert all(np.isfinite(losses)) and losses[0] != losses[1]
    assert not torch.equal(k0, ts.eng.drop_keep) and set(ts.eng.drop_keep.unique().tolist()) <= {0.0, float(torch.tensor(1.0) / 0.7)}
