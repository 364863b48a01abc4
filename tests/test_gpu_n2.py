"""GPU parity of the SURVEY 8f N2 pieces: the L1 branch of the loss (vs reference-generated G12 and the oracle), the
EMA kernels (bit-exact vs reference-generated G11) and the captured step that follows a learning-rate schedule, keeps
an EMA copy and switches the L1 branch on without leaving the device."""
import numpy as np
import pytest
import torch

from conftest import t
from ep24 import synth
from test_n2_host import l1_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _l1_inputs(labels, raw, origin):
    outputs = synth.decode_head(raw).to(DEV).requires_grad_(True)
    origin = [o.to(DEV).requires_grad_(True) for o in origin]
    tup5 = list(synth.outputs_train_tuple(outputs))
    tup5[4] = origin
    return outputs, origin, tuple(tup5)


def test_loss_l1_vs_golden(golden):
    from ep24 import loss as L
    z = golden("g12_loss_l1")
    labels, raw, origin = l1_case(z)
    outputs, origin, tup5 = _l1_inputs(labels, raw, origin)
    lf = L.Loss_Function(80)
    lf.use_l1 = True
    tup = lf(tup5, labels.to(DEV))
    tup[0].backward()
    for k, v in (("loss", tup[0]), ("loss_iou_w", tup[1]), ("loss_obj", tup[2]), ("loss_cls", tup[3]), ("loss_l1", tup[4])):
        torch.testing.assert_close(v.detach().cpu(), t(z[k]), rtol=1e-4, atol=1e-6)      # north star: fp32 loss within 1e-4
    assert abs(float(tup[5]) - float(z["fg_per_gt"])) < 1e-6
    g = torch.cat([o.grad for o in origin], 1).reshape(-1, 26).cpu()
    rows = t(z["d_origin_rows"])
    assert torch.equal(g.abs().sum(-1).nonzero().reshape(-1), rows)                     # exactly the matched anchors
    torch.testing.assert_close(g[rows], t(z["d_origin_vals"]), rtol=1e-6, atol=0)        # +-1/num_fg
    go = outputs.grad.cpu()
    assert abs(float(go.double().abs().sum()) - float(z["grad_abs_sum"])) < 1e-4 * float(z["grad_abs_sum"])
    torch.testing.assert_close(go[..., 26].reshape(-1)[::7], t(z["grad_obj"]), rtol=1e-4, atol=1e-9)
    # without origin_preds the switch cannot work (the reference's torch.cat of an empty list raises too)
    with pytest.raises((RuntimeError, ValueError)):
        lf(synth.outputs_train_tuple(outputs.detach()), labels.to(DEV))
    lf.use_l1 = False
    assert lf(synth.outputs_train_tuple(outputs.detach()), labels.to(DEV))[4] == 0.0


def test_loss_l1_full_batch_vs_oracle():
    """BASELINE config-2 loss sizes (B=20, 8400 anchors) with an image without labels (the reference itself cannot
    run that case under use_l1: its (0,50) placeholder does not concatenate with (n,26) targets, losses.py:215,278)."""
    from ep24 import loss as L
    from oracle.loss import LossOracle
    B = 20
    counts = [10] * B
    counts[7] = 0
    labels = synth.make_labels(B, counts, seed=131)
    raw = synth.make_raw_head(B, seed=132)
    origin, a0 = [], 0
    for s in synth.STRIDES:
        n = (640 // s) ** 2
        origin.append(raw[:, a0:a0 + n, :26].clone())
        a0 += n
    o_out = synth.decode_head(raw).requires_grad_(True)
    o_or = [o.clone().requires_grad_(True) for o in origin]
    tup5 = list(synth.outputs_train_tuple(o_out))
    tup5[4] = o_or
    o_tup = LossOracle(80, use_l1=True)(tuple(tup5), labels)
    o_tup[0].backward()
    outputs, origin_d, tup5d = _l1_inputs(labels, raw, origin)
    lf = L.Loss_Function(80)
    lf.use_l1 = True
    tup = lf(tup5d, labels.to(DEV))
    tup[0].backward()
    torch.testing.assert_close(tup[0].detach().cpu(), o_tup[0].detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(tup[4].detach().cpu(), o_tup[4].detach(), rtol=1e-4, atol=1e-6)
    g = torch.cat([o.grad for o in origin_d], 1).cpu()
    og = torch.cat([o.grad for o in o_or], 1)
    assert torch.equal(g != 0, og != 0)
    torch.testing.assert_close(g, og, rtol=1e-6, atol=0)
    assert float(g[7].abs().sum()) == 0.0


@pytest.mark.parametrize("start", [0, 1500])
def test_ema_kernel_bit_exact_vs_golden(golden, start):
    from ep24._lib import call, ptr, stream_ptr
    from oracle import ema as oema
    z = golden("g11_ema_%d" % start)
    w, stat = t(z["w0"]).to(DEV), t(z["stat0"]).to(DEV)
    hp = torch.zeros(8, device=DEV)
    for step in range(4):
        d = oema.decay_at(int(z["start"]) + step + 1, float(z["decay"]))
        call("ema_update", ptr(w), ptr(t(z["w_model%d" % step]).to(DEV)), w.numel(), d, 1.0 - d, None, stream_ptr())
        # the same through the device-resident hyper-parameter block
        call("set_hparams", ptr(hp), 0.0, 0.0, 1.0, d, 1.0 - d, stream_ptr())
        call("ema_update", ptr(stat), ptr(t(z["stat_model%d" % step]).to(DEV)), stat.numel(), 0.0, 0.0, ptr(hp), stream_ptr())
        assert torch.equal(w.cpu(), t(z["w_ema%d" % step]))
        assert torch.equal(stat.cpu(), t(z["stat_ema%d" % step]))


def test_sgd_with_device_hyperparameters_and_fused_ema():
    from ep24._lib import call, ptr, stream_ptr
    n = 4099 + 1
    g = torch.Generator().manual_seed(140)
    p0, gr, buf0, e0 = [torch.randn(n, generator=g).to(DEV) for _ in range(4)]
    lr, mom, gs, d = 0.0123, 0.9, 0.5, 0.99871
    first = torch.zeros(1, dtype=torch.int32, device=DEV)
    pa, ba = p0.clone(), buf0.clone()
    call("sgd_nesterov", ptr(pa), ptr(gr), ptr(ba), n, lr, mom, gs, ptr(first), stream_ptr())
    ea = e0.clone()
    call("ema_update", ptr(ea), ptr(pa), n, d, 1.0 - d, None, stream_ptr())
    hp = torch.zeros(8, device=DEV)
    call("set_hparams", ptr(hp), lr, mom, gs, d, 1.0 - d, stream_ptr())
    pb, bb, eb = p0.clone(), buf0.clone(), e0.clone()
    call("sgd_nesterov_hp", ptr(pb), ptr(gr), ptr(bb), n, ptr(hp), ptr(first), ptr(eb), stream_ptr())
    assert torch.equal(pa, pb) and torch.equal(ba, bb) and torch.equal(ea, eb)
    want = (e0.cpu() * torch.tensor(d).float()) + torch.tensor(1.0 - d).float() * pa.cpu()
    assert torch.equal(ea.cpu(), want)
    pc, bc = p0.clone(), buf0.clone()
    call("sgd_nesterov_hp", ptr(pc), ptr(gr), ptr(bc), n, ptr(hp), ptr(first), None, stream_ptr())
    assert torch.equal(pa, pc)


def test_model_ema_follows_the_reference_update():
    """ModelEMA over a real model: every floating-point state_dict entry (parameters AND BatchNorm running statistics)
    follows the oracle bit for bit, integer entries keep the copy's values, the copy is a separate eval-mode model."""
    from ep24 import loss as eloss, train as etrain
    from ep24.ema import ModelEMA
    from oracle import ema as oema
    from test_gpu_engine import tiny_model
    torch.manual_seed(0)
    m = tiny_model()
    B, S = 2, 64
    images = synth.make_images(B, S, seed=1).to(DEV)
    labels = synth.make_labels(B, [2, 1], size=S, seed=2).to(DEV)
    lf = eloss.Loss_Function(80)
    lf.draw = False
    opt = etrain.SGD(m.parameters(), lr=0.01, momentum=0.9, nesterov=True, model=m)
    ema = ModelEMA(m, decay=0.9998, updates=10)
    assert ema.ema is not m and not ema.ema.training and m.training
    assert all(not p.requires_grad for p in ema.ema.parameters())
    state = {k: v.detach().cpu().clone() for k, v in ema.ema.state_dict().items()}
    assert all(torch.equal(v, m.state_dict()[k].cpu()) for k, v in state.items())
    updates = 10
    for _ in range(3):
        opt.zero_grad()
        lf(m(images, train=True), labels)[0].backward()
        opt.step()
        ema.update(m)
        updates = oema.update(state, {k: v.detach().cpu() for k, v in m.state_dict().items()}, updates, 0.9998)
    got = ema.ema.state_dict()
    assert ema.updates == updates == 13
    assert set(got) == set(state)
    for k, v in state.items():
        assert torch.equal(got[k].cpu(), v), k
    bn = "backbone.backbone.stem.conv.bn."
    assert int(got[bn + "num_batches_tracked"]) == 0 and int(m.state_dict()[bn + "num_batches_tracked"]) == 3
    assert not torch.equal(got[bn + "running_mean"].cpu(), m.state_dict()[bn + "running_mean"].cpu())
    # the EMA copy runs the inference path on its own buffers
    out = ema.ema(images, train=False)
    assert out.shape == (B, 84, 107) and bool(torch.isfinite(out[..., 26:]).all())   # (an untrained net's exp() radii may overflow)


@pytest.mark.parametrize("graph_backward", [False, True])
def test_captured_step_with_schedule_ema_and_l1(graph_backward):
    from ep24 import loss as eloss, train as etrain
    from ep24.ema import ModelEMA
    from ep24.schedule import LRScheduler
    from oracle import ema as oema
    from test_gpu_engine import tiny_model
    torch.manual_seed(0)
    ma, mb = tiny_model(), tiny_model()
    mb.load_state_dict(ma.state_dict())
    B, S = 4, 128
    images = synth.make_images(B, S, seed=1).to(DEV)
    labels = synth.make_labels(B, [3, 0, 5, 2], size=S, seed=2).to(DEV)
    sch = LRScheduler("yoloxwarmcos", 0.01, 2, 4, warmup_epochs=1, no_aug_epochs=1, min_lr_ratio=0.05)
    # a) the reference-style loop through the drop-in API
    lf_a = eloss.Loss_Function(80)
    lf_a.draw = False
    opt = etrain.SGD(ma.parameters(), lr=0.01, momentum=0.9, nesterov=True, model=ma)
    ema_a = ModelEMA(ma)
    # b) the captured step
    lf_b = eloss.Loss_Function(80)
    ema_b = ModelEMA(mb)
    ts = etrain.TrainStep(mb, lf_b, lr=0.01, momentum=0.9, batch=B, size=S, graph_backward=graph_backward, ema=ema_b)
    state = {k: v.detach().cpu().clone() for k, v in ema_b.ema.state_dict().items()}
    updates, upd_graph = 0, None
    for it in range(1, 7):
        l1 = it >= 4                                             # "L1_epoch" reached: both paths switch the branch on
        lr = sch.update_lr(it)
        ma.head.use_l1 = lf_a.use_l1 = l1
        for g in opt.param_groups:
            g["lr"] = lr
        opt.zero_grad()
        tup = lf_a(ma(images, train=True), labels)
        tup[0].backward()
        opt.step()
        ema_a.update(ma)
        ts.set_use_l1(l1)
        ts.set_lr(lr)
        before = ts.home.flat.clone()
        res = ts.step(images, labels)
        torch.cuda.synchronize()
        # same graphs as long as only the learning rate / EMA decay change; a new capture when the L1 branch flips
        if it in (2, 3, 5, 6):
            assert ts.g_upd is upd_graph
        upd_graph = ts.g_upd
        assert (float(res[56]) > 0) == l1 and (float(tup[4]) > 0 if l1 else tup[4] == 0.0)
        if it == 1:
            assert float(res[0]) == float(tup[0])                # the forward pass is bitwise reproducible
        np.testing.assert_allclose(float(res[0]), float(tup[0]), rtol=5e-2)
        if l1:
            np.testing.assert_allclose(float(res[56]), float(tup[4]), rtol=5e-2)
        assert (lr == 0.0) == bool(torch.equal(before, ts.home.flat))
        # the fused EMA equals the oracle applied to this path's own parameters, bit for bit
        updates = oema.update(state, {k: v.detach().cpu() for k, v in mb.state_dict().items()}, updates)
        got = ema_b.ema.state_dict()
        for k, v in state.items():
            assert torch.equal(got[k].cpu(), v), (it, k)
    assert ema_a.updates == ema_b.updates == 6
    ts.set_lr(0.0)                                               # read from the device block at replay: nothing moves
    before = ts.home.flat.clone()
    ts.step(images, labels)
    assert torch.equal(before, ts.home.flat) and ts.g_upd is upd_graph
    pa = torch.cat([p.detach().reshape(-1) for p in ema_a.ema.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in ema_b.ema.parameters()])
    assert float((pa - pb).abs().max()) < 1e-3 * float(pa.abs().max())
