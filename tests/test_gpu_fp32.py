"""End-to-end fp32 parity: images in -> head outputs, SimOTA assignment, loss and parameter gradients out, against the
CPU oracle / the reference-generated golden model.

The reference trains in fp32 with no AMP (yolox_24p/train_24p.py:86-104, models/network_blocks.py:50-51) and north_star
asks for "bit-exact SimOTA assignment indices, fp32 loss within 1e-4 relative on identical synthetic 640x640 inputs".  The
product path stores activations in bf16, which a random-init deep BatchNorm network amplifies to ~30 % rms at the head,
so that sentence is checked in the engine's fp32 PARITY MODE: the same launch plan (buffers, concat slots, residual
aliasing, accumulate flags, flat parameters, merged units) on fp32 activations through the ep24_f32_* entry points.
Tolerances: 1e-4 relative on outputs and loss, indices exact, gradients 2e-3 of the tensor's largest magnitude."""
import numpy as np
import pytest
import torch

from conftest import t
from ep24 import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


def test_tiny_model_vs_reference_golden_fp32(golden):
    """G7 `g7_model_tiny` = the reference's own YOLOX (width 0.125, depth 0.33, 64x64, B = 2): outputs, every parameter
    gradient and the running statistics, now to fp32 accuracy (the bf16 path only asserts direction: cos > 0.99)."""
    from ep24 import nn as enn
    z = golden("g7_model_tiny")
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125))
    m.load_state_dict({k[2:]: t(z[k]) for k in z.files if k.startswith("w:")}, strict=True)
    m.to(DEV).set_compute_dtype(torch.float32)
    x = t(z["x"]).to(DEV)
    xs, ys, ss, out, extra = m(x, train=True)
    want = t(z["out"])
    assert out.shape == want.shape
    # 64x64 input: the 32-stride level is 2x2, its BatchNorms normalise over 8 values and amplify rounding differences ~100x;
    # the tolerances here are 5e-4, the 640x640 test below holds 1e-4
    torch.testing.assert_close(out.cpu()[..., :2], want[..., :2], rtol=5e-4, atol=2e-3)       # centres, pixels
    torch.testing.assert_close(out.cpu()[..., 2:26], want[..., 2:26], rtol=5e-4, atol=1e-4)   # radii
    torch.testing.assert_close(out.cpu()[..., 26:], want[..., 26:], rtol=5e-4, atol=5e-4)     # logits
    out.backward(t(z["gy"]).to(DEV))
    params = dict(m.named_parameters())
    worst = 0.0
    for k in z.files:
        if k.startswith("g:"):
            g, w = params[k[2:]].grad, t(z[k])
            assert g.shape == w.shape
            e = rel_err(g, w)
            worst = max(worst, e)
            assert e < 2e-3, (k, e)
    sd = m.state_dict()
    assert rel_err(sd["backbone.backbone.stem.conv.bn.running_mean"], t(z["after:stem_rm"])) < 1e-5
    assert rel_err(sd["backbone.backbone.stem.conv.bn.running_var"], t(z["after:stem_rv"])) < 1e-5
    print("worst gradient rel err", worst)


def _paired(depth, width, seed=3):
    from oracle import model as om
    from ep24 import nn as enn
    torch.manual_seed(seed)
    ref = om.Net(depth, width)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.uniform_(mod.bias, -0.2, 0.2)
            mod.eps, mod.momentum = 1e-3, 0.03
    m = enn.YOLOX(enn.YOLOPAFPN(depth, width), enn.YOLOXHead(80, width))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    m.load_state_dict(ref.state_dict(), strict=True)
    return ref, m.to(DEV).set_compute_dtype(torch.float32)


# (depth, width, B, S, GTs per image, tolerance scale).  The last case is BASELINE config 1 (YOLOX-l 24p forward + circle_inter
# loss on 1x3x640x640 synthetic, the reference's CPU plumbing run) with the HIP path in the fp32 parity mode on the other side:
# 131 conv layers with batch statistics over ONE image amplify fp32 rounding differences (torch's own fp32 conv vs a
# double-accumulated one) about 30x more than the batch-2 half-width network, so its output tolerances are scaled; the
# assignment indices and the 1e-4 loss bound are the same for every case.
CASES = [(0.33, 0.5, 2, 640, [6, 3], 1.0), (0.33, 0.25, 3, 320, [4, 0, 9], 1.0), (1.0, 1.0, 1, 640, [5], 50.0)]


@pytest.mark.parametrize("depth,width,B,S,gts,scale", CASES)
def test_images_to_assignment_loss_and_gradients_fp32(depth, width, B, S, gts, scale):
    """The north-star parity sentence: identical synthetic inputs through the network, SimOTA and the 24-circle loss."""
    from ep24 import loss as eloss
    from oracle.loss import LossOracle
    ref, m = _paired(depth, width)
    images = synth.make_images(B, S, seed=5)
    labels = synth.make_labels(B, gts, size=S, seed=6)
    # ---- CPU oracle (fp32, the reference's algorithm)
    ref.train()
    o_in = ref(images, train=True)
    ora = LossOracle(80)
    o_tup = ora(o_in, labels)
    o_tup[0].backward()
    # ---- HIP path, fp32 parity mode
    lf = eloss.Loss_Function(80)
    tup_in = m(images.to(DEV), train=True)
    tup = lf(tup_in, labels.to(DEV))
    tup[0].backward()
    torch.cuda.synchronize()
    out, want = tup_in[3].detach().cpu(), o_in[3].detach()
    loss, oloss = float(tup[0].detach()), float(o_tup[0].detach())
    rp = dict(ref.named_parameters())
    gerr = {k: rel_err(p.grad, rp[k].grad) for k, p in m.named_parameters()}
    worst = max(gerr, key=gerr.get)
    print("centres max abs diff %.3g px, radii max rel %.3g, logits max abs %.3g, loss %.7g vs %.7g (rel %.3g), worst gradient %s %.3g"
          % (float((out[..., :2] - want[..., :2]).abs().max()), float(((out[..., 2:26] - want[..., 2:26]).abs() / want[..., 2:26].abs()).max()),
             float((out[..., 26:] - want[..., 26:]).abs().max()), loss, oloss, abs(loss - oloss) / abs(oloss), worst, gerr[worst]))
    torch.testing.assert_close(out[..., :2], want[..., :2], rtol=1e-4 * scale, atol=1e-3 * scale)          # centres (pixels)
    torch.testing.assert_close(out[..., 2:26], want[..., 2:26], rtol=2e-4 * scale, atol=1e-4 * scale)      # radii = exp(t) * stride
    torch.testing.assert_close(out[..., 26:], want[..., 26:], rtol=1e-4 * scale, atol=2e-4 * scale)        # logits
    # SimOTA: foreground masks, matched ground truths and classes identical, image by image
    for b in range(B):
        o = ora.trace[b]
        cls_m, fg, ious, gt_idx, nfg = lf.assignment_of(labels, b)
        if o is None:
            assert nfg == 0
            continue
        assert nfg == o[4], (b, nfg, o[4])
        assert torch.equal(fg.cpu(), o[1]) and torch.equal(gt_idx.cpu(), o[3]) and torch.equal(cls_m.cpu().long(), o[0].long())
    # loss: total within 1e-4 relative (north_star), the three groups within 2e-4
    assert abs(loss - oloss) <= 1e-4 * abs(oloss), (loss, oloss)
    torch.testing.assert_close(tup[1].detach().cpu(), o_tup[1].detach(), rtol=2e-4 * scale ** 0.5, atol=1e-6)
    for i in (2, 3):
        assert abs(float(tup[i].detach()) - float(o_tup[i].detach())) <= 2e-4 * scale ** 0.5 * abs(float(o_tup[i].detach())) + 1e-7
    # gradients of every parameter and the BatchNorm running statistics
    for k, e in gerr.items():
        assert e < 2e-3 * scale ** 0.5, (k, e)
    rsd = ref.state_dict()
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v, rsd[k]) < 1e-4, k
        if "num_batches" in k:
            assert int(v) == int(rsd[k])


def test_fp32_mode_shares_parameters_with_the_bf16_plan():
    """One parameter home: switching the compute dtype changes the plan, not the weights; the bf16 outputs stay within the
    bf16-storage noise of the fp32 ones on a shallow network (a sanity link between the two modes)."""
    from ep24 import nn as enn
    torch.manual_seed(0)
    m = enn.YOLOX(enn.YOLOPAFPN(0.33, 0.125), enn.YOLOXHead(80, 0.125)).to(DEV)
    x = synth.make_images(2, 128, seed=3).to(DEV)
    with torch.no_grad():
        o32 = m.set_compute_dtype(torch.float32)(x, train=True)[3].clone()
        o16 = m.set_compute_dtype(torch.bfloat16)(x, train=True)[3].clone()
    assert m.engine(2, 128).home is m.engine(2, 128, torch.float32).home
    c = torch.nn.functional.cosine_similarity(o32[..., 26:].reshape(1, -1), o16[..., 26:].reshape(1, -1))
    assert float(c) > 0.95                  # measured 0.978 at 64x64 (a 2x2 last level), higher at this size
