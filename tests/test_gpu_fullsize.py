"""Full-size runs of the BASELINE configurations that round 2 only ran small.

* config 4 (backbone swap, resnet50 / densenet121 behind the unchanged neck and head): one captured training step each at
  B = 20, 640 x 640 with the property set of ``test_full_size_step_properties`` - two independent runs bitwise identical, a
  step with lr = 0 leaves the weights untouched, gradients finite and reaching every parameter the graph uses.
* config 5 (B = 8, 1280 x 1280, 50 GTs per image, the sector warp in the loop): one step whose input is produced by the public batch
  warp + letterbox API inside the step; finite, deterministic, SimOTA matches present on every image.
* the bf16 end-to-end bridge: YOLOX-l at B = 2, 640 x 640, the product's bf16 forward against the bf16-STORAGE-emulating oracle
  (oracle/model.py EMULATE_BF16) chained through the whole network - the gap between "the fp32 parity mode equals the reference"
  (tests/test_gpu_fp32.py) and "every bf16 unit equals the bf16 restatement on the plan's own inputs" (tests/test_gpu_engine.py).
"""
import numpy as np
import pytest
import torch

from ep24 import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(backbone):
    from ep24 import nn as enn
    torch.manual_seed(0)
    m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0, backbone_type=backbone), enn.YOLOXHead(80, 1.0))
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    m.head.initialize_biases(1e-2)
    return m.to(DEV)


def _run(backbone, lr, steps, batch=20, size=640, gts=10):
    from ep24 import loss as eloss, train as etrain
    m = _model(backbone)
    ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=lr, momentum=0.9, batch=batch, size=size)
    if getattr(ts.eng, "drop_keep", None) is not None:
        ts.eng.fixed_dropout = True                      # DenseNet: the same Dropout2d draws in both runs (device RNG otherwise)
        g = torch.Generator().manual_seed(11)
        ts.eng.drop_keep.copy_((torch.rand(ts.eng.drop_keep.shape, generator=g) > 0.3).float().div(0.7).to(DEV))
    ts.eng.images.copy_(synth.make_images(batch, size, seed=1).to(DEV))
    ts.labels.copy_(synth.make_labels(batch, gts, size=size, seed=1000).to(DEV))
    losses = [float(ts.step()[0]) for _ in range(steps)]
    torch.cuda.synchronize()
    return losses, ts.home.flat.clone(), ts.home.gflat.clone(), m, ts


@pytest.mark.parametrize("backbone", ["resnet", "densenet"])
def test_config4_full_size_step_properties(backbone):
    la, wa, ga, ma, _ = _run(backbone, 0.001, 2)
    lb, wb, gb, _, _ = _run(backbone, 0.001, 2)
    assert la == lb and torch.equal(wa, wb) and torch.equal(ga, gb), (la, lb)        # bitwise deterministic, as the CSPDarknet step
    assert all(np.isfinite(la)) and la[0] != la[1]
    assert bool(torch.isfinite(ga).all())
    used = 0
    for n, p in ma.named_parameters():                   # the gradient reaches every parameter the graph uses
        if backbone == "resnet" and (".fc." in n or "baseconv" in n):
            assert float(p.grad.abs().sum()) == 0, n      # in the state dict, never run by the reference (darknet.py:311-330)
            continue
        assert float(p.grad.abs().sum()) > 0, n
        used += 1
    assert used > 200
    l0, w0, _, m0, _ = _run(backbone, 0.0, 1)
    ref = _model(backbone)
    assert l0[0] == la[0]
    for (n, p_new), p_ref in zip(m0.named_parameters(), ref.parameters()):           # lr = 0 changes nothing
        assert torch.equal(p_new.detach(), p_ref.detach()), n


def test_config5_full_size_step_with_the_warp_in_the_loop():
    """B = 8, 1280 x 1280, 50 GTs per image; every step first warps its uint8 source images with the sector warp on the GPU (public
    batch API, Theta 30 .. 90) and letterboxes the results into the network input, then runs the captured step."""
    from ep24 import input as ein, loss as eloss, sector as esec, train as etrain
    B, S = 8, 1280

    def run():
        m = _model("darknet")
        ts = etrain.TrainStep(m, eloss.Loss_Function(80), lr=0.001, momentum=0.9, batch=B, size=S)
        ts.labels.copy_(synth.make_labels(B, 50, size=S, seed=1000).to(DEV))
        dist_op = esec.Image_Distortion(DEV)
        g8 = torch.Generator().manual_seed(77)
        src = [(torch.rand(S, S, 3, generator=g8) * 255).to(torch.uint8).to(DEV) for _ in range(B)]
        msk = [torch.zeros(S, S, 3, dtype=torch.uint8, device=DEV) for _ in range(B)]
        for mk in msk:
            mk[S // 4: S // 2, S // 4: S // 2] = 255
        thetas = [30 + 60 * i // (B - 1) for i in range(B)]
        out = []
        for _ in range(2):
            warped, _m, boxes = dist_op.distort_batch(src, msk, thetas)
            ein.preproc_batch(warped, (S, S), device=DEV, out=ts.eng.images)
            out.append(float(ts.step()[0]))
        torch.cuda.synchronize()
        img_sum = float(ts.eng.images.double().sum())
        fg = (ts.ws.matched_gt >= 0).sum(1).cpu()
        return out, ts.home.flat.clone(), img_sum, fg, ts

    la, wa, sa, fg, ts = run()
    lb, wb, sb, _, _ = run()
    assert la == lb and torch.equal(wa, wb) and sa == sb                      # the warp, the letterbox and the step are deterministic
    assert all(np.isfinite(la)) and la[0] != la[1]
    assert ts.eng.A == 33600 and tuple(ts.eng.outputs.shape) == (B, 33600, 107)
    assert int(fg.min()) >= 50 and int(fg.max()) < 33600                      # every image has at least one anchor per GT
    assert bool(torch.isfinite(ts.home.gflat).all()) and float((ts.home.gflat != 0).float().mean()) > 0.99
    assert abs(sa) > 0                                                         # the input really is the warped, letterboxed batch


def test_bf16_product_forward_vs_bf16_emulating_oracle_stage_by_stage():
    """The bridge between the timed bf16 kernels and the restatement, as a MEASUREMENT with bounds that are not read off its own output
    (tests/bridge_stages.py has the model).  YOLOX-l at the configuration's batch, B = 20, 640 x 640 (every
    BatchNorm sees >= 8 000 values per channel), product against the oracle in its bf16-STORAGE-emulating mode:

      * teacher-forced, stage by stage (stem, dark2 .. dark5, the two halves of the PAFPN, the head; each product stage on the
        ORACLE's inputs): relative rms error <= 2 x 2.3e-3 x sqrt(conv units of the stage) - two bf16 stores per unit, fully
        decorrelated roundings, quadrature - i.e. 4.6e-3 (stem) .. 2.2e-2 (22 units);
      * chained (each product stage on the product's own inputs): <= that bound + 2 x what the ORACLE stage itself makes of random
        input perturbations of the size of the chained input errors;
      * the loss two ways: the product's loss kernels against the oracle's loss on the product's own head outputs (1e-4), and against
        the oracle's loss on the oracle's outputs (2 %: different inputs, SimOTA's matching is discrete).
    The table is printed (pytest -s) and kept in profiles/r04_bridge.txt."""
    import bridge_stages as bs
    from ep24 import loss as eloss
    from oracle.loss import LossOracle
    ref, m = bs.build_pair(seed=3, dev=DEV)
    B, S = 20, 640
    x = synth.make_images(B, S, seed=9)
    labels = synth.make_labels(B, 10, size=S, seed=2)
    rows, head_out, head_ref = bs.bridge_table(ref, m, x, dev=DEV)
    print("\nstage            units | teacher-forced rms   bound | chained rms   propagated   bound")
    bad = []
    for stage, n, tf_rms, tf_max, ch_rms, ch_max, prop in rows:
        b_tf = bs.rms_bound(stage)
        b_ch = b_tf + bs.PROP_SLACK * (prop or 0.0)
        print("%-16s %5d | %18.3e %7.1e | %11.3e %12s %7.1e" % (stage, n, tf_rms, b_tf, ch_rms, "-" if prop is None else "%.3e" % prop, b_ch))
        if not tf_rms <= b_tf:
            bad.append((stage, "teacher-forced", tf_rms, b_tf))
        if not ch_rms <= b_ch:
            bad.append((stage, "chained", ch_rms, b_ch))
    assert not bad, bad
    # the loss on the product's own head outputs, and on the oracle's
    out = m(x.to(DEV), train=True)
    got = out[3].detach().float().cpu()
    assert torch.equal(got, head_out.float().cpu())                                # the stage-wise chain IS the model's forward
    loss_got = float(eloss.Loss_Function(80).forward(out, labels.to(DEV))[0])
    loss_same_inputs = float(LossOracle(80)(synth.outputs_train_tuple(got.clone(), size=S), labels)[0])
    want_tuple = synth.outputs_train_tuple(head_ref.float(), size=S)
    loss_want = float(LossOracle(80)(want_tuple, labels)[0])
    print("loss: product %.5f, oracle on the product's outputs %.5f, oracle on its own outputs %.5f" % (loss_got, loss_same_inputs, loss_want))
    assert abs(loss_got - loss_same_inputs) < 1e-4 * abs(loss_same_inputs), (loss_got, loss_same_inputs)
    assert abs(loss_got - loss_want) < 2e-2 * abs(loss_want), (loss_got, loss_want)
