"""Oracle vs golden vectors generated from the reference: candidate masks G4, assignment G5, loss G6."""
import numpy as np
import pytest
import torch

from conftest import t
from ep24 import synth
from oracle import assign
from oracle.loss import LossOracle


@pytest.mark.parametrize("tag", ["convex", "star"])
def test_g4_masks(golden, tag):
    z = golden("g4_masks_" + tag)
    lab = synth.make_labels(1, int(z["G"]), seed=int(z["label_seed"]), star=bool(z["star"]))[0, :int(z["G"]), 1:]
    assert float(lab.double().sum()) == float(z["lab_sum"])
    xs, ys, ss = synth.anchor_grid()
    fg, both, in_box, _ = assign.candidate_masks(lab, xs, ys, ss)
    assert torch.equal(fg, t(z["fg"]))
    assert torch.equal(both, t(z["in_both"]))
    assert torch.equal(in_box, t(z["in_box"]))
    assert float(z["min_margin"]) > 1e-3          # vector is not sitting on the 350-degree threshold


def _g6_inputs(z):
    B = int(z["B"])
    labels = synth.make_labels(B, [int(c) for c in z["counts"]], seed=int(z["label_seed"]))
    raw = synth.make_raw_head(B, seed=int(z["head_seed"]))
    assert float(labels.double().sum()) == float(z["labels_sum"])
    assert float(raw.double().sum()) == float(z["raw_sum"])
    return B, labels, raw


def test_g6_loss_two_calls(golden):
    z = golden("g6_loss")
    B, labels, raw = _g6_inputs(z)
    lf = LossOracle(80)
    for call in range(2):
        p = "c%d_" % call
        outputs = synth.decode_head(raw if call == 0 else raw * 0.98 + 0.01).requires_grad_(True)
        tup = lf(synth.outputs_train_tuple(outputs), labels)
        tup[0].backward()
        # assignment: bit-exact indices
        for b in range(B):
            if int(z["counts"][b]) == 0:
                assert lf.trace[b] is None
                continue
            cls_m, fg, ious, gt_idx, nfg = lf.trace[b]
            assert nfg == int(z[p + "img%d_nfg" % b])
            assert torch.equal(fg, t(z[p + "img%d_fg" % b]))
            assert torch.equal(gt_idx, t(z[p + "img%d_gt" % b]))
            assert torch.equal(cls_m, t(z[p + "img%d_cls" % b]))
            assert torch.equal(ious, t(z[p + "img%d_pious" % b]))
        # losses
        for k, v in (("loss", tup[0]), ("loss_iou_w", tup[1]), ("loss_obj", tup[2]), ("loss_cls", tup[3]),
                     ("reg_w", tup[6][3]), ("obj_w", tup[6][4]), ("cls_w", tup[6][5]),
                     ("draw_cx", tup[6][0]), ("draw_r", tup[6][2])):
            torch.testing.assert_close(v.detach(), t(z[p + k]), rtol=1e-6, atol=1e-6)
        assert tup[4] == float(z[p + "loss_l1"]) and tup[5] == float(z[p + "fg_per_gt"])
        g = outputs.grad
        rows = t(z[p + "grad_rows"])
        torch.testing.assert_close(g.reshape(-1, g.shape[-1])[rows], t(z[p + "grad_vals"]), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(g[..., 26].reshape(-1)[::5], t(z[p + "grad_obj"]), rtol=1e-5, atol=1e-9)
        assert abs(float(g.double().abs().sum()) - float(z[p + "grad_abs_sum"])) < 1e-5 * float(z[p + "grad_abs_sum"])
    # the weights are stateful: second call differs from a fresh instance
    assert not np.allclose(z["c0_reg_w"], z["c1_reg_w"])


def test_g5_assign_g50(golden):
    z = golden("g5_assign_g50")
    labels = synth.make_labels(1, 50, seed=int(z["label_seed"]))
    outputs = synth.decode_head(synth.make_raw_head(1, seed=int(z["head_seed"])))
    lf = LossOracle(80)
    tup = lf(synth.outputs_train_tuple(outputs), labels)
    cls_m, fg, ious, gt_idx, nfg = lf.trace[0]
    assert nfg == int(z["nfg"])
    assert torch.equal(fg, t(z["fg"])) and torch.equal(gt_idx, t(z["gt"])) and torch.equal(cls_m, t(z["cls"]))
    assert torch.equal(ious, t(z["pious"]))
    torch.testing.assert_close(tup[0], t(z["loss"]), rtol=1e-6, atol=1e-6)
