"""Oracle postprocess (oracle/post.py) against the reference's outputs (G9) and brute-force NMS properties."""
import numpy as np
import torch

from conftest import t
from ep24 import synth
from oracle import post as opost


def _pred(z):
    raw = synth.make_raw_head(3, seed=int(z["head_seed"]), num_classes=80)
    pred = synth.decode_head(raw)
    g = torch.Generator().manual_seed(int(z["noise_seed"]))
    pred[..., 26:] = torch.sigmoid(raw[..., 26:] + 4.5 + torch.randn(raw[..., 26:].shape, generator=g))
    pred[2, :, 26] = 0.0
    return pred


def test_postprocess_vs_reference(golden):
    for tag in ("a", "b"):
        z = golden("g9_postprocess_" + tag)
        pred = _pred(z)
        outs = opost.postprocess(pred.clone(), 80, 0.7, 0.45, class_agnostic=bool(int(z["agnostic"])))   # whole batch at once
        for i in range(3):
            n = int(z["img%d_n" % i])
            if n < 0:
                assert outs[i] is None
            else:
                assert outs[i].shape == (n, 29)
                got, want = outs[i], t(z["img%d_det" % i])
                keep_cols = [0, 1, 26, 27, 28]
                assert torch.equal(got[:, keep_cols], want[:, keep_cols])
                torch.testing.assert_close(got[:, 2:26], want[:, 2:26], rtol=1e-6, atol=0)   # exp() differs in the last bit between hosts


def test_nms_properties():
    """Kept boxes are mutually below the threshold, every dropped box overlaps an earlier kept one above it, and the
    result is in decreasing score order."""
    g = torch.Generator().manual_seed(5)
    c = torch.rand(400, 2, generator=g) * 100
    wh = torch.rand(400, 2, generator=g) * 30 + 2
    boxes = torch.cat((c - wh / 2, c + wh / 2), 1)
    scores = torch.rand(400, generator=g)
    keep = opost.nms(boxes, scores, 0.45)

    def iou(a, b):
        iw = (torch.minimum(a[2], b[2]) - torch.maximum(a[0], b[0])).clamp(min=0)
        ih = (torch.minimum(a[3], b[3]) - torch.maximum(a[1], b[1])).clamp(min=0)
        inter = iw * ih
        return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)

    ks = keep.tolist()
    assert all(scores[ks[i]] >= scores[ks[i + 1]] for i in range(len(ks) - 1))
    for i in range(len(ks)):
        for j in range(i):
            assert iou(boxes[ks[i]], boxes[ks[j]]) <= 0.45
    kept = set(ks)
    for d in range(400):
        if d not in kept:
            assert any(scores[k] >= scores[d] and iou(boxes[d], boxes[k]) > 0.45 for k in ks)
    assert len(opost.batched_nms(boxes[:0], scores[:0], scores[:0], 0.45)) == 0
