"""CPU restatement of the reference's inference post-processing - TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

``postprocess`` follows yolox_24p/utils/boxes.py:29-99 line by line.  The reference delegates the suppression itself
to ``torchvision.ops.nms`` / ``batched_nms``; torchvision is not installed in this image and is not vendored by the
reference, so ``nms`` below restates torchvision's published algorithm (boxes sorted by score descending, a box is
dropped when its IoU with an earlier kept box is > threshold, IoU = inter / (area_a + area_b - inter) with areas
(x2-x1)*(y2-y1); batched_nms suppresses within a class only).  Pinning: tests/golden/g9_postprocess.npz is the
REFERENCE's postprocess run with this ``nms`` installed as the stand-in for the missing torchvision ops (the same
device used for cv2.resize in the sector warp) - everything except the suppression loop itself is therefore pinned
by the reference; the loop is "parity unpinned" and checked only against brute-force properties in the tests.
"""
import numpy as np
import torch


def nms(boxes, scores, thr):
    """Kept indices in decreasing score order (ties: lower index first)."""
    s = scores.detach().cpu().numpy()
    order = np.lexsort((np.arange(len(s)), -s))
    keep, dead = [], np.zeros(len(s), bool)
    bf = boxes.detach().cpu().float().numpy()
    areaf = (bf[:, 2] - bf[:, 0]) * (bf[:, 3] - bf[:, 1])
    for pos, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        iw = np.maximum(np.minimum(bf[i, 2], bf[rest, 2]) - np.maximum(bf[i, 0], bf[rest, 0]), np.float32(0))
        ih = np.maximum(np.minimum(bf[i, 3], bf[rest, 3]) - np.maximum(bf[i, 1], bf[rest, 1]), np.float32(0))
        inter = iw * ih
        iou = inter / (areaf[i] + areaf[rest] - inter)
        dead[rest[iou > np.float32(thr)]] = True
    return torch.tensor(keep, dtype=torch.int64)


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.batched_nms: suppression only inside a class; result in decreasing score order."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    keep = []
    for c in torch.unique(idxs):
        m = torch.nonzero(idxs == c).reshape(-1)
        keep.append(m[nms(boxes[m], scores[m], thr)])
    keep = torch.cat(keep)
    s = scores[keep].detach().cpu().numpy()
    return keep[torch.from_numpy(np.lexsort((keep.numpy(), -s)))]


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """boxes.py:29-99.  Note theta_all * cos(theta_all): the angle itself multiplies the cosine (as written there)."""
    theta = torch.tensor(15 * np.pi / 180)
    theta_all = torch.arange(24) * theta
    cos_t = theta_all * torch.cos(theta_all)
    sin_t = theta_all * torch.sin(theta_all)
    output = [None for _ in range(len(prediction))]
    for i, image_pred in enumerate(prediction):
        if not image_pred.size(0):
            continue
        class_conf, class_pred = torch.max(image_pred[:, 27:27 + num_classes], 1, keepdim=True)
        conf_mask = (image_pred[:, 26] * class_conf.squeeze() >= conf_thre).squeeze()
        det = torch.cat((image_pred[:, :27], class_conf, class_pred.float()), 1)[conf_mask]
        if not det.size(0):
            continue
        px = det[:, 2:26] * cos_t[None] + det[:, 0:1]
        py = det[:, 2:26] * sin_t[None] + det[:, 1:2]
        rect = torch.stack((px.min(1).values, py.min(1).values, px.max(1).values, py.max(1).values), 1)
        sc = det[:, 26] * det[:, 27]
        keep = nms(rect, sc, nms_thre) if class_agnostic else batched_nms(rect, sc, det[:, 28], nms_thre)
        output[i] = det[keep]
    return output
