"""Inference-side mirror of the reference: ``postprocess`` (yolox_24p/utils/boxes.py:29-99).

The eval-mode network itself is ``model.eval(); model(images, train=False)`` (ep24.nn / ep24.engine.forward_eval):
decoded predictions ``[B, A, 27 + C]`` with sigmoid objectness / class scores, as ``YOLOXHead`` returns them with
``decode_in_inference`` (yolo_head_24p.py:190-210, 239-256).
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


class _Scratch:
    def __init__(self, B, A, dev):
        self.key = (B, A, str(dev))
        P = 1
        while P < A:
            P <<= 1
        self.P = P
        f = dict(device=dev)
        self.score = torch.empty(B * A, dtype=torch.float32, **f)
        self.conf = torch.empty(B * A, dtype=torch.float32, **f)
        self.cls = torch.empty(B * A, dtype=torch.int32, **f)
        self.rect = torch.empty(B * A * 4, dtype=torch.float32, **f)
        self.skey = torch.empty(B * P, dtype=torch.float32, **f)
        self.sidx = torch.empty(B * P, dtype=torch.int32, **f)
        self.dead = torch.empty(B * P, dtype=torch.uint8, **f)
        self.keep = torch.empty(B * A, dtype=torch.int32, **f)
        self.count = torch.zeros(B, dtype=torch.int32, **f)
        theta = torch.arange(24) * torch.tensor(15 * 3.141592653589793 / 180)          # boxes.py:31-33, fp32
        self.ray = torch.cat((theta * torch.cos(theta), theta * torch.sin(theta))).float().to(dev)


_scratch = {}


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """``prediction [B, A, 27 + C]`` (decoded, sigmoid scores) -> list of B entries: ``None`` (nothing kept) or
    ``[n, 29]`` = (cx, cy, 24 radii, obj_conf, class_conf, class_pred) in NMS order.  Three launches + one small D2H copy
    of the per-image counts; rectangle / score / NMS semantics are the reference's (torchvision batched_nms)."""
    _lib.require_gpu()
    if not prediction.is_cuda:
        raise _lib.Ep24Error("ep24: predictions must live on the GPU (no CPU fallback on the product path)")
    if prediction.dim() != 3 or prediction.shape[2] != 27 + num_classes:
        raise IndexError("expected predictions [B, A, 27 + %d], got %s" % (num_classes, tuple(prediction.shape)))
    B, A, ncols = prediction.shape
    output = [None for _ in range(B)]
    if A == 0 or B == 0:
        return output
    pred = prediction.detach().float().contiguous()
    key = (B, A, str(pred.device))
    ws = _scratch.get(key)
    if ws is None:
        ws = _scratch[key] = _Scratch(B, A, pred.device)
    s = stream_ptr()
    call("post_prepare", ptr(pred), ncols, num_classes, B * A, float(conf_thre), ptr(ws.ray), ptr(ws.score), ptr(ws.conf),
         ptr(ws.cls), ptr(ws.rect), s)
    call("post_nms", ptr(ws.score), ptr(ws.cls), ptr(ws.rect), B, A, float(nms_thre), 1 if class_agnostic else 0, ptr(ws.skey),
         ptr(ws.sidx), ptr(ws.dead), ptr(ws.keep), ptr(ws.count), ws.P, s)
    counts = ws.count.tolist()                                   # the API returns per-image tensors: one sync
    for b, n in enumerate(counts):
        if n == 0:
            continue
        det = torch.empty(n, 29, dtype=torch.float32, device=pred.device)
        call("post_gather", ptr(pred, b * A * ncols), ncols, ptr(ws.conf, b * A), ptr(ws.cls, b * A), ptr(ws.keep, b * A), n,
             ptr(det), s)
        output[b] = det
    return output
