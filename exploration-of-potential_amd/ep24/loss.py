"""Host side of the 24p loss: mirrors ``models.losses.Loss_Function`` / ``IOUloss`` and ``utils.bboxes_iou``
of the reference (yolox_24p/models/losses.py:14-603, yolox_24p/utils/boxes.py:166-243) on top of the C ABI.

Everything between the head outputs and the scalar loss runs as HIP kernels chained on the current stream;
nothing is copied to the host (the reference synchronises >= 3+G times per image).  The per-call stateful
task weights (``last_*`` losses) live in a device buffer.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr, stream_ptr

MAX_GT = 50
NCOLS_BASE = 27
RESULT = 64


class LossWorkspace:
    """Fixed-capacity device buffers for one (B, A, num_classes) problem size."""

    def __init__(self, B, A, num_classes, device):
        self.B, self.A, self.C = B, A, num_classes
        z = dict(device=device)
        self.num_gt = torch.zeros(B, dtype=torch.int32, **z)
        self.masks = torch.zeros(3, B, A, dtype=torch.int64, **z)         # in_box, in_ctr, match (uint64 bit per GT)
        self.pw = torch.zeros(B, MAX_GT, A, dtype=torch.float32, **z)
        self.cost = torch.zeros(B, MAX_GT, A, dtype=torch.float32, **z)
        self.ks = torch.zeros(B, MAX_GT, dtype=torch.int32, **z)
        self.matched_gt = torch.full((B, A), -1, dtype=torch.int32, **z)
        self.matched_iou = torch.zeros(B, A, dtype=torch.float32, **z)
        self.nblocks = _lib.lib().fn["ep24_loss_blocks"](B, A)
        self.partials = torch.zeros(self.nblocks, 32, dtype=torch.float32, **z)
        self.result = torch.zeros(RESULT, dtype=torch.float32, **z)
        self.dout = torch.zeros(B, A, NCOLS_BASE + num_classes, dtype=torch.float32, **z)
        self.d_origin = None                                              # [B,A,26], allocated by the L1 branch

    def l1_grad_buffer(self):
        if self.d_origin is None:
            self.d_origin = torch.zeros(self.B, self.A, 26, dtype=torch.float32, device=self.dout.device)
        return self.d_origin


def assign_candidates(ws, labels, xs, ys, strides):
    """a4+a5: the geometric candidate masks depend on the labels and the anchor grid only, so a captured step runs them
    ahead of the forward pass (ep24.train: first thing on the main lane)."""
    in_box, in_ctr = ws.masks[0], ws.masks[1]
    call("assign_candidates", ptr(labels), ptr(xs), ptr(ys), ptr(strides), ptr(ws.num_gt), ptr(in_box), ptr(in_ctr), ws.B, ws.A,
         stream_ptr())


def assign_cost_range(ws, outputs, labels, a_lo, a_hi):
    """pw / cost rows of anchors [a_lo, a_hi) (candidate masks must be there): what ep24.train runs on the forward lane of a head
    level as soon as that level's outputs exist."""
    call("assign_cost_range", ptr(outputs), NCOLS_BASE + ws.C, ptr(labels), ptr(ws.num_gt), ptr(ws.masks[0]), ptr(ws.masks[1]),
         ptr(ws.pw), ptr(ws.cost), ws.B, ws.A, ws.C, int(a_lo), int(a_hi), stream_ptr())


def assign_and_reduce(ws, outputs, labels, xs, ys, strides, state, origin=None, candidates_done=False, cost_done=()):
    """Kernels a4..a10 forward: fills ws.matched_* and ws.result; updates `state` (device [26]).  ``origin`` [B,A,26]
    (the head's raw regression outputs) switches the L1 branch on (losses.py:197-198, 304-309).  ``cost_done``: anchor ranges
    ``(lo, hi)`` whose pw / cost rows assign_cost_range has written already."""
    B, A, C = ws.B, ws.A, ws.C
    ncols = NCOLS_BASE + C
    s = stream_ptr()
    in_box, in_ctr, match = ws.masks[0], ws.masks[1], ws.masks[2]
    if not candidates_done:
        call("assign_candidates", ptr(labels), ptr(xs), ptr(ys), ptr(strides), ptr(ws.num_gt), ptr(in_box), ptr(in_ctr), B, A, s)
    pos = 0
    for lo, hi in sorted(cost_done) + [(A, A)]:               # the ranges nobody has done
        if lo > pos:
            assign_cost_range(ws, outputs, labels, pos, lo)
        pos = max(pos, hi)
    call("memset_zero", ptr(match), match.numel() * 8, s)
    call("dynamic_k", ptr(ws.pw), ptr(ws.cost), ptr(ws.num_gt), ptr(in_box), ptr(in_ctr), ptr(match), ptr(ws.ks), B, A, s)
    call("assign_resolve", ptr(match), ptr(ws.pw), ptr(ws.cost), ptr(ws.num_gt), ptr(ws.matched_gt), ptr(ws.matched_iou),
         B, A, s)
    call("loss_terms", ptr(outputs), ncols, ptr(labels), ptr(ws.matched_gt), ptr(ws.matched_iou), ptr(ws.partials),
         B, A, C, ptr(origin), ptr(xs), ptr(ys), ptr(strides), s)
    call("loss_finalize", ptr(ws.partials), ws.nblocks, ptr(ws.num_gt), B, ptr(state), ptr(ws.result), s)


def loss_grad(ws, outputs, labels, grad_scale=None, origin=None, grid=None):
    """d loss / d outputs into ws.dout (fully overwritten); with ``origin`` also d loss / d origin into ws.d_origin."""
    d_origin = ws.l1_grad_buffer() if origin is not None else None
    xs, ys, strides = grid if grid is not None else (None, None, None)
    call("loss_grad", ptr(outputs), NCOLS_BASE + ws.C, ptr(labels), ptr(ws.matched_gt), ptr(ws.matched_iou),
         ptr(ws.result), ptr(grad_scale), ptr(ws.dout), ws.B, ws.A, ws.C, ptr(origin), ptr(xs), ptr(ys), ptr(strides),
         ptr(d_origin), stream_ptr())
    return ws.dout


def loss_grad_decode(ws, outputs, labels, levels):
    """Round 5, the captured step: d loss / d outputs with the head's decode backward applied, written as the bf16 rows the prediction
    convs' backward consumes (``levels``: ep24.engine.Engine.decode_levels(), a host table) - no dense fp32 [B,A,27+C] gradient."""
    call("loss_grad_decode", ptr(outputs), NCOLS_BASE + ws.C, ptr(labels), ptr(ws.matched_gt), ptr(ws.matched_iou), ptr(ws.result),
         ws.B, ws.A, ws.C, levels.shape[0], levels.data_ptr(), stream_ptr())


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, labels, xs, ys, strides, ws, state, origin):
        assign_and_reduce(ws, outputs, labels, xs, ys, strides, state, origin)
        ctx.ws = ws
        ctx.l1 = origin is not None
        ctx.save_for_backward(outputs, labels, xs, ys, strides, origin)
        return ws.result.clone()

    @staticmethod
    def backward(ctx, g):
        outputs, labels, xs, ys, strides, origin = ctx.saved_tensors
        # only element 0 (the weighted total) carries gradient; 1..3 are reported values
        dout = loss_grad(ctx.ws, outputs, labels, g[0:1].contiguous(), origin, (xs, ys, strides))
        return dout, None, None, None, None, None, None, (ctx.ws.d_origin if ctx.l1 else None)


def _check_cuda(t, name):
    if not t.is_cuda:
        raise _lib.Ep24Error("ep24: %s must live on the GPU (no CPU fallback on the product path)" % name)


class Loss_Function(nn.Module):
    """Drop-in for the reference ``Loss_Function(num_classes)`` (yolox_24p/models/losses.py:159-357).

    ``forward(outputs_train, labels)`` takes the head's train-mode 5-tuple and ``labels [B,50,51]`` and returns
    ``(loss, reg_w*loss_iou[24], loss_obj, loss_cls, loss_l1, num_fg/num_gts, draw_content)``; ``loss_l1`` is the python
    float 0.0 unless ``use_l1`` is set (then a 0-dim tensor, losses.py:304-309, and the head must have produced
    ``origin_preds``).
    With ``self.draw`` True (the default, as the reference runs) position 5 is a python float as in the reference
    (losses.py:349-357) and draw_content[0:3] (matched cx / cy / radii, dynamic length) are materialised - both cost a host
    sync.  With ``self.draw`` False (what ep24.train's captured step uses) nothing synchronises: position 5 is a 0-dim device
    tensor (float()-able) and draw_content[0:3] are None.
    """

    def __init__(self, num_classes):
        super().__init__()
        self.num_classes = num_classes
        self.use_l1 = False
        self.iou_loss = IOUloss(reduction="none")
        self.draw = True
        self._ws = None
        self._state = None          # device [26]: last_iou_loss[24], last_obj_loss, last_cls_loss (init 1.0)
        self._grid = None

    # the reference exposes these as attributes that start at 1.0 (losses.py:170-172)
    @property
    def last_iou_loss(self):
        return 1.0 if self._state is None else self._state[:24]

    @property
    def last_obj_loss(self):
        return 1.0 if self._state is None else self._state[24]

    @property
    def last_cls_loss(self):
        return 1.0 if self._state is None else self._state[25]

    def workspace(self, B, A, device):
        if self._ws is None or (self._ws.B, self._ws.A) != (B, A):
            self._ws = LossWorkspace(B, A, self.num_classes, device)
        if self._state is None:
            self._state = torch.ones(26, dtype=torch.float32, device=device)
        return self._ws

    def _anchors(self, x_shifts, y_shifts, strides):
        key = tuple(t.data_ptr() for t in x_shifts)
        if self._grid is None or self._grid[0] != key:
            self._grid = (key, torch.cat(x_shifts, 1)[0].contiguous().float(),
                          torch.cat(y_shifts, 1)[0].contiguous().float(), torch.cat(strides, 1)[0].contiguous().float())
        return self._grid[1:]

    def forward(self, outputs_train, labels):
        _lib.require_gpu()
        x_shifts, y_shifts, expanded_strides, outputs, origin_preds = outputs_train
        _check_cuda(outputs, "outputs")
        if outputs.shape[2] != NCOLS_BASE + self.num_classes or labels.shape[1:] != (MAX_GT, 51):
            raise IndexError
        B, A = outputs.shape[0], outputs.shape[1]
        ws = self.workspace(B, A, outputs.device)
        xs, ys, st = self._anchors(x_shifts, y_shifts, expanded_strides)
        labels = labels.to(device=outputs.device, dtype=torch.float32).contiguous()
        outputs_c = outputs if outputs.is_contiguous() else outputs.contiguous()
        origin = None
        if self.use_l1:                                                       # losses.py:197-198
            origin = torch.cat(origin_preds, 1).float().contiguous()
            if origin.shape != (B, A, 26):
                raise IndexError
        res = _LossFn.apply(outputs_c, labels, xs, ys, st, ws, self._state, origin)
        reg_w, obj_w, cls_w = res[29:53], res[53], res[54]
        if self.draw:
            fg = ws.matched_gt.reshape(-1) >= 0
            rows = outputs_c.detach().reshape(-1, outputs_c.shape[2])[fg]
            draw = [rows[:, 0], rows[:, 1], rows[:, 2:26]]
        else:
            draw = [None, None, None]
        draw += [reg_w, obj_w, cls_w]
        loss_l1 = res[56] if self.use_l1 else 0.0
        ratio = res[55] / torch.clamp(res[28], min=1.0)                       # num_fg / max(num_gts, 1)
        if self.draw:
            ratio = float(ratio)                                              # the reference's python float (the mask above has synchronised already)
        return res[0], res[1:25], res[25], res[26], loss_l1, ratio, draw

    # --- reference helper kept for API parity (losses.py:360-442): assignment of one image of the last call
    def assignment_of(self, labels, b):
        ws = self._ws
        g = ws.matched_gt[b]
        fg = g >= 0
        idx = g[fg].long()
        return labels[b, :, 0].to(g.device)[idx], fg, ws.matched_iou[b][fg], idx, int(fg.sum())


class _MatchedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        out = torch.empty(pred.shape[0], 24, dtype=torch.float32, device=pred.device)
        call("circle_matched_fwd", ptr(pred), ptr(target), ptr(out), pred.shape[0], stream_ptr())
        ctx.save_for_backward(pred, target)
        return out

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        d = torch.empty_like(pred)
        call("circle_matched_bwd", ptr(pred), ptr(target), ptr(g.contiguous()), ptr(d), pred.shape[0], stream_ptr())
        return d, None


class IOUloss(nn.Module):
    """Drop-in for the reference ``IOUloss`` (yolox_24p/models/losses.py:14-157): matched rows, returns
    ``(1 - giou)[N,24]`` and ``[pd_cx, pd_cy, scale_pd]``."""

    def __init__(self, reduction="none"):
        super().__init__()
        self.reduction = reduction

    def circle_inter(self, c_gtx, c_gty, gt_r, c_pdx, c_pdy, pd_r):
        """Method form of the reference (losses.py:23-78): row i against row i -> (res_inter [N,24], dist [N,24])."""
        return _circle_lens(c_gtx, c_gty, gt_r, c_pdx, c_pdy, pd_r, pairwise=False)

    def forward(self, pred, target):
        if pred.shape[1] != 26 or target.shape[1] != 50:
            raise IndexError
        _lib.require_gpu()
        _check_cuda(pred, "pred")
        pred = pred.reshape(-1, 26).float()
        target = target.reshape(-1, 50).float().to(pred.device)
        if pred.shape[0] == 0 or target.shape[0] == 0:           # placeholder path, losses.py:111-115
            z = pred.new_zeros(1, 24)
            return z, [pred.new_zeros(1, 24), pred.new_zeros(1, 24), pred.new_zeros(1, 24)]
        predc = pred.contiguous()
        loss24 = _MatchedFn.apply(predc, target.contiguous())
        return loss24, [predc[:, 0], predc[:, 1], predc[:, 2:]]


def _circle_lens(c_gtx, c_gty, gt_r, c_pdx, c_pdy, pd_r, pairwise):
    """One launch of ``ep24_circle_lens``.  Neither form of the reference carries gradient users (both callers wrap the
    result in their own arithmetic on detached geometry only inside ``no_grad`` or re-derive it), so this is a plain forward."""
    _lib.require_gpu()
    _check_cuda(pd_r, "pd_r")
    dev = pd_r.device
    f = lambda t, *shape: t.detach().to(device=dev, dtype=torch.float32).reshape(*shape).contiguous()
    G, P = gt_r.shape[0], pd_r.shape[0]
    if gt_r.shape[1:] != (24,) or pd_r.shape[1:] != (24,):
        raise IndexError
    if not pairwise and G != P:
        raise RuntimeError("circle_inter: the matched form pairs row i with row i (%d gt rows, %d pred rows)" % (G, P))
    n = G * P if pairwise else G
    res = torch.zeros(n, 24, dtype=torch.float32, device=dev)
    dist = torch.zeros(n, 24, dtype=torch.float32, device=dev)
    if n:
        # the converted operands stay bound until the launch is enqueued: a temporary made by f() (strided view, other dtype or
        # device - the reference's own call passes pred[:, 0], pred[:, 1], pred[:, 2:]) would be freed as soon as ptr() returned
        # and the caching allocator would hand the same block to the next f() of equal size
        ops = [f(c_gtx, G), f(c_gty, G), f(gt_r, G, 24), f(c_pdx, P), f(c_pdy, P), f(pd_r, P, 24)]
        call("circle_lens", *[ptr(o) for o in ops], ptr(res), ptr(dist), G, P, int(pairwise), stream_ptr())
        del ops
    return res, dist


def circle_inter(c_gtx, c_gty, gt_r, c_pdx, c_pdy, pd_r):
    """Drop-in for the module-level ``utils.boxes.circle_inter`` (yolox_24p/utils/boxes.py:102-163): every gt row against
    every pred row -> (res_inter [G*P,24], dist [G*P,24]), pair index g*P + p."""
    return _circle_lens(c_gtx, c_gty, gt_r, c_pdx, c_pdy, pd_r, pairwise=True)


def bboxes_iou(bboxes_a, bboxes_b, imgs=None):
    """Drop-in for ``utils.bboxes_iou`` (yolox_24p/utils/boxes.py:166-243): [G,50] x [P,26] -> [G,P]."""
    if bboxes_b.shape[1] != 26 or bboxes_a.shape[1] != 50:
        raise IndexError
    _lib.require_gpu()
    _check_cuda(bboxes_b, "bboxes_b")
    a = bboxes_a.reshape(-1, 50).float().contiguous().to(bboxes_b.device)
    b = bboxes_b.reshape(-1, 26).float().contiguous()
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=b.device)
    call("circle_pairwise", ptr(a), ptr(b), ptr(out), a.shape[0], b.shape[0], stream_ptr())
    return out
