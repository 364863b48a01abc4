"""Oracle: SimOTA label assignment for the 24-point head.
Test infrastructure only.

Restates Loss_Function.{pts_in_poly, get_in_boxes_info, get_assignments, dynamic_k_matching}
(yolox_24p/models/losses.py:555-592, :497-551, :360-442, :444-494) without the per-GT python loops where
the arithmetic allows; selection steps that the reference does per GT with torch.topk keep that call so
that tie behaviour is the reference's own.
"""
import torch
import torch.nn.functional as F

from . import geometry

CENTER_RADIUS = 2.5


def anchor_centres(x_shifts, y_shifts, strides):
    """(shift + 0.5) * stride, computed as shift*stride + 0.5*stride (losses.py:505-516)."""
    xs = x_shifts * strides
    ys = y_shifts * strides
    return xs + 0.5 * strides, ys + 0.5 * strides


def polygon_hits(gt50, xc, yc):
    """pts_in_poly: sum over the 24 edges of the unsigned angle subtended at the anchor centre, in
    degrees, >= 350 (losses.py:555-592).  gt50 [G,50] (uses cols 2..49), xc/yc [A] -> bool [G,A]."""
    G = gt50.shape[0]
    out = torch.zeros(G, xc.shape[0], dtype=torch.bool)
    px = gt50[:, 2::2]
    py = gt50[:, 3::2]
    for g in range(G):
        sx = px[g].reshape(24, 1) - xc                     # [24,A]
        sy = py[g].reshape(24, 1) - yc
        ex = px[g].roll(-1, 0).reshape(24, 1) - xc
        ey = py[g].roll(-1, 0).reshape(24, 1) - yc
        cross = torch.mul(sx, ey) - torch.mul(ex, sy)
        dot = sx * ex + sy * ey                            # two-term sum: order-free
        deg = torch.rad2deg(torch.atan2(torch.abs(cross), dot)).sum(0)
        out[g] = deg >= 350
    return out


def centre_hits(gt50, xc, yc, strides):
    """Anchor centre strictly inside the (2*2.5*stride)^2 square round the GT centre (losses.py:523-543)."""
    A = xc.shape[0]
    gx = gt50[:, 0].unsqueeze(1).repeat(1, A)
    gy = gt50[:, 1].unsqueeze(1).repeat(1, A)
    rad = CENTER_RADIUS * strides.unsqueeze(0)
    deltas = torch.stack([xc - (gx - rad), yc - (gy - rad), (gx + rad) - xc, (gy + rad) - yc], 2)
    return deltas.min(dim=-1).values > 0.0


def candidate_masks(gt50, x_shifts, y_shifts, strides):
    """get_in_boxes_info -> (fg_mask [A], in_both [G,P], in_box [G,A], in_ctr [G,A]) (losses.py:497-551)."""
    xc, yc = anchor_centres(x_shifts, y_shifts, strides)
    in_box = polygon_hits(gt50, xc, yc)
    in_ctr = centre_hits(gt50, xc, yc, strides)
    fg = (in_box.sum(0) > 0) | (in_ctr.sum(0) > 0)
    return fg, in_box[:, fg] & in_ctr[:, fg], in_box, in_ctr


def class_cost(cls_logits, obj_logits, gt_classes, num_classes):
    """sum_c BCE(sqrt(sigmoid(cls)*sigmoid(obj)), onehot) -> [G,P] (losses.py:399-416)."""
    G, P = gt_classes.shape[0], cls_logits.shape[0]
    onehot = F.one_hot(gt_classes.to(torch.int64), num_classes).float().unsqueeze(1).repeat(1, P, 1)
    p = (cls_logits.float().unsqueeze(0).repeat(G, 1, 1).sigmoid_()
         * obj_logits.float().unsqueeze(0).repeat(G, 1, 1).sigmoid_())
    return F.binary_cross_entropy(p.sqrt_(), onehot, reduction="none").sum(-1)


def dynamic_k(cost, pw, gt_classes, fg_mask):
    """dynamic_k_matching (losses.py:444-494).  Mutates fg_mask in place like the reference."""
    G = cost.shape[0]
    match = torch.zeros_like(cost, dtype=torch.uint8)
    top, _ = torch.topk(pw, min(10, pw.size(1)), dim=1)
    ks = torch.clamp(top.sum(1).int(), min=1).tolist()
    for g in range(G):
        _, pos = torch.topk(cost[g], k=ks[g], largest=False)
        match[g][pos] = 1
    per_anchor = match.sum(0)
    if (per_anchor > 1).sum() > 0:
        _, best = torch.min(cost[:, per_anchor > 1], dim=0)
        match[:, per_anchor > 1] *= 0
        match[best, per_anchor > 1] = 1
    keep = match.sum(0) > 0
    num_fg = keep.sum().item()
    fg_mask[fg_mask.clone()] = keep
    matched_gt = match[:, keep].argmax(0)
    return num_fg, gt_classes[matched_gt], (match * pw).sum(0)[keep], matched_gt, ks


def assign_image(gt50, gt_classes, pred26, cls_logits, obj_logits, x_shifts, y_shifts, strides, num_classes=80,
                 detail=False):
    """get_assignments for one image (losses.py:360-442).

    gt50 [G,50], gt_classes [G], pred26 [A,26] decoded, cls_logits [A,C], obj_logits [A,1];
    x_shifts / y_shifts / strides [A].  Returns (gt_matched_classes, fg_mask[A], pred_ious, matched_gt_inds,
    num_fg) and, with detail=True, a dict of the intermediate [G,P] matrices.
    """
    with torch.no_grad():
        fg, in_both, in_box, in_ctr = candidate_masks(gt50, x_shifts, y_shifts, strides)
        pw = geometry.pairwise(gt50, pred26[fg])
        cls_c = class_cost(cls_logits[fg], obj_logits[fg], gt_classes, num_classes)
        cost = cls_c + 3.0 * (-torch.log(pw + 1e-8)) + 100000.0 * (~in_both)
        fg_pre = fg.clone()
        num_fg, cls_m, ious, gt_idx, ks = dynamic_k(cost, pw, gt_classes, fg)
    out = (cls_m, fg, ious, gt_idx, num_fg)
    if detail:
        return out, dict(fg_pre=fg_pre, in_both=in_both, pw=pw, cls_cost=cls_c, cost=cost, ks=ks,
                         in_box=in_box, in_ctr=in_ctr)
    return out
