"""Oracle: 24-concentric-circle geometry.  Test infrastructure only.

Restates
  * utils.boxes.circle_inter / bboxes_iou   (yolox_24p/utils/boxes.py:102-163, :166-243)   -> pairwise()
  * IOUloss.circle_inter / IOUloss.forward  (yolox_24p/models/losses.py:23-78, :80-157)    -> lens_area(), matched()
using broadcasting instead of the reference's repeat/repeat_interleave expansion; the per-element
arithmetic (operation order, epsilons, clip bounds, float32 pi) is kept identical.
"""
import numpy as np
import torch

RAYS = 24


def _pi():
    # torch.tensor(np.pi) is a float32 0-dim tensor (boxes.py:104,170; losses.py:24,84)
    return torch.tensor(np.pi)


def ray_lengths(boxes50):
    """r_k = |vertex_k - centre| via torch.norm over a [N,2,24] view (boxes.py:186-197, losses.py:97-108)."""
    cx = boxes50[:, 0].to(torch.float)
    cy = boxes50[:, 1].to(torch.float)
    vx = boxes50[:, 2::2].to(torch.float) - cx.reshape(-1, 1)
    vy = boxes50[:, 3::2].to(torch.float) - cy.reshape(-1, 1)
    stacked = torch.cat((vx, vy), 1).reshape(-1, 2, vx.shape[1])
    return torch.norm(stacked, dim=1)


def lens_area(r_gt, r_pd, dist):
    """Intersection area of two circles, all operands broadcastable to a common [...,24] shape.

    Case order as in the reference (boxes.py:146-157, losses.py:60-72): contained -> pi*rmin^2, then
    disjoint (d >= r1+r2) overrides with 0, otherwise the lens formula with cosines clipped to +-0.99.
    """
    rmin = torch.minimum(r_gt, r_pd)
    rmax = torch.maximum(r_gt, r_pd)
    c1 = (rmin ** 2 + dist ** 2 - rmax ** 2) / (2 * rmin * dist + 1e-8)
    c2 = (rmax ** 2 + dist ** 2 - rmin ** 2) / (2 * rmax * dist + 1e-8)
    c1 = torch.clip(c1, min=-0.99, max=0.99)
    c2 = torch.clip(c2, min=-0.99, max=0.99)
    a1 = torch.acos(c1)
    a2 = torch.acos(c2)
    lens = a1 * rmin ** 2 + a2 * rmax ** 2 - rmin * dist * torch.sin(a1)
    contained = torch.abs(r_gt - r_pd) >= dist
    disjoint = dist >= r_gt + r_pd
    small = _pi() * (rmin ** 2)
    out = torch.where(contained, small, torch.zeros_like(lens))
    out = torch.where(disjoint, torch.zeros_like(lens), out)
    out = torch.where(~(contained | disjoint), lens, out)
    return out


def _giou24(r_gt, r_pd, dist):
    pi = _pi()
    area_gt = pi * r_gt ** 2
    area_pd = pi * r_pd ** 2
    inter = lens_area(r_gt, r_pd, dist)
    iou = inter / (area_gt + area_pd - inter + 1e-6)
    contained = torch.abs(r_gt - r_pd) >= dist
    enclose_r = torch.where(contained, torch.maximum(r_gt, r_pd), (r_gt + r_pd + dist) / 2)
    enclose = pi * enclose_r ** 2
    top = enclose - (area_gt + area_pd - inter)
    return iou - top / enclose


def pairwise(gt50, pred26):
    """utils.bboxes_iou: [G,50] x [P,26] -> [G,P] = mean_k(1-giou_k)/2 (boxes.py:166-243)."""
    if pred26.shape[1] != 26 or gt50.shape[1] != 50:
        raise IndexError
    G, P = gt50.shape[0], pred26.shape[0]
    r_gt = ray_lengths(gt50)                                   # [G,24]
    r_pd = pred26[:, 2:]
    dx = gt50[:, 0].to(torch.float).reshape(G, 1) - pred26[:, 0].to(torch.float).reshape(1, P)
    dy = gt50[:, 1].to(torch.float).reshape(G, 1) - pred26[:, 1].to(torch.float).reshape(1, P)
    dist = torch.sqrt(dx ** 2 + dy ** 2).reshape(G * P, 1)
    rg = r_gt.reshape(G, 1, RAYS).expand(G, P, RAYS).reshape(G * P, RAYS)
    rp = r_pd.reshape(1, P, RAYS).expand(G, P, RAYS).reshape(G * P, RAYS)
    giou = _giou24(rg, rp, dist)
    loss = (1 - giou).sum(1) / 24
    return loss.reshape(G, P) / 2


def matched(pred26, target50):
    """IOUloss.forward: row i of pred vs row i of target -> (1-giou)[N,24], [cx, cy, r] (losses.py:80-157)."""
    if pred26.shape[1] != 26 or target50.shape[1] != 50:
        raise IndexError
    pred26 = pred26.view(-1, 26)
    target50 = target50.view(-1, 50)
    r_gt = ray_lengths(target50)
    r_pd = pred26[:, 2:]
    pcx = pred26[:, 0].to(torch.float)
    pcy = pred26[:, 1].to(torch.float)
    if r_gt.shape[0] == 0 or r_pd.shape[0] == 0:
        z = r_gt.new_zeros(1, 24)
        return z, [pcx.new_zeros(1, 24), pcy.new_zeros(1, 24), r_pd.new_zeros(1, 24)]
    dist = torch.sqrt((target50[:, 0].to(torch.float) - pcx) ** 2 + (target50[:, 1].to(torch.float) - pcy) ** 2)
    dist = dist.unsqueeze(1).repeat(1, RAYS)
    return 1 - _giou24(r_gt, r_pd, dist), [pcx, pcy, r_pd]


def matched_lens(gt_cx, gt_cy, gt_r, pd_cx, pd_cy, pd_r):
    """IOUloss.circle_inter (losses.py:23-78): returns (intersection area [N,24], centre distance [N,24])."""
    dist = torch.sqrt((gt_cx - pd_cx) ** 2 + (gt_cy - pd_cy) ** 2).unsqueeze(1).repeat(1, RAYS)
    if gt_r.shape[0] == 0 or pd_r.shape[0] == 0:
        return torch.zeros_like(gt_r), dist
    return lens_area(gt_r, pd_r, dist), dist
