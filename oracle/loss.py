"""Oracle: the 24p training loss with dynamic task weights.
Test infrastructure only.

Restates Loss_Function.forward (yolox_24p/models/losses.py:175-357): per-image SimOTA targets, the three
loss terms normalised by num_fg, and the stateful softmax(T=20) weights over the 24+1+1 tasks; with ``use_l1``
the unweighted L1 term on the head's raw regression outputs as well (losses.py:197-198, 255-262, 304-309, 594-604).
"""
import torch
import torch.nn.functional as F

from . import assign, geometry


class LossOracle:
    def __init__(self, num_classes=80, use_l1=False):
        self.num_classes = num_classes
        self.use_l1 = use_l1                                                          # :163
        # losses.py:170-172
        self.last_iou = 1.0
        self.last_obj = 1.0
        self.last_cls = 1.0
        self.trace = []          # per-image assignment results of the latest call

    def __call__(self, outputs_train, labels):
        x_shifts, y_shifts, strides, outputs, origin_preds = outputs_train
        C = self.num_classes
        box = outputs[:, :, :26]
        obj = outputs[:, :, 26].unsqueeze(-1)
        cls = outputs[:, :, 27:]
        B, A = outputs.shape[0], outputs.shape[1]
        nlabel = (labels.sum(dim=2) > 0).sum(dim=1)                                   # :190
        xs = torch.cat(x_shifts, 1)[0]
        ys = torch.cat(y_shifts, 1)[0]
        st = torch.cat(strides, 1)[0]

        origin = torch.cat(origin_preds, 1) if self.use_l1 else None                  # :197-198
        l1_t = []
        cls_t, reg_t, obj_t, fgs = [], [], [], []
        num_fg, num_gts = 0.0, 0.0
        self.trace = []
        for b in range(B):
            n = int(nlabel[b])
            num_gts += n
            if n == 0:                                                                # :212-217
                cls_t.append(outputs.new_zeros((0, C)))
                reg_t.append(outputs.new_zeros((0, 50)))
                obj_t.append(outputs.new_zeros((A, 1)))
                fgs.append(outputs.new_zeros(A).bool())
                self.trace.append(None)
                continue
            gt50 = labels[b, :n, 1:]
            gcls = labels[b, :n, 0]
            res = assign.assign_image(gt50, gcls, box[b].detach(), cls[b].detach(), obj[b].detach(), xs, ys, st, C)
            cls_m, fg, ious, gt_idx, nfg = res
            self.trace.append(res)
            num_fg += nfg
            cls_t.append(F.one_hot(cls_m.to(torch.int64), C) * ious.unsqueeze(-1))   # :246-248
            obj_t.append(fg.unsqueeze(-1).to(torch.float))
            reg_t.append(gt50[gt_idx])
            fgs.append(fg)
            if self.use_l1:                                                           # :255-262, get_l1_target :594-604
                gt = gt50[gt_idx]
                s_fg = st[fg]
                t = outputs.new_zeros((nfg, 26))
                t[:, 0] = gt[:, 0] / s_fg - xs[fg]
                t[:, 1] = gt[:, 1] / s_fg - ys[fg]
                # the 24 targets use the contour POINTS' distance from the image origin (gt[:, 2::2], gt[:, 3::2]), as written
                t[:, 2:] = torch.log(torch.sqrt(gt[:, 2::2] ** 2 + gt[:, 3::2] ** 2) / s_fg.unsqueeze(1).repeat(1, 24) + 1e-8)
                l1_t.append(t)
        cls_t = torch.cat(cls_t, 0)
        reg_t = torch.cat(reg_t, 0)
        obj_t = torch.cat(obj_t, 0)
        fgs = torch.cat(fgs, 0)
        num_fg = max(num_fg, 1)

        iou24, draw = geometry.matched(box.reshape(-1, 26)[fgs], reg_t)               # :283
        loss_iou = iou24.sum(0) / num_fg
        loss_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), obj_t, reduction="none").sum() / num_fg
        loss_cls = F.binary_cross_entropy_with_logits(cls.reshape(-1, C)[fgs], cls_t, reduction="none").sum() / num_fg

        if self.use_l1:                                                               # :304-309
            l1_t = torch.cat(l1_t, 0) if l1_t else outputs.new_zeros((0, 26))
            loss_l1 = F.l1_loss(origin.reshape(-1, 26)[fgs], l1_t, reduction="none").sum() / num_fg
        else:
            loss_l1 = 0.0

        vi, vo, vc = loss_iou.detach().clone(), loss_obj.detach().clone(), loss_cls.detach().clone()
        ri = torch.clip(vi / (self.last_iou + 1e-8), 0, 2)                            # :316-323
        ro = torch.clip(vo / (self.last_obj + 1e-8), 0, 2)
        rc = torch.clip(vc / (self.last_cls + 1e-8), 0, 2)
        T = torch.tensor(20.0)
        den = torch.exp(ri / T).sum() + torch.exp(ro / T) + torch.exp(rc / T)
        reg_w = 26 * torch.exp(ri / T) / den
        obj_w = 26 * torch.exp(ro / T) / den
        cls_w = 26 * torch.exp(rc / T) / den
        loss = (reg_w * loss_iou).sum() + obj_w * loss_obj + cls_w * loss_cls + loss_l1
        self.last_iou, self.last_obj, self.last_cls = vi, vo, vc
        draw = list(draw) + [reg_w, obj_w, cls_w]
        return loss, reg_w * loss_iou, loss_obj, loss_cls, loss_l1, num_fg / max(num_gts, 1), draw
