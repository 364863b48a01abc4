"""Oracle: the 24p training loss with dynamic task weights.
Test infrastructure only.

Restates Loss_Function.forward (yolox_24p/models/losses.py:175-357): per-image SimOTA targets, the three
loss terms normalised by num_fg, and the stateful softmax(T=20) weights over the 24+1+1 tasks.
"""
import torch
import torch.nn.functional as F

from . import assign, geometry


class LossOracle:
    def __init__(self, num_classes=80):
        self.num_classes = num_classes
        # losses.py:170-172
        self.last_iou = 1.0
        self.last_obj = 1.0
        self.last_cls = 1.0
        self.trace = []          # per-image assignment results of the latest call

    def __call__(self, outputs_train, labels):
        x_shifts, y_shifts, strides, outputs, _ = outputs_train
        C = self.num_classes
        box = outputs[:, :, :26]
        obj = outputs[:, :, 26].unsqueeze(-1)
        cls = outputs[:, :, 27:]
        B, A = outputs.shape[0], outputs.shape[1]
        nlabel = (labels.sum(dim=2) > 0).sum(dim=1)                                   # :190
        xs = torch.cat(x_shifts, 1)[0]
        ys = torch.cat(y_shifts, 1)[0]
        st = torch.cat(strides, 1)[0]

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
        cls_t = torch.cat(cls_t, 0)
        reg_t = torch.cat(reg_t, 0)
        obj_t = torch.cat(obj_t, 0)
        fgs = torch.cat(fgs, 0)
        num_fg = max(num_fg, 1)

        iou24, draw = geometry.matched(box.reshape(-1, 26)[fgs], reg_t)               # :283
        loss_iou = iou24.sum(0) / num_fg
        loss_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), obj_t, reduction="none").sum() / num_fg
        loss_cls = F.binary_cross_entropy_with_logits(cls.reshape(-1, C)[fgs], cls_t, reduction="none").sum() / num_fg

        vi, vo, vc = loss_iou.detach().clone(), loss_obj.detach().clone(), loss_cls.detach().clone()
        ri = torch.clip(vi / (self.last_iou + 1e-8), 0, 2)                            # :316-323
        ro = torch.clip(vo / (self.last_obj + 1e-8), 0, 2)
        rc = torch.clip(vc / (self.last_cls + 1e-8), 0, 2)
        T = torch.tensor(20.0)
        den = torch.exp(ri / T).sum() + torch.exp(ro / T) + torch.exp(rc / T)
        reg_w = 26 * torch.exp(ri / T) / den
        obj_w = 26 * torch.exp(ro / T) / den
        cls_w = 26 * torch.exp(rc / T) / den
        loss = (reg_w * loss_iou).sum() + obj_w * loss_obj + cls_w * loss_cls + 0.0
        self.last_iou, self.last_obj, self.last_cls = vi, vo, vc
        draw = list(draw) + [reg_w, obj_w, cls_w]
        return loss, reg_w * loss_iou, loss_obj, loss_cls, 0.0, num_fg / max(num_gts, 1), draw
