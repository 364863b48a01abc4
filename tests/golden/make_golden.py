#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE on this container's CPU.

Run only in the build container (the reference tree lives at /root/reference and never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is committed is data only: seeds / small inputs and the reference's outputs as .npz.  The harness
follows SURVEY.md Appendix A: permissive stubs for third-party modules the path imports but never uses
arithmetically, and a wrapper that rewrites the reference's hard-coded ``device='cuda'`` factory calls
(yolo_head_24p.py:176, losses.py:561,566) to CPU.  Reference files are not modified or copied.
"""
import importlib
import os
import sys
import types
import zlib

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))

from ep24 import synth  # noqa: E402


# --------------------------------------------------------------------------- harness
class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        child = sys.modules.get(full)
        if child is None:
            child = _Stub(full)
            child.__path__ = []
            sys.modules[full] = child
        return child

    def __call__(self, *a, **k):
        return self


def install_stubs():
    for name in ["loguru", "cv2", "zmq", "thop", "torchvision", "torchvision.ops", "tensorboard",
                 "torch.utils.tensorboard", "pycocotools", "pycocotools.coco", "pycocotools.cocoeval",
                 "pycocotools.mask", "seaborn", "prettytable", "skimage", "skimage.measure"]:
        if name not in sys.modules:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
    # loguru.logger.catch is used as a decorator
    lg = sys.modules["loguru"]
    lg.logger = _Stub("loguru.logger")


def cpu_factories():
    for fname in ("zeros", "arange", "ones", "tensor"):
        orig = getattr(torch, fname)

        def wrapped(*a, __orig=orig, **k):
            dev = k.get("device")
            if isinstance(dev, str) and dev.startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)

        setattr(torch, fname, wrapped)


def load_reference():
    install_stubs()
    cpu_factories()
    sys.path.insert(0, os.path.join(REF, "yolox_24p"))
    utils = importlib.import_module("utils")
    models = importlib.import_module("models")
    return utils, models


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(os.environ.get("EP24_GOLDEN_OUT", HERE), name + ".npz")     # EP24_GOLDEN_OUT: regenerate elsewhere (drift test)
    np.savez_compressed(path, **out)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def checksum(t):
    return float(t.double().sum())


# --------------------------------------------------------------------------- G1..G3 circle geometry
def rows_for_matched(n, seed):
    """pred [n,26] / target [n,50] rows covering contained / disjoint / lens cases."""
    g = torch.Generator().manual_seed(seed)
    lab = synth.make_labels(1, min(n, 50), seed=seed + 1)[0, :min(n, 50), 1:]
    reps = (n + lab.shape[0] - 1) // lab.shape[0]
    target = lab.repeat(reps, 1)[:n].clone()
    pred = torch.empty(n, 26)
    # centre offsets from 0 to 250 px: small -> containment, large -> disjoint
    off = torch.rand(n, generator=g) * 250.0
    off[::7] = 0.0                                   # exactly coincident centres (d == 0)
    th = torch.rand(n, generator=g) * 6.2831853
    pred[:, 0] = target[:, 0] + off * torch.cos(th)
    pred[:, 1] = target[:, 1] + off * torch.sin(th)
    pred[:, 2:] = torch.exp(torch.randn(n, 24, generator=g) * 0.8 + 3.6)
    return pred, target


def gen_geometry(utils, models):
    iou = models.IOUloss(reduction="none")
    # G1: method circle_inter
    pred, target = rows_for_matched(96, 11)
    gt_r = torch.sqrt((target[:, 2::2] - target[:, :1]) ** 2 + (target[:, 3::2] - target[:, 1:2]) ** 2)
    res, dist = iou.circle_inter(target[:, 0], target[:, 1], gt_r, pred[:, 0], pred[:, 1], pred[:, 2:])
    e_res, e_dist = iou.circle_inter(target[:0, 0], target[:0, 1], gt_r[:0], pred[:0, 0], pred[:0, 1], pred[:0, 2:])
    save("g1_circle_inter", gt_cx=target[:, 0], gt_cy=target[:, 1], gt_r=gt_r, pd_cx=pred[:, 0], pd_cy=pred[:, 1],
         pd_r=pred[:, 2:], res_inter=res, dist=dist, empty_res_shape=np.array(e_res.shape),
         empty_dist_shape=np.array(e_dist.shape))

    # G2: pairwise bboxes_iou
    raw = synth.make_raw_head(1, seed=21)
    dec = synth.decode_head(raw)[0]
    for tag, (G, P) in {"a": (1, 1), "b": (3, 17), "c": (10, 4000), "d": (50, 8400)}.items():
        a = synth.make_labels(1, G, seed=22 + G)[0, :G, 1:]
        gsel = torch.Generator().manual_seed(5)
        idx = torch.randperm(dec.shape[0], generator=gsel)[:P].sort().values
        b = dec[idx, :26].contiguous()
        out = utils.bboxes_iou(a, b)
        if tag == "d":
            save("g2_pairwise_" + tag, G=G, P=P, label_seed=22 + G, head_seed=21, sel_seed=5,
                 a_sum=checksum(a), b_sum=checksum(b), out_sub=out[:, ::7], out_sum=checksum(out))
        else:
            save("g2_pairwise_" + tag, a=a, b=b, out=out)
    # error convention (boxes.py:167-168)
    try:
        utils.bboxes_iou(torch.zeros(2, 49), torch.zeros(2, 26))
        raised = False
    except IndexError:
        raised = True
    assert raised

    # G3: matched loss forward + gradient
    pred, target = rows_for_matched(200, 31)
    pred.requires_grad_(True)
    loss24, draw = iou(pred, target)
    w = torch.linspace(0.5, 1.5, 24)
    (loss24 * w).sum().backward()
    e_loss, e_draw = iou(pred[:0], target[:0])
    save("g3_matched", pred=pred.detach(), target=target, loss24=loss24, w=w, grad=pred.grad,
         empty_loss=e_loss, empty_draw0_shape=np.array(e_draw[0].shape))


# --------------------------------------------------------------------------- G4..G6 assignment + loss
def gen_assign(utils, models):
    lf = models.Loss_Function(80)
    xs, ys, ss = synth.anchor_grid()
    A = xs.numel()
    # G4: masks for convex and star polygons; record the margin to the 350-degree threshold
    for tag, star in (("convex", False), ("star", True)):
        lab = synth.make_labels(1, 12, seed=41, star=star)[0, :12, 1:]
        captured = {}
        orig_rad2deg = torch.rad2deg

        def spy(x, _o=orig_rad2deg):
            r = _o(x)
            captured.setdefault("deg", []).append(r.sum(0))
            return r

        torch.rad2deg = spy
        try:
            fg, both = lf.get_in_boxes_info(lab, ss[None], xs[None], ys[None], A)
        finally:
            torch.rad2deg = orig_rad2deg
        deg = torch.stack(captured["deg"])
        xc = (xs * ss + 0.5 * ss)
        yc = (ys * ss + 0.5 * ss)
        in_box = lf.pts_in_poly(lab, xc, yc)
        save("g4_masks_" + tag, label_seed=41, star=int(star), G=12, fg=fg, in_both=both, in_box=in_box,
             min_margin=float((deg - 350.0).abs().min()), lab_sum=checksum(lab))

    # G5/G6: full loss, two consecutive calls, batch with an empty image, gradient wrt outputs
    B = 4
    counts = [10, 0, 3, 25]
    labels = synth.make_labels(B, counts, seed=51)
    raw = synth.make_raw_head(B, seed=52)
    rec = []
    orig_assign = lf.get_assignments

    def spy_assign(*a, **k):
        out = orig_assign(*a, **k)
        rec.append(out)
        return out

    lf.get_assignments = spy_assign
    store = dict(B=B, counts=np.array(counts), label_seed=51, head_seed=52, labels_sum=checksum(labels),
                 raw_sum=checksum(raw))
    for call in range(2):
        outputs = synth.decode_head(raw if call == 0 else raw * 0.98 + 0.01)
        outputs.requires_grad_(True)
        rec.clear()
        tup = lf.forward(synth.outputs_train_tuple(outputs), labels)
        tup[0].backward()
        p = "c%d_" % call
        store[p + "loss"] = tup[0]
        store[p + "loss_iou_w"] = tup[1]
        store[p + "loss_obj"] = tup[2]
        store[p + "loss_cls"] = tup[3]
        store[p + "loss_l1"] = float(tup[4])
        store[p + "fg_per_gt"] = float(tup[5])
        store[p + "draw_cx"] = tup[6][0]
        store[p + "draw_cy"] = tup[6][1]
        store[p + "draw_r"] = tup[6][2]
        store[p + "reg_w"] = tup[6][3]
        store[p + "obj_w"] = tup[6][4]
        store[p + "cls_w"] = tup[6][5]
        g = outputs.grad
        store[p + "grad_sum"] = checksum(g)
        store[p + "grad_abs_sum"] = checksum(g.abs())
        nz = g.abs().sum(-1).reshape(-1).topk(400).indices.sort().values      # rows with the largest gradients
        store[p + "grad_rows"] = nz
        store[p + "grad_vals"] = g.reshape(-1, g.shape[-1])[nz]
        store[p + "grad_obj"] = g[..., 26].reshape(-1)[::5]
        img = 0
        for b in range(B):
            if counts[b] == 0:
                continue
            cls_, fgm, pious, minds, nfg = rec[img]
            img += 1
            store[p + "img%d_fg" % b] = fgm
            store[p + "img%d_cls" % b] = cls_
            store[p + "img%d_pious" % b] = pious
            store[p + "img%d_gt" % b] = minds
            store[p + "img%d_nfg" % b] = int(nfg)
    save("g6_loss", **store)

    # G5: one dense image at G=50 (stress) - assignment only
    lf2 = models.Loss_Function(80)
    labels = synth.make_labels(1, 50, seed=61)
    outputs = synth.decode_head(synth.make_raw_head(1, seed=62))
    rec2 = []
    o2 = lf2.get_assignments
    lf2.get_assignments = lambda *a, **k: (rec2.append(o2(*a, **k)) or rec2[-1])
    tup = lf2.forward(synth.outputs_train_tuple(outputs), labels)
    cls_, fgm, pious, minds, nfg = rec2[0]
    save("g5_assign_g50", label_seed=61, head_seed=62, fg=fgm, cls=cls_, pious=pious, gt=minds, nfg=int(nfg),
         loss=tup[0], loss_iou_w=tup[1], loss_obj=tup[2], loss_cls=tup[3])


# --------------------------------------------------------------------------- G7 model
def gen_model(utils, models):
    nb = importlib.import_module("models.network_blocks")
    torch.manual_seed(0)

    def bn_patch(m):
        for x in m.modules():
            if isinstance(x, torch.nn.BatchNorm2d):
                x.eps, x.momentum = 1e-3, 0.03
                with torch.no_grad():
                    x.weight.uniform_(0.5, 1.5)
                    x.bias.uniform_(-0.3, 0.3)

    def run_block(name, mod, x):
        bn_patch(mod)
        mod.train()
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        x = x.clone().requires_grad_(True)
        y = mod(x)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        y.backward(gy)
        store = {"x": x.detach(), "y": y, "gy": gy, "gx": x.grad}
        for k, v in sd.items():
            store["w:" + k] = v
        for k, v in mod.state_dict().items():
            if "running" in k or "num_batches" in k:
                store["after:" + k] = v
        for k, p in mod.named_parameters():
            store["g:" + k] = p.grad
        save("g7_" + name, **store)

    g = torch.Generator().manual_seed(70)
    run_block("baseconv3", nb.BaseConv(16, 24, 3, 1), torch.randn(2, 16, 12, 12, generator=g))
    run_block("baseconv3s2", nb.BaseConv(16, 32, 3, 2), torch.randn(2, 16, 12, 12, generator=g))
    run_block("baseconv1", nb.BaseConv(16, 8, 1, 1), torch.randn(2, 16, 10, 10, generator=g))
    run_block("bottleneck", nb.Bottleneck(16, 16, True, 1.0), torch.randn(2, 16, 10, 10, generator=g))
    run_block("csp", nb.CSPLayer(16, 16, n=2), torch.randn(2, 16, 10, 10, generator=g))
    run_block("csp_noshort", nb.CSPLayer(32, 16, n=1, shortcut=False), torch.randn(2, 32, 10, 10, generator=g))
    run_block("spp", nb.SPPBottleneck(16, 16), torch.randn(2, 16, 20, 20, generator=g))
    run_block("focus", nb.Focus(3, 8, ksize=3), torch.randn(2, 3, 16, 16, generator=g))

    # whole model at width 0.125 / depth 0.33, 64x64 input -> train-mode 5-tuple
    torch.manual_seed(0)
    in_ch = [256, 512, 1024]
    model = models.YOLOX(models.YOLOPAFPN(0.33, 0.125, in_channels=in_ch, act="silu"),
                         models.YOLOXHead(80, 0.125, in_channels=in_ch, act="silu"))
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    model.train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(71)) * 255.0
    xs, ys, st, out, orig = model(x, train=True)
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(72)) * 1e-3
    out.backward(gy)
    store = {"x": x, "out": out, "gy": gy, "x_shift0": xs[0], "y_shift1": ys[1], "stride2": st[2],
             "n_params": sum(p.numel() for p in model.parameters())}
    for k, v in sd.items():
        store["w:" + k] = v
    for k in ("backbone.backbone.stem.conv.conv.weight", "backbone.C3_p3.conv3.conv.weight",
              "head.reg_preds.1.weight", "head.reg_preds.1.bias", "head.cls_preds.0.bias",
              "backbone.backbone.dark5.2.m.0.conv2.bn.weight", "head.stems.2.bn.bias"):
        store["g:" + k] = dict(model.named_parameters())[k].grad
    store["after:stem_rm"] = model.state_dict()["backbone.backbone.stem.conv.bn.running_mean"]
    store["after:stem_rv"] = model.state_dict()["backbone.backbone.stem.conv.bn.running_var"]
    model.eval()
    with torch.no_grad():
        store["out_eval"] = model(x, train=False)
    save("g7_model_tiny", **store)

    # -l: names / shapes / parameter count only
    torch.manual_seed(0)
    big = models.YOLOX(models.YOLOPAFPN(1.0, 1.0, in_channels=in_ch), models.YOLOXHead(80, 1.0, in_channels=in_ch))
    keys = list(big.state_dict().keys())
    shapes = [tuple(v.shape) for v in big.state_dict().values()]
    save("g7_model_l_keys", keys=np.array(keys), shapes=np.array([str(s) for s in shapes]),
         n_params=sum(p.numel() for p in big.parameters()))


# --------------------------------------------------------------------------- G19 depthwise variants (DWConv, network_blocks.py:57-76)
def gen_depthwise(utils, models):
    """The reference's own DWConv block (depthwise BaseConv + 1x1 BaseConv), a Bottleneck built from it, and the whole network with
    depthwise=True in backbone, neck and head (darknet.py:107, network_blocks.py:92, yolo_pafpn.py:30, yolo_head_24p.py:45)."""
    nb = importlib.import_module("models.network_blocks")
    torch.manual_seed(190)

    def bn_patch(m):
        for x in m.modules():
            if isinstance(x, torch.nn.BatchNorm2d):
                x.eps, x.momentum = 1e-3, 0.03
                with torch.no_grad():
                    x.weight.uniform_(0.5, 1.5)
                    x.bias.uniform_(-0.3, 0.3)

    def run_block(name, mod, x):
        bn_patch(mod)
        mod.train()
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        x = x.clone().requires_grad_(True)
        y = mod(x)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(197))
        y.backward(gy)
        store = {"x": x.detach(), "y": y, "gy": gy, "gx": x.grad}
        for k, v in sd.items():
            store["w:" + k] = v
        for k, v in mod.state_dict().items():
            if "running" in k or "num_batches" in k:
                store["after:" + k] = v
        for k, p in mod.named_parameters():
            store["g:" + k] = p.grad
        mod.eval()
        with torch.no_grad():
            store["y_eval"] = mod(x.detach())
        save("g19_" + name, **store)

    g = torch.Generator().manual_seed(191)
    run_block("dwconv3", nb.DWConv(16, 24, 3, 1), torch.randn(2, 16, 12, 12, generator=g))
    run_block("dwconv3s2", nb.DWConv(16, 32, 3, 2), torch.randn(2, 16, 12, 12, generator=g))
    run_block("bottleneck_dw", nb.Bottleneck(16, 16, True, 1.0, depthwise=True), torch.randn(2, 16, 10, 10, generator=g))
    run_block("csp_dw", nb.CSPLayer(16, 16, n=2, depthwise=True), torch.randn(2, 16, 10, 10, generator=g))

    torch.manual_seed(192)
    in_ch = [256, 512, 1024]
    model = models.YOLOX(models.YOLOPAFPN(0.33, 0.125, in_channels=in_ch, depthwise=True, act="silu"),
                         models.YOLOXHead(80, 0.125, in_channels=in_ch, act="silu", depthwise=True))
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    model.train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(193)) * 255.0
    xs, ys, st, out, orig = model(x, train=True)
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(194)) * 1e-3
    out.backward(gy)
    store = {"x": x, "out": out, "gy": gy, "n_params": sum(p.numel() for p in model.parameters()),
             "keys": np.array(list(sd.keys())), "shapes": np.array([str(tuple(v.shape)) for v in sd.values()])}
    for k, v in sd.items():
        store["w:" + k] = v
    named = dict(model.named_parameters())
    for k in ("backbone.backbone.dark3.0.dconv.conv.weight", "backbone.backbone.dark3.0.pconv.conv.weight",
              "backbone.backbone.dark4.1.m.0.conv2.dconv.conv.weight", "backbone.bu_conv2.dconv.conv.weight",
              "head.cls_convs.0.0.dconv.conv.weight", "head.reg_convs.1.1.pconv.bn.weight", "backbone.backbone.stem.conv.conv.weight"):
        store["g:" + k] = named[k].grad
    model.eval()
    with torch.no_grad():
        store["out_eval"] = model(x, train=False)
    save("g19_model_dw_tiny", **store)


# --------------------------------------------------------------------------- G8 sector warp
def gen_sector():
    sys.path.insert(0, REF)
    np.bool8 = np.bool_
    demo = importlib.import_module("yolox.demo_featuremap")
    calls = {}

    def fake_resize(img, dsize, *a, **k):
        # identity stand-in: inputs are generated at the resize target so cv2's bilinear step is bypassed;
        # the size the reference asked for is recorded
        calls["dsize"] = dsize
        assert img.shape[1] == dsize[0] and img.shape[0] == dsize[1], (img.shape, dsize)
        return img

    demo.cv2.resize = fake_resize
    dist = demo.Image_Distortion()
    store = {}
    for theta in (30, 60, 90, 180):
        for (h, w) in ((427, 640), (640, 640), (1280, 1280)):
            # first call on a dummy to learn target rows T for this (theta, aspect)
            T = _sector_rows(theta, h, w)
            rng = np.random.RandomState(theta * 7 + h)
            # index image: encode (row, col) of the resized source in the three channels exactly
            rows = np.arange(T, dtype=np.int64)[:, None].repeat(13200, 1)
            cols = np.arange(13200, dtype=np.int64)[None, :].repeat(T, 0)
            flat = rows * 13200 + cols + 1                      # 1-based, < 2^24
            img = np.stack([flat & 255, (flat >> 8) & 255, (flat >> 16) & 255], -1).astype(np.uint8)
            mask = np.zeros_like(img)
            r0, r1 = sorted(rng.randint(0, T, 2).tolist())
            c0, c1 = sorted(rng.randint(0, 13200, 2).tolist())
            mask[r0:r1 + 1, c0:c1 + 1] = 255
            # the reference derives T from image.shape -> hand it an image of the ORIGINAL aspect only
            # through scale_hw; we pass custom_rows to pin T and a [T,13200] source so resize is identity
            new_img, bbox = dist.sector_distort(img, mask, Theta=theta, custom_rows=T)
            code = (new_img[..., 0].astype(np.int64) | (new_img[..., 1].astype(np.int64) << 8)
                    | (new_img[..., 2].astype(np.int64) << 16))
            fill = (new_img[..., 0] == 114) & (new_img[..., 1] == 114) & (new_img[..., 2] == 114)
            key = "t%d_%dx%d_" % (theta, h, w)
            store[key + "T"] = T
            store[key + "shape"] = np.array(new_img.shape)
            src = np.ascontiguousarray(np.where(fill, -1, code - 1).astype(np.int32))
            store[key + "src_sub"] = src[::4, ::4]               # full map is MBs: keep a lattice + CRC
            store[key + "src_crc"] = zlib.crc32(src.tobytes())
            store[key + "n_fill"] = int(fill.sum())
            store[key + "bbox"] = np.array(bbox, dtype=np.int64)
            store[key + "mask_rect"] = np.array([r0, r1, c0, c1])
    save("g8_sector", **store)


def _sector_rows(theta, h, w):
    """T = clip(int(arc_len*H/W), 0, 900) computed by calling the reference's own arithmetic path:
    run sector_distort on a 1-row dummy and read the dsize it requests from cv2.resize."""
    sys.path.insert(0, REF)
    demo = importlib.import_module("yolox.demo_featuremap")
    got = {}

    class Stop(Exception):
        pass

    def probe(img, dsize, *a, **k):
        got["dsize"] = dsize
        raise Stop()

    keep = demo.cv2.resize
    demo.cv2.resize = probe
    try:
        demo.Image_Distortion().sector_distort(np.zeros((h, w, 3), np.uint8), np.zeros((h, w, 3), np.uint8), Theta=theta)
    except Stop:
        pass
    finally:
        demo.cv2.resize = keep
    return int(got["dsize"][1])


def gen_post(utils, models):
    # G9: postprocess (utils/boxes.py:29-99) with the oracle's NMS standing in for the absent torchvision ops
    sys.path.insert(0, ROOT)
    from oracle import post as opost
    tv_ops = sys.modules["torchvision.ops"]
    tv_ops.nms = opost.nms
    tv_ops.batched_nms = opost.batched_nms
    sys.modules["torchvision"].ops = tv_ops
    for tag, agnostic in (("a", False), ("b", True)):
        raw = synth.make_raw_head(3, seed=90, num_classes=80)
        pred = synth.decode_head(raw)
        gq = torch.Generator().manual_seed(91)
        pred[..., 26:] = torch.sigmoid(raw[..., 26:] + 4.5 + torch.randn(raw[..., 26:].shape, generator=gq))
        pred[2, :, 26] = 0.0                                   # an image where nothing passes the confidence filter
        # the reference re-binds cos_theta_all inside its image loop (boxes.py:64-65) and raises on the second image that
        # has detections, so it is run one image at a time (show_24p.py uses batch 1)
        outs = [utils.postprocess(pred[i:i + 1].clone(), 80, conf_thre=0.7, nms_thre=0.45, class_agnostic=agnostic)[0]
                for i in range(3)]
        store = {"head_seed": 90, "noise_seed": 91, "agnostic": int(agnostic), "n_img": 3}
        for i, o in enumerate(outs):
            store["img%d_n" % i] = -1 if o is None else o.shape[0]
            if o is not None:
                store["img%d_det" % i] = o
        save("g9_postprocess_" + tag, **store)


# --------------------------------------------------------------------------- G10..G12 the "long run" pieces
LR_CASES = [
    ("cos", dict(), 0.01, 37, 6),
    ("warmcos", dict(warmup_epochs=2), 0.02, 25, 8),
    ("warmcos", dict(warmup_epochs=1, warmup_lr_start=1e-4), 0.02, 25, 8),
    ("yoloxwarmcos", dict(warmup_epochs=5, warmup_lr_start=0, no_aug_epochs=100, min_lr_ratio=0.05), 0.01 / 64.0 * 20, 50, 300),
    ("yoloxwarmcos", dict(warmup_epochs=1, no_aug_epochs=2), 0.005, 40, 10),
    ("yoloxsemiwarmcos", dict(warmup_epochs=1, no_aug_epochs=2, semi_epoch=5, iters_per_epoch_semi=17), 0.01, 30, 12),
    ("multistep", dict(milestones=[3, 7]), 0.1, 20, 10),
    ("multistep", dict(milestones=[2, 4, 6], gamma=0.5), 0.1, 20, 8),
]


def gen_n2(utils, models):
    # G10: learning-rate schedules (utils/lr_scheduler.py) sampled over whole runs
    store = {"n_cases": len(LR_CASES)}
    for i, (name, kw, lr, ipe, epochs) in enumerate(LR_CASES):
        sch = utils.LRScheduler(name, lr, ipe, epochs, **kw)
        total = ipe * epochs
        its = np.unique(np.concatenate([np.arange(0, total + 1, max(total // 400, 1)), np.arange(0, min(total, 3 * ipe) + 1),
                                        np.arange(max(total - 3 * ipe, 0), total + 1)]))
        store["c%d_iters" % i] = its
        store["c%d_lr" % i] = np.array([sch.update_lr(int(t)) for t in its], dtype=np.float64)
    save("g10_lr", **store)

    # G11: ModelEMA (utils/ema.py) over a parameter, a float buffer and an integer buffer, four updates
    class Toy(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            g = torch.Generator().manual_seed(110)
            self.w = torch.nn.Parameter(torch.randn(n, generator=g))
            self.register_buffer("stat", torch.rand(n // 4, generator=g))
            self.register_buffer("count", torch.tensor(7, dtype=torch.long))

    toy = Toy(4099)
    for start in (0, 1500):
        toy.load_state_dict(Toy(4099).state_dict())
        ema = utils.ModelEMA(toy, decay=0.9998 if start else 0.9999, updates=start)
        store = {"start": start, "decay": 0.9998 if start else 0.9999, "w0": toy.w.detach().clone(), "stat0": toy.stat.clone()}
        g = torch.Generator().manual_seed(111 + start)
        for step in range(4):
            with torch.no_grad():
                toy.w.add_(torch.randn(toy.w.shape, generator=g) * 0.05)
                toy.stat.mul_(0.9).add_(torch.rand(toy.stat.shape, generator=g) * 0.1)
                toy.count.add_(1)
            ema.update(toy)
            store["w_model%d" % step] = toy.w.detach().clone()
            store["stat_model%d" % step] = toy.stat.clone()
            store["w_ema%d" % step] = ema.ema.w.detach().clone()
            store["stat_ema%d" % step] = ema.ema.stat.clone()
        store["count_ema"] = int(ema.ema.count)
        store["updates"] = ema.updates
        save("g11_ema_%d" % start, **store)

    # G12: the loss with use_l1 (losses.py:197-198, 255-262, 304-309): totals and the gradient wrt origin_preds
    lf = models.Loss_Function(80)
    lf.use_l1 = True
    B, counts = 3, [10, 3, 25]
    labels = synth.make_labels(B, counts, seed=121)
    raw = synth.make_raw_head(B, seed=122)
    outputs = synth.decode_head(raw).requires_grad_(True)
    tup5 = list(synth.outputs_train_tuple(outputs))
    origin, a0 = [], 0
    for s in synth.STRIDES:
        n = (640 // s) ** 2
        origin.append(raw[:, a0:a0 + n, :26].clone().requires_grad_(True))
        a0 += n
    tup5[4] = origin
    tup = lf.forward(tuple(tup5), labels)
    tup[0].backward()
    g_or = torch.cat([o.grad for o in origin], 1)
    rows = g_or.abs().sum(-1).reshape(-1).nonzero().reshape(-1)
    g_out = outputs.grad
    save("g12_loss_l1", B=B, counts=np.array(counts), label_seed=121, head_seed=122, loss=tup[0], loss_iou_w=tup[1],
         loss_obj=tup[2], loss_cls=tup[3], loss_l1=tup[4], fg_per_gt=float(tup[5]), d_origin_rows=rows,
         d_origin_vals=g_or.reshape(-1, 26)[rows], d_origin_abs_sum=checksum(g_or.abs()), grad_abs_sum=checksum(g_out.abs()),
         grad_obj=g_out[..., 26].reshape(-1)[::7])

# --------------------------------------------------------------------------- G13 label generation
def label_masks():
    """Synthetic instance masks: (tag, mask uint8 [H,W], centre x, centre y) - convex, star-shaped, concave with a hole,
    cut by the image border, and a one-pixel object."""
    out = []

    def grid(H, W):
        return np.mgrid[0:H, 0:W]

    yy, xx = grid(97, 131)
    m = (((xx - 60.3) / 40.0) ** 2 + ((yy - 45.2) / 22.0) ** 2 <= 1.0).astype(np.uint8)
    out.append(("ellipse", m))
    yy, xx = grid(240, 320)
    ang = np.arctan2(yy - 118.0, xx - 171.0)
    rr = np.hypot(yy - 118.0, xx - 171.0)
    m = (rr <= 55.0 + 30.0 * np.cos(5 * ang)).astype(np.uint8)
    out.append(("star", m))
    yy, xx = grid(200, 260)
    m = ((np.hypot(yy - 100.0, xx - 120.0) <= 70) & ~(np.hypot(yy - 100.0, xx - 150.0) <= 45)).astype(np.uint8)   # crescent: the centre is outside
    m[95:106, 40:80] = 0                                                                                           # and a slit
    out.append(("crescent", m))
    yy, xx = grid(180, 240)
    m = (((xx - 225.0) / 50.0) ** 2 + ((yy - 12.0) / 40.0) ** 2 <= 1.0).astype(np.uint8)                           # cut by two borders
    out.append(("border", m))
    m = np.zeros((64, 80), np.uint8)
    m[31, 47] = 1
    out.append(("pixel", m))
    yy, xx = grid(480, 640)
    g = np.random.RandomState(131)
    m = np.zeros((480, 640), np.uint8)
    for _ in range(14):                                                                                            # blobby union
        cy, cx, r = g.uniform(150, 330), g.uniform(200, 440), g.uniform(25, 70)
        m |= (np.hypot(yy - cy, xx - cx) <= r).astype(np.uint8)
    out.append(("blobs", m))
    res = []
    for tag, m in out:
        ys, xs = np.nonzero(m)
        x0, y0, w, h = float(xs.min()), float(ys.min()), float(xs.max() - xs.min()) + 0.37, float(ys.max() - ys.min()) + 0.81
        res.append((tag, m, x0 + w / 2, y0 + h / 2))                                                                # bbox centre as :167-168
    return res


def gen_labels():
    import importlib.util
    cv2 = sys.modules["cv2"]
    # cv2 is not installed: zero padding is the one cv2 call rotation_for_24p makes (no arithmetic involved)
    cv2.BORDER_CONSTANT = 0
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=0: np.pad(img, ((t, b), (l, r)), constant_values=value)
    spec = importlib.util.spec_from_file_location("labels_create_ref", os.path.join(REF, "yolox_24p", "datasets", "2+24_labels_create.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    poly = mod.Polygon_24.__new__(mod.Polygon_24)                     # __init__ opens the COCO json of the author's machine
    store = {"tags": np.array([t for t, *_ in label_masks()])}
    for tag, m, cx, cy in label_masks():
        pts, rad = poly.rotation_for_24p(cx, cy, m)
        store[tag + "_mask"] = np.packbits(m, axis=1)
        store[tag + "_shape"] = np.array(m.shape)
        store[tag + "_centre"] = np.array([cx, cy], dtype=np.float64)
        store[tag + "_pts"] = pts.astype(np.int64)
        store[tag + "_rad"] = rad.astype(np.float64)
    save("g13_labels24", **store)

# --------------------------------------------------------------------------- G14 input pipeline
INPUT_CASES = [("wide", 97, 131, (160, 160), 3), ("tall", 211, 120, (160, 192), 0), ("up", 48, 64, (160, 160), 60),
               ("exact", 160, 160, (160, 160), 1), ("vga", 480, 640, (640, 640), 7)]


def input_case(tag, h, w, k, seed):
    g = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy) % 256, (xx + yy * 5) % 256, g.randint(0, 256, (h, w))], -1).astype(np.uint8)
    targets = np.concatenate([g.randint(0, 80, (k, 1)).astype(np.float64), np.round(g.rand(k, 50), 4)], 1) if k else np.zeros((0, 0))
    return img, targets


def gen_input():
    sys.path.insert(0, ROOT)
    from oracle.sector import resize_linear_u8
    cv2 = sys.modules["cv2"]
    cv2.INTER_LINEAR = 1
    # cv2 is not installed: its INTER_LINEAR resize is stood in for by the oracle's restatement of OpenCV's fixed-point
    # arithmetic (that step stays "parity unpinned"); everything else below is the reference's own code
    cv2.resize = lambda img, dsize, interpolation=None: resize_linear_u8(img, dsize[0], dsize[1])
    sys.path.insert(0, os.path.join(REF, "yolox_24p"))
    da = importlib.import_module("datasets.data_augment")
    tt = da.TrainTransform(max_labels=50)
    store = {}
    for i, (tag, h, w, size, k) in enumerate(INPUT_CASES):
        img, targets = input_case(tag, h, w, k, 140 + i)
        out, r, padded = da.preproc(img, size)
        image_t, labels = tt(img, targets.copy(), size)
        assert np.array_equal(out, image_t)
        store[tag + "_r"] = float(r)
        store[tag + "_crc"] = zlib.crc32(out.tobytes())
        store[tag + "_sub"] = out[:, ::7, ::5]
        store[tag + "_labels"] = labels
        assert labels.dtype == np.float32 and out.dtype == np.float32
    save("g14_input", **store)

# --------------------------------------------------------------------------- G15 swapped backbone (BASELINE config 4)
def gen_resnet(utils, models):
    """YOLOX with ``model.backbone.backbone = resnet50()`` (the switch of yolox/models/yolo_pafpn.py:31-38 applied to the
    24p network): train-mode outputs, gradients of a fixed cotangent, BN running statistics, and the
    eval-mode outputs - weights from synth.fill_state (name-keyed, 35 M values: not stored)."""
    dk = importlib.import_module("models.darknet")
    torch.manual_seed(0)
    backbone = models.YOLOPAFPN(0.33, 1.0, in_channels=[256, 512, 1024], act="silu")
    backbone.backbone = dk.resnet50()
    head = models.YOLOXHead(80, 1.0, in_channels=[256, 512, 1024], act="silu")
    model = models.YOLOX(backbone, head)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    synth.fill_state(model, seed=15)
    model.train()
    B, S = 2, 256
    x = synth.make_images(B, S, seed=151)
    out = model(x, train=True)[3]
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(152)) * torch.tensor([0.05] * 26 + [1.0] * 81)
    (out * gy).sum().backward()
    sd = dict(model.named_parameters())
    store = {"B": B, "S": S, "out": out.detach()[:, ::3], "keys": np.array(sorted(model.state_dict().keys())),
             "n_params": sum(p.numel() for p in model.parameters())}
    for name in ("backbone.backbone.conv1.weight", "backbone.backbone.bn1.weight", "backbone.backbone.layer1.0.downsample.0.weight",
                 "backbone.backbone.layer1.0.conv2.weight", "backbone.backbone.layer2.0.downsample.0.weight",
                 "backbone.backbone.layer2.3.bn3.bias", "backbone.backbone.layer3.5.conv1.weight",
                 "backbone.backbone.layer4.0.conv2.weight", "backbone.backbone.layer4.2.bn3.weight", "backbone.lateral_conv0.conv.weight",
                 "head.stems.0.conv.weight"):
        g = sd[name].grad
        store["g:" + name] = g if g.numel() <= 40000 else g.reshape(-1)[:: g.numel() // 20000 + 1]
        store["gn:" + name] = float(g.double().norm())
    store["unused_grad_is_none"] = int(all(sd[n].grad is None for n in ("backbone.backbone.fc.weight", "backbone.backbone.baseconv1.0.weight")))
    msd = model.state_dict()
    for name in ("backbone.backbone.bn1.running_mean", "backbone.backbone.bn1.running_var", "backbone.backbone.layer3.0.downsample.1.running_var"):
        store["b:" + name] = msd[name]
    model.eval()
    with torch.no_grad():
        store["out_eval"] = model(x, train=False)[:, ::3]
    save("g15_resnet", **store)

DENSE_GRADS = ("backbone.backbone.stem.0.conv.weight", "backbone.backbone.D1.denseblock.0.conv_block.0.bn.weight",
               "backbone.backbone.D1.denseblock.5.conv_block.1.conv.weight", "backbone.backbone.T1.trans.0.conv.weight",
               "backbone.backbone.D2.denseblock.3.conv_block.0.conv.weight", "backbone.backbone.baseconv1.conv.weight",
               "backbone.backbone.D3.denseblock.23.conv_block.1.bn.bias", "backbone.backbone.T3.trans.0.bn.weight",
               "backbone.backbone.D4.denseblock.15.conv_block.1.conv.weight", "backbone.lateral_conv0.conv.weight",
               "head.stems.0.conv.weight")


def gen_densenet(utils, models):
    """YOLOX with ``backbone.backbone = densenet121()``: like G15; the Dropout2d(0.3) draws of the training-mode run are
    recorded (per layer, sample and channel) so that the oracle and the product can replay them."""
    dk = importlib.import_module("models.darknet")
    torch.manual_seed(0)
    backbone = models.YOLOPAFPN(0.33, 1.0, in_channels=[256, 512, 1024], act="silu")
    backbone.backbone = dk.densenet121()
    head = models.YOLOXHead(80, 1.0, in_channels=[256, 512, 1024], act="silu")
    model = models.YOLOX(backbone, head)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    synth.fill_state(model, seed=16)
    model.train()
    keeps = []
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.register_forward_hook(lambda mod, inp, out: keeps.append((out.abs().sum((2, 3)) > 0).float() / (1.0 - mod.p)))
    B, S = 2, 256
    torch.manual_seed(161)
    x = synth.make_images(B, S, seed=162)
    out = model(x, train=True)[3]
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(152)) * torch.tensor([0.05] * 26 + [1.0] * 81)
    (out * gy).sum().backward()
    sd = dict(model.named_parameters())
    keep = torch.stack(keeps)
    assert keep.shape == (58, B, 32) and 0.5 < float((keep > 0).float().mean()) < 0.9
    store = {"B": B, "S": S, "out": out.detach()[:, ::3], "keys": np.array(sorted(model.state_dict().keys())), "keep": keep,
             "n_params": sum(p.numel() for p in model.parameters())}
    for name in DENSE_GRADS:
        g = sd[name].grad
        store["g:" + name] = g if g.numel() <= 40000 else g.reshape(-1)[:: g.numel() // 20000 + 1]
        store["gn:" + name] = float(g.double().norm())
    msd = model.state_dict()
    for name in ("backbone.backbone.stem.0.bn.running_mean", "backbone.backbone.D1.denseblock.2.conv_block.0.bn.running_var",
                 "backbone.backbone.T2.trans.0.bn.running_mean"):
        store["b:" + name] = msd[name]
    model.eval()
    with torch.no_grad():
        store["out_eval"] = model(x, train=False)[:, ::3]
    save("g16_densenet", **store)

VGG_GRADS = ("backbone.backbone.conv_pool1.0.conv.weight", "backbone.backbone.conv_pool1.1.bn.weight", "backbone.backbone.conv_pool3.2.conv.weight",
             "backbone.backbone.conv_pool5.3.bn.bias", "backbone.backbone.conv_add.conv.weight", "backbone.lateral_conv0.conv.weight",
             "head.stems.0.conv.weight")


def gen_vgg(utils, models):
    """YOLOX with ``backbone.backbone = vgg19()`` (the fourth value of the backbone switch; not in BASELINE's configs)."""
    dk = importlib.import_module("models.darknet")
    torch.manual_seed(0)
    backbone = models.YOLOPAFPN(0.33, 1.0, in_channels=[256, 512, 1024], act="silu")
    backbone.backbone = dk.vgg19()
    head = models.YOLOXHead(80, 1.0, in_channels=[256, 512, 1024], act="silu")
    model = models.YOLOX(backbone, head)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    synth.fill_state(model, seed=17)
    model.train()
    B, S = 2, 128
    x = synth.make_images(B, S, seed=171)
    out = model(x, train=True)[3]
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(152)) * torch.tensor([0.05] * 26 + [1.0] * 81)
    (out * gy).sum().backward()
    sd = dict(model.named_parameters())
    store = {"B": B, "S": S, "out": out.detach(), "keys": np.array(sorted(model.state_dict().keys())),
             "n_params": sum(p.numel() for p in model.parameters())}
    for name in VGG_GRADS:
        g = sd[name].grad
        store["g:" + name] = g if g.numel() <= 40000 else g.reshape(-1)[:: g.numel() // 20000 + 1]
        store["gn:" + name] = float(g.double().norm())
    msd = model.state_dict()
    for name in ("backbone.backbone.conv_pool1.0.bn.running_mean", "backbone.backbone.conv_pool1.0.bn.running_var"):
        store["b:" + name] = msd[name]
    model.eval()
    with torch.no_grad():
        store["out_eval"] = model(x, train=False)
    save("g17_vgg", **store)


# --------------------------------------------------------------------------- G18 training curve (round 5)
CURVE = dict(depth=0.33, width=0.25, size=320, batch=4, n_batches=8, steps=300, lr=0.01, momentum=0.9, init_seed=11,
             gts=[[3, 5, 2, 4], [1, 6, 0, 3], [4, 4, 2, 7], [5, 1, 3, 2], [2, 3, 6, 1], [7, 2, 4, 0], [3, 3, 5, 4], [6, 2, 1, 5]])


def curve_data(i):
    """Batch i of the fixed synthetic set (also what tests/test_gpu_curve.py feeds the HIP paths)."""
    c = CURVE
    return (synth.make_images(c["batch"], c["size"], seed=500 + i), synth.make_labels(c["batch"], c["gts"][i], size=c["size"], seed=600 + i))


def curve_init(member):
    """Initial state dict of ensemble member `member`: the oracle network of seed CURVE['init_seed'], every floating-point parameter
    scaled by (1 + 1e-6 N(0,1)) drawn from seed 9000 + member (member 0: unperturbed).  SGD with lr 0.01 on SimOTA's discrete assignment
    is chaotic - two fp32 runs that differ in the seventh digit separate within ten steps and differ by 10 - 25 % in single-step loss
    later on - so 'does bf16 train like fp32' is a statement about ENSEMBLES of runs, and this is how their members are made."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import model as om
    c = CURVE
    torch.manual_seed(c["init_seed"])
    net = om.Net(c["depth"], c["width"])
    if member:
        g = torch.Generator().manual_seed(9000 + member)
        with torch.no_grad():
            for p_ in net.parameters():
                p_.mul_(1.0 + 1e-6 * torch.randn(p_.shape, generator=g))
    return net


def gen_curve(utils, models):
    """G18 (round 5).  300 SGD steps (lr 0.01, momentum 0.9, nesterov: exp/yolox_base.py:120-124) of the REFERENCE's
    own YOLOX-24p (depth 0.33, width 0.25, 320 x 320, batch 4) with the reference's own Loss_Function on a fixed set of 8 synthetic
    batches walked in order, fp32 on the CPU as the reference trains (train_24p.py:80-111): FOUR ensemble members (curve_init 0..3) -
    and member 0 once more through the oracle (oracle.model.Net, LossOracle, sgd_nesterov_step).  The initial parameters are the oracle
    network's (its construction order differs from the reference's in the head, so the state dict is loaded into the reference model -
    same keys), which lets the GPU test rebuild them without a 9 MB fixture.  Stored: the loss curves [4, 300] + [300], member
    0's three components and num_fg / num_gt per step."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import model as om
    from oracle.loss import LossOracle
    c = CURVE
    in_ch = [256, 512, 1024]
    data = [curve_data(i) for i in range(c["n_batches"])]

    def bn_cfg(mod):
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03                  # Exp.get_model (exp/yolox_base.py:58-62)

    r_loss, r_parts = [], []
    for member in range(4):
        net = curve_init(member)
        ref = models.YOLOX(models.YOLOPAFPN(c["depth"], c["width"], in_channels=in_ch, act="silu"), models.YOLOXHead(80, c["width"], in_channels=in_ch, act="silu"))
        ref.load_state_dict(net.state_dict(), strict=True)
        bn_cfg(ref)
        ref.train()
        lf = models.Loss_Function(80)
        opt = torch.optim.SGD(ref.parameters(), lr=c["lr"], momentum=c["momentum"], nesterov=True)
        cur = []
        for step in range(c["steps"]):
            imgs, labs = data[step % c["n_batches"]]
            opt.zero_grad()
            tup = lf.forward(ref(imgs, train=True), labs)
            tup[0].backward()
            opt.step()
            cur.append(float(tup[0].detach()))
            if member == 0:
                r_parts.append([float(tup[1].detach().sum()), float(tup[2].detach()), float(tup[3].detach()), float(tup[5])])
            if step % 100 == 0:
                print("reference member %d step %3d loss %.4f" % (member, step, cur[-1]))
        r_loss.append(cur)
    # ---- the oracle, member 0
    net = curve_init(0)
    bn_cfg(net)
    net.train()
    ora = LossOracle(80)
    params = list(net.parameters())
    bufs = [None] * len(params)
    o_loss = []
    for step in range(c["steps"]):
        imgs, labs = data[step % c["n_batches"]]
        for p_ in params:
            p_.grad = None
        tup = ora(net(imgs, train=True), labs)
        tup[0].backward()
        om.sgd_nesterov_step(params, bufs, c["lr"], c["momentum"])
        o_loss.append(float(tup[0].detach()))
        if step % 100 == 0:
            print("oracle    member 0 step %3d loss %.4f" % (step, o_loss[-1]))
    save("g18_train_curve", ref_loss=np.asarray(r_loss, dtype=np.float32), ref_parts=np.asarray(r_parts, dtype=np.float32),
         oracle_loss=np.asarray(o_loss, dtype=np.float32),
         config=np.asarray([c["depth"], c["width"], c["size"], c["batch"], c["n_batches"], c["steps"], c["lr"], c["momentum"], c["init_seed"]], dtype=np.float64),
         gts=np.asarray(c["gts"], dtype=np.int64))


def write_manifest():
    """tests/golden/MANIFEST.json: shape, dtype and CRC-32 of every array of every committed fixture (tests/test_golden_manifest.py)."""
    import json
    import zlib
    man = {}
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            z = np.load(os.path.join(HERE, f), allow_pickle=False)
            man[f] = {k: [list(z[k].shape), str(z[k].dtype), zlib.crc32(np.ascontiguousarray(z[k]).tobytes())] for k in sorted(z.files)}
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=0, sort_keys=True)
    print("wrote MANIFEST.json: %d files, %d arrays" % (len(man), sum(len(v) for v in man.values())))


if __name__ == "__main__":
    if sys.argv[1:] == ["--manifest"]:
        write_manifest()
        sys.exit(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["geometry", "assign", "model", "depthwise", "sector", "post", "n2", "labels", "input", "resnet", "densenet", "vgg"]
    utils, models = load_reference()
    if "geometry" in which:
        gen_geometry(utils, models)
    if "assign" in which:
        gen_assign(utils, models)
    if "model" in which:
        gen_model(utils, models)
    if "depthwise" in which:
        gen_depthwise(utils, models)
    if "sector" in which:
        gen_sector()
    if "post" in which:
        gen_post(utils, models)
    if "n2" in which:
        gen_n2(utils, models)
    if "labels" in which:
        gen_labels()
    if "input" in which:
        gen_input()
    if "resnet" in which:
        gen_resnet(utils, models)
    if "densenet" in which:
        gen_densenet(utils, models)
    if "vgg" in which:
        gen_vgg(utils, models)
    if "curve" in which:                                       # not in the default list: two 300-step CPU trainings (~3 min)
        gen_curve(utils, models)
