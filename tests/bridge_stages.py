"""Shared by tests/test_gpu_fullsize.py and tests/probes/bridge_probe.py: the bf16 bridge between the product's kernels and the restatement,
stage by stage.  Test infrastructure (imports the oracle).

The product (YOLOX-l, bf16 storage) and the oracle in its bf16-STORAGE-emulating mode round at the same points and accumulate in
fp32; what differs is the accumulation order, i.e. now and then one rounding falls the other way (1 bf16 ulp = 2^-8 relative).
Stages: stem, dark2..dark5 (models/darknet.py:95-177), the two halves of the PAFPN (yolo_pafpn.py:83-124), the head
(yolo_head_24p.py:143-210).  Two runs:

  * TEACHER-FORCED: every product stage gets the ORACLE's stage input, so a stage's figure is that stage's own kernels' doing;
  * CHAINED: every product stage gets the product's own previous output, as in the real forward - the drift table.

A-priori bound of a teacher-forced stage with n conv units (the model, not a fit): a unit stores two tensors (raw conv output, activated
output); if the two paths' roundings were fully decorrelated each store would differ by a uniform rounding error on either side,
relative rms sqrt(2) * 2^-9 / sqrt(3) = 1.6e-3, two stores per unit 2.3e-3, n units in quadrature 2.3e-3 * sqrt(n), and BatchNorm
over >= 8 000 values per channel (B = 20) passes relative noise on with gain ~1; STAGE_GAIN = 2 allows for residual sums and SiLU's
slope above 1.  So: rms(err) <= 2 * 2.3e-3 * sqrt(n) * rms(ref), and as a second, cruder statement max|err| <= 1e-2 * max|ref|
(2.5 bf16 ulp of the stage's range).
"""
import math

import torch
import torch.nn.functional as F

UNITS = {"stem": 1, "dark2": 10, "dark3": 22, "dark4": 22, "dark5": 12, "neck_top_down": 20, "neck_bottom_up": 20, "head": 4}
STAGE_GAIN, UNIT_RMS = 2.0, 2.3e-3


def rms_bound(stage):
    return STAGE_GAIN * UNIT_RMS * math.sqrt(UNITS[stage])


def build_pair(seed=3, dev="cuda:0"):
    from ep24 import nn as enn
    from oracle import model as om
    torch.manual_seed(seed)
    ref = om.Net(1.0, 1.0)
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.uniform_(mod.bias, -0.2, 0.2)
    m = enn.YOLOX(enn.YOLOPAFPN(1.0, 1.0), enn.YOLOXHead(80, 1.0))
    m.load_state_dict(ref.state_dict(), strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return ref, m.to(dev)


def oracle_stages(ref, x):
    """-> {stage: (inputs, outputs)} of the oracle in bf16-storage mode (tensors on the host)."""
    from oracle import model as om
    om.EMULATE_BF16 = True
    try:
        ref.train()
        with torch.no_grad():
            bb, nk, hd = ref.backbone.backbone, ref.backbone, ref.head
            st = {}
            s0 = bb.stem(x); st["stem"] = ((x,), (s0,))
            d2 = bb.dark2(s0); st["dark2"] = ((s0,), (d2,))
            d3 = bb.dark3(d2); st["dark3"] = ((d2,), (d3,))
            d4 = bb.dark4(d3); st["dark4"] = ((d3,), (d4,))
            d5 = bb.dark5(d4); st["dark5"] = ((d4,), (d5,))
            f0 = nk.lateral_conv0(d5)
            p4 = nk.C3_p4(torch.cat([F.interpolate(f0, scale_factor=2, mode="nearest"), d4], 1))
            f1 = nk.reduce_conv1(p4)
            o2 = nk.C3_p3(torch.cat([F.interpolate(f1, scale_factor=2, mode="nearest"), d3], 1))
            st["neck_top_down"] = ((d3, d4, d5), (f0, f1, o2))
            o1 = nk.C3_n3(torch.cat([nk.bu_conv2(o2), f1], 1))
            o0 = nk.C3_n4(torch.cat([nk.bu_conv1(o1), f0], 1))
            st["neck_bottom_up"] = ((f0, f1, o2), (o1, o0))
            out = hd((o2, o1, o0), train=True)[3]
            st["head"] = ((o2, o1, o0), (out,))
    finally:
        om.EMULATE_BF16 = False
    return st


def product_stage(m, stage, inputs):
    """The product's modules of one stage on the given inputs (NCHW fp32 tensors on the GPU); glue (upsample, concat) is exact."""
    bb, nk, hd = m.backbone.backbone, m.backbone, m.head
    with torch.no_grad():
        if stage == "stem":
            return (bb.stem(inputs[0]),)
        if stage in ("dark2", "dark3", "dark4", "dark5"):
            y = inputs[0]
            for mod in getattr(bb, stage):
                y = mod(y)
            return (y,)
        if stage == "neck_top_down":
            d3, d4, d5 = inputs
            f0 = nk.lateral_conv0(d5)
            p4 = nk.C3_p4(torch.cat([F.interpolate(f0, scale_factor=2, mode="nearest"), d4], 1))
            f1 = nk.reduce_conv1(p4)
            return (f0, f1, nk.C3_p3(torch.cat([F.interpolate(f1, scale_factor=2, mode="nearest"), d3], 1)))
        if stage == "neck_bottom_up":
            f0, f1, o2 = inputs
            o1 = nk.C3_n3(torch.cat([nk.bu_conv2(o2), f1], 1))
            return (o1, nk.C3_n4(torch.cat([nk.bu_conv1(o1), f0], 1)))
        if stage == "head":
            return (hd(list(inputs), train=True)[3],)
    raise KeyError(stage)


ORDER = ["stem", "dark2", "dark3", "dark4", "dark5", "neck_top_down", "neck_bottom_up", "head"]
FEEDS = {"stem": None, "dark2": [("stem", 0)], "dark3": [("dark2", 0)], "dark4": [("dark3", 0)], "dark5": [("dark4", 0)],
         "neck_top_down": [("dark3", 0), ("dark4", 0), ("dark5", 0)], "neck_bottom_up": [("neck_top_down", 0), ("neck_top_down", 1), ("neck_top_down", 2)],
         "head": [("neck_top_down", 2), ("neck_bottom_up", 0), ("neck_bottom_up", 1)]}


def head_view(t):
    """Decoded head outputs in the units errors are meaningful in: centres, LOG radii, logits."""
    return torch.cat([t[..., :2], torch.log(t[..., 2:26]), t[..., 26:]], -1)


def errors(got, want):
    got, want = got.float().cpu(), want.float()
    d = got - want
    return float((d.pow(2).mean() / want.pow(2).mean()).sqrt()), float(d.abs().max() / want.abs().max())


def bridge_table(ref, m, x, dev="cuda:0"):
    """-> rows (stage, units, teacher-forced rms, teacher-forced max / range, chained rms, chained max / range)."""
    st = oracle_stages(ref, x)
    rows, chained = [], {}
    for stage in ORDER:
        ins_ref, outs_ref = st[stage]
        tf = product_stage(m, stage, [t.to(dev) for t in ins_ref])
        ch_in = [x.to(dev)] if FEEDS[stage] is None else [chained[s][i] for s, i in FEEDS[stage]]
        ch = product_stage(m, stage, ch_in)
        chained[stage] = ch
        view = head_view if stage == "head" else (lambda t: t)
        e_tf = [errors(view(a), view(b)) for a, b in zip(tf, outs_ref)]
        e_ch = [errors(view(a), view(b)) for a, b in zip(ch, outs_ref)]
        rows.append((stage, UNITS[stage], max(e[0] for e in e_tf), max(e[1] for e in e_tf), max(e[0] for e in e_ch), max(e[1] for e in e_ch)))
    return rows, chained["head"][0]
