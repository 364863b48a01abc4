"""Shared by tests/test_gpu_fullsize.py and tests/probes/bridge_probe.py: the bf16 bridge between the product's kernels and the restatement,
stage by stage.  Test infrastructure (imports the oracle).

The product (YOLOX-l, bf16 storage) and the oracle in its bf16-STORAGE-emulating mode round at the same points and accumulate in
fp32; what differs is the accumulation order, i.e. now and then one rounding falls the other way (1 bf16 ulp = 2^-8 relative).
Stages: stem, dark2..dark5 (models/darknet.py:95-177), the two halves of the PAFPN (yolo_pafpn.py:83-124), the head
(yolo_head_24p.py:143-210).  Two runs:

  * TEACHER-FORCED: every product stage gets the ORACLE's stage input, so a stage's figure is that stage's own kernels' doing;
  * CHAINED: every product stage gets the product's own previous output, as in the real forward - the drift table.  A chained
    stage's error is its own (teacher-forced) error plus what it makes of its inputs' errors; the second part is PREDICTED from the
    oracle alone: the oracle stage is run once more on its own inputs perturbed by random relative noise of exactly the chained input
    errors' rms, and the relative rms change of its outputs is what the stage passes on.  Asserted: chained <= teacher-forced bound
    + PROP_SLACK x predicted (the deviation between the two paths is rounding flips, not white noise: PROP_SLACK = 2 allows for that).
    A kernel whose own error doubles fails its teacher-forced row; a stage that amplifies more than the network itself does fails its
    chained row - neither bound is read off the product's output.

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
STAGE_GAIN, UNIT_RMS, PROP_SLACK = 2.0, 2.3e-3, 2.0


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


def oracle_stage(ref, stage, inputs):
    """One stage of the oracle in bf16-storage mode on the given inputs (host tensors) -> tuple of outputs."""
    from oracle import model as om
    bb, nk, hd = ref.backbone.backbone, ref.backbone, ref.head
    om.EMULATE_BF16 = True
    try:
        ref.train()
        with torch.no_grad():
            if stage == "stem":
                return (bb.stem(inputs[0]),)
            if stage in ("dark2", "dark3", "dark4", "dark5"):
                return (getattr(bb, stage)(inputs[0]),)
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
                return (hd(tuple(inputs), train=True)[3],)
    finally:
        om.EMULATE_BF16 = False
    raise KeyError(stage)


def oracle_stages(ref, x):
    """-> {stage: (inputs, outputs)} of the oracle chained through the whole network (tensors on the host)."""
    st, outs = {}, {}
    for stage in ORDER:
        ins = (x,) if FEEDS[stage] is None else tuple(outs[s][i] for s, i in FEEDS[stage])
        outs[stage] = oracle_stage(ref, stage, ins)
        st[stage] = (ins, outs[stage])
    return st


def propagated(ref, stage, ins_ref, outs_ref, in_errs, seed=0):
    """What a RANDOM relative perturbation of rms in_errs[i] on input i does to the oracle stage's outputs (max over the outputs of the
    relative rms change): the part of a chained stage's error that it merely passes on (and amplifies) from its inputs."""
    g = torch.Generator().manual_seed(seed)
    pert = tuple(t * (1.0 + e * torch.randn(t.shape, generator=g)) for t, e in zip(ins_ref, in_errs))
    outs = oracle_stage(ref, stage, pert)
    view = head_view if stage == "head" else (lambda t: t)
    return max(errors(view(a), view(b))[0] for a, b in zip(outs, outs_ref))


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


def bridge_table(ref, m, x, dev="cuda:0", with_propagation=True):
    """-> rows (stage, units, teacher-forced rms, teacher-forced max / range, chained rms, chained max / range, propagated rms), the
    product's chained head output and the oracle's.  propagated = what random perturbations of the size of the chained INPUT errors do to the oracle
    stage (None for the stem, whose input is exact)."""
    st = oracle_stages(ref, x)
    rows, chained, ch_err = [], {}, {}
    for stage in ORDER:
        ins_ref, outs_ref = st[stage]
        tf = product_stage(m, stage, [t.to(dev) for t in ins_ref])
        ch_in = [x.to(dev)] if FEEDS[stage] is None else [chained[s][i] for s, i in FEEDS[stage]]
        ch = product_stage(m, stage, ch_in)
        chained[stage] = ch
        view = head_view if stage == "head" else (lambda t: t)
        e_tf = [errors(view(a), view(b)) for a, b in zip(tf, outs_ref)]
        e_ch = [errors(view(a), view(b)) for a, b in zip(ch, outs_ref)]
        ch_err[stage] = [errors(a, b)[0] for a, b in zip(ch, outs_ref)]          # per output, in the tensor's own units (feeds the next stage)
        prop = None
        if with_propagation and FEEDS[stage] is not None:
            prop = propagated(ref, stage, ins_ref, outs_ref, [ch_err[s][i] for s, i in FEEDS[stage]])
        rows.append((stage, UNITS[stage], max(e[0] for e in e_tf), max(e[1] for e in e_tf), max(e[0] for e in e_ch), max(e[1] for e in e_ch), prop))
    return rows, chained["head"][0], st["head"][1][0]
