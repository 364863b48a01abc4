"""Static execution plan of the YOLOX-24p network on one MI355X.

The reference runs ``YOLOX.forward`` / autograd as ~131 cuDNN convs plus hundreds of small aten kernels
(yolox_24p/models/yolox.py:24-34, yolo_pafpn.py:83-124, darknet.py:165-177, yolo_head_24p.py:143-237).
Here the module tree is lowered ONCE per (batch, size) into two flat lists of C-ABI launches (forward,
backward) over pre-allocated NHWC bf16 buffers:

* every conv is the MFMA implicit GEMM (``ep24_conv_fwd_bf16`` / ``_dgrad`` / ``_wgrad``), BN batch statistics
  are accumulated in the conv epilogue, BN-apply + SiLU (+ residual) is one streaming kernel that writes
  straight into its consumer's concat slot (torch.cat never materialises);
* gradients of activations live in mirror buffers; whether a producer overwrites or accumulates is decided
  statically while the backward list is built, residual adds alias their gradient buffers;
* all parameters live in ONE flat fp32 buffer (conv weights physically [Cout][kh][kw][Cin], i.e. torch
  channels_last) with flat gradient / momentum twins, so the optimizer and the RCCL all-reduce are single
  contiguous passes; ``nn.Parameter`` objects are re-homed as views so state_dict keys and shapes stay the
  reference's.

The lists contain no host synchronisation, no allocation and no shape-dependent Python, so a training step
is captured into a hipGraph (``ep24.train.TrainStep``).
"""
import torch

from . import _lib, nn as enn
from ._lib import call, ptr, stream_ptr
from .options import PlanOptions, get_options

BF16 = torch.bfloat16
WGRAD_REDUCE_GROUP = 16        # layers per weight-gradient reduce launch
WGRAD_GROUP_MAX = 16           # problems per grouped weight-gradient launch (csrc/conv_wgrad.hip WG_MAX: the table travels as kernel arguments)
WGRAD_GROUP_FILL = 0.88        # a class is launched once its (tile, split) grid fills this share of the chip's resident slots
WGRAD_GROUP_TAIL = 0.93        # the last builders of backward (dark2, the stem) launch their weight gradients one by one again:
                               # what waits in a group there is exposed behind the main lane's last kernel
STATS_REPLICAS = 8
STRIDES = (8, 16, 32)


def _r8(n):
    return (n + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------ buffers
class Buf:
    """Row-major [rows, ld] bf16 buffer with a lazily created gradient twin."""

    def __init__(self, dev, rows, ld, dtype=BF16):
        self.rows, self.ld, self.dtype = rows, ld, dtype
        self.esize = 4 if dtype == torch.float32 else 2
        self.t = torch.zeros(rows * ld, dtype=dtype, device=dev)
        self.g = None
        self.gwritten = []                       # channel intervals of the gradient already produced

    def grad(self):
        if self.g is None:
            self.g = torch.zeros(self.rows * self.ld, dtype=self.dtype, device=self.t.device)
        return self.g


class Dyn:
    """Launch argument resolved when the list RUNS (input / incoming-gradient pointers)."""

    def __init__(self, table, key):
        self.table, self.key = table, key

    def get(self):
        return self.table[self.key]


_PENDING_GW = []


class Act:
    """Channel slice [c0, c0+C) of a Buf, viewed as an NHWC tensor [B,H,W,C]."""

    def __init__(self, buf, c0, C, B, H, W):
        self.buf, self.c0, self.C, self.B, self.H, self.W = buf, c0, C, B, H, W
        self._alias = None                       # gradient shares storage with this Act's gradient
        self.needs_grad = True

    @property
    def M(self):
        return self.B * self.H * self.W

    @property
    def ld(self):
        return self.buf.ld

    def ptr(self):
        return self.buf.t.data_ptr() + self.buf.esize * self.c0

    def _groot(self):
        a = self
        while a._alias is not None:
            a = a._alias
        return a

    @property
    def gld(self):
        return self._groot().buf.ld

    def gptr(self):
        r = self._groot()
        return r.buf.grad().data_ptr() + r.buf.esize * r.c0

    def slice(self, c0, C):
        assert self._alias is None
        a = Act(self.buf, self.c0 + c0, C, self.B, self.H, self.W)
        if getattr(self, "bn_info", None) is not None:       # a channel range of a unit's output: the range of its BatchNorm
            a.bn_info, a.bn_c0 = self.bn_info, getattr(self, "bn_c0", 0) + c0
        return a

    def alias_grad(self, other):
        """d(self) shares storage with d(other): used by y = f(x) + x, where dx = dy + f'(...)."""
        assert self.C == other.C and self._alias is None
        self._alias = other

    def _interval(self):
        r = self._groot()
        return r.buf.gwritten, r.c0, r.c0 + r.C

    def gregion(self):
        """(gradient buffer identity, first channel, one past the last) of this Act's gradient."""
        r = self._groot()
        return (id(r.buf), r.c0, r.c0 + r.C)

    def gwrite(self):
        """Called while the backward list is built: 0 = first producer (overwrite), 1 = accumulate."""
        _PENDING_GW.append(self.gregion())           # picked up by the next Engine._b(): that entry writes this region
        w, lo, hi = self._interval()
        if any(a <= lo and hi <= b for a, b in w):
            return 1
        assert not any(not (hi <= a or b <= lo) for a, b in w), "partially written gradient region"
        w.append((lo, hi))
        w.sort()
        merged = [w[0]]
        for a, b in w[1:]:                            # adjacent producers make one region (a merged CSP unit reads [x_2 | x_1])
            if a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        w[:] = merged
        return 0

    def gready(self):
        w, lo, hi = self._interval()
        return any(a <= lo and hi <= b for a, b in w)


# ------------------------------------------------------------------------------------------------ parameters
class ConvSeg:
    def __init__(self, params, cout, taps, cin, need_dgrad=True):
        self.params, self.cout, self.taps, self.cin, self.need_dgrad = params, cout, taps, cin, need_dgrad
        self.cin_pad, self.cout_pad = _r8(cin), _r8(cout)
        self.off = self.wf_off = self.wd_off = None

    @property
    def numel(self):
        return self.cout * self.taps * self.cin


class VecSeg:
    def __init__(self, params):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.off = None


class ParamHome:
    """Flat fp32 parameter / gradient / momentum buffers + packed bf16 weight copies for one model.  ``options``
    (ep24.options.PlanOptions) decides which units are merged, i.e. the layout; every plan of the model shares it."""

    def __init__(self, model, options=None):
        self.options = options if options is not None else get_options(model)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.Ep24Error("ep24: move the model to the GPU before running it (model.to('cuda'))")
        self.dev = dev
        self.convs, self.vecs, self.order, self.by_param = [], [], [], {}
        self.merged_bn = []                               # (bn of conv2, bn of conv1) of the merged CSP units
        stems = {m.conv for m in model.modules() if isinstance(m, enn.Focus)}
        for mod in exec_order(model, self.options):
            if isinstance(mod, tuple) and mod[0] == "unit":          # (conv, bn) pair of a swapped backbone
                _, conv, bn_, stem = mod
                w = conv.weight
                k = w.shape[2]
                if stem:                                             # im2col rows x [Cout][k*k*Cin]
                    self._add(ConvSeg([w], w.shape[0], 1, k * k * w.shape[1], need_dgrad=False))
                else:
                    self._add(ConvSeg([w], w.shape[0], k * k, w.shape[1]))
                self._add(VecSeg([bn_.weight]))
                self._add(VecSeg([bn_.bias]))
            elif isinstance(mod, tuple) and mod[0] == "unused":      # in the state dict, not in the graph: gradient stays 0
                self._add(VecSeg([mod[1]]))
            elif isinstance(mod, tuple) and mod[0] in ("csp_merged", "pair_merged"):
                # two BaseConv units over the same input as one GEMM + one BN launch; (first, second) = channel order of the output
                c2, c1 = (mod[1].conv2, mod[1].conv1) if mod[0] == "csp_merged" else (mod[1], mod[2])
                w2, w1 = c2.conv.weight, c1.conv.weight
                self._add(ConvSeg([w2, w1], w2.shape[0] + w1.shape[0], w2.shape[2] * w2.shape[3], w2.shape[1]))
                self._add(VecSeg([c2.bn.weight, c1.bn.weight]))
                self._add(VecSeg([c2.bn.bias, c1.bn.bias]))
                self.merged_bn.append((c2.bn, c1.bn))
            elif isinstance(mod, tuple) and mod[0] == "conv":        # a conv on its own (DenseNet: BN sits in front of it)
                w = mod[1].weight
                k = w.shape[2]
                self._add(ConvSeg([w], w.shape[0], 1, k * k * w.shape[1], need_dgrad=False) if mod[2] else
                          ConvSeg([w], w.shape[0], k * k, w.shape[1]))
            elif isinstance(mod, tuple) and mod[0] == "bn":
                self._add(VecSeg([mod[1].weight]))
                self._add(VecSeg([mod[1].bias]))
            elif isinstance(mod, enn.BaseConv):
                w = mod.conv.weight
                k = w.shape[2]
                if mod in stems:      # Focus stem runs as im2col x [Cout][108]: one "tap" of 108 (kh,kw,c) columns
                    self._add(ConvSeg([w], w.shape[0], 1, k * k * w.shape[1], need_dgrad=False))
                elif mod.conv.groups > 1:                  # depthwise (DWConv.dconv): [C][9] fp32, the kernels read the master itself
                    self._add(ConvSeg([w], w.shape[0], k * k, 1, need_dgrad=False))
                else:
                    self._add(ConvSeg([w], w.shape[0], k * k, w.shape[1]))
                self._add(VecSeg([mod.bn.weight]))
                self._add(VecSeg([mod.bn.bias]))
            elif isinstance(mod, enn.YOLOXHead):
                for k in range(len(mod.stems)):
                    rw, ow, cw = mod.reg_preds[k].weight, mod.obj_preds[k].weight, mod.cls_preds[k].weight
                    # reg(26) and obj(1) predictors read the same feature: one [27][h] GEMM, adjacent in the flat buffer
                    self._add(ConvSeg([rw, ow], rw.shape[0] + ow.shape[0], 1, rw.shape[1]))
                    self._add(VecSeg([mod.reg_preds[k].bias, mod.obj_preds[k].bias]))
                    self._add(ConvSeg([cw], cw.shape[0], 1, cw.shape[1]))
                    self._add(VecSeg([mod.cls_preds[k].bias]))
        missing = [n for n, p in model.named_parameters() if p not in self.by_param]
        if missing:
            raise _lib.Ep24Error("ep24: parameters outside the supported graph: %s" % missing[:4])
        n = wf = wd = 0
        for seg in self.order:                      # execution order: backward completes the buffer from its tail
            seg.off = n
            n += (seg.numel + 63) // 64 * 64        # 64-element alignment: a group of the update lies in ONE segment (wf_delta below)
        for seg in self.convs:
            seg.wf_off, seg.wd_off = wf, wd
            # 64-element steps like the masters: the fused update converts whole 4-element groups, and the alignment padding behind
            # a master whose size is not a multiple of 64 (zeros) lands in this copy's own padding, not in the next layer's weights
            wf += (seg.cout * seg.taps * seg.cin_pad + 63) // 64 * 64
            wd += seg.cin * seg.taps * seg.cout_pad if seg.need_dgrad else 0
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.mflat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.wf = torch.zeros(max(wf, 8), dtype=BF16, device=dev)
        self.wd = torch.zeros(max(wd, 8), dtype=BF16, device=dev)
        self.first_flag = torch.ones(1, dtype=torch.int32, device=dev)
        # Round 5: the fused update writes the packed forward copy itself (csrc/elementwise.hip sgd_kernel).  One int32 per 64 flat
        # elements: (offset of the element's segment in wf) - (its offset in flat) for a conv weight whose Cin is a multiple of 8
        # (its [Cout][T][Cin] master IS the packed layout), INT32_MIN otherwise (BatchNorm vectors, biases, padded-Cin convs: the
        # Focus stem's 108 -> 112 columns).  `wf_current`: the packed copy follows the masters; anything that writes parameters
        # other than the fused update (load_state_dict, a user's p.data.copy_) must clear it - mark_weights_changed() - and the
        # next step packs everything once.
        delta = torch.full((max(n // 64, 1),), -2 ** 31, dtype=torch.int32)
        self.pack_rest = []                               # conv segments the update cannot keep current: packed every step
        for seg in self.convs:
            if seg.cin_pad == seg.cin:
                delta[seg.off // 64:(seg.off + seg.numel + 63) // 64] = seg.wf_off - seg.off
            else:
                self.pack_rest.append(seg)
        self.wf_delta = delta.to(dev)
        self.wf_current = False
        self._rest_tables = None
        if self.pack_rest:                                # descriptor tables of the packing kernel for just those segments (built now:
            rows, pref, tpref = [], [0], [0]              # the call sits inside captured graphs)
            for seg in self.pack_rest:
                rows.append([seg.off, seg.wf_off, seg.wd_off if seg.need_dgrad else -1, seg.cout, seg.taps, seg.cin, seg.cin_pad, seg.cout_pad])
                pref.append(pref[-1] + seg.numel)
                tpref.append(tpref[-1] + seg.taps * ((seg.cout + 63) // 64) * ((seg.cin + 63) // 64))
            t = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
            self._rest_tables = (t(rows), t(pref), t(tpref), len(rows), pref[-1], tpref[-1])
        # BatchNorm running statistics in one flat buffer too (module buffers become views): ModelEMA averages every
        # floating-point state_dict entry (utils/ema.py:55-60), i.e. these next to the parameters, in two launches
        paired = {id(b) for pair in self.merged_bn for b in pair}
        groups = [[m] for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d) and id(m) not in paired]
        groups += [list(pair) for pair in self.merged_bn]     # [rm2 | rm1 | rv2 | rv1]: one BN launch covers both
        nb = sum((b.num_features + 3) // 4 * 8 for g in groups for b in g)
        self.bflat = torch.zeros(max(nb, 4), dtype=torch.float32, device=dev)
        self.bnumel = nb
        o = 0
        with torch.no_grad():
            for g in groups:
                for name in ("running_mean", "running_var"):
                    for b in g:
                        c = b.num_features
                        assert len(g) == 1 or c % 4 == 0
                        v = self.bflat[o:o + c]
                        v.copy_(getattr(b, name))
                        setattr(b, name, v)
                        o += (c + 3) // 4 * 4
        rows, pref, tpref = [], [0], [0]
        for seg in self.convs:
            rows.append([seg.off, seg.wf_off, seg.wd_off if seg.need_dgrad else -1, seg.cout, seg.taps, seg.cin, seg.cin_pad, seg.cout_pad])
            pref.append(pref[-1] + seg.numel)
            tpref.append(tpref[-1] + seg.taps * ((seg.cout + 63) // 64) * ((seg.cin + 63) // 64))
        self.pack_tprefix = torch.tensor(tpref, dtype=torch.int64, device=dev)
        self.pack_tiles = tpref[-1]
        # lookup tables of the packing kernels: segment of every 4096-element chunk / of every transpose tile (built once)
        import bisect
        self.pack_chunk_seg = torch.tensor([bisect.bisect_right(pref, c * 4096) - 1 for c in range((pref[-1] + 4095) // 4096)] or [0],
                                           dtype=torch.int32, device=dev)
        tseg = []
        for si in range(len(self.convs)):
            tseg += [si] * (tpref[si + 1] - tpref[si])
        self.pack_tile_seg = torch.tensor(tseg or [0], dtype=torch.int32, device=dev)
        self.pack_desc = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.pack_prefix = torch.tensor(pref, dtype=torch.int64, device=dev)
        self.pack_total = pref[-1]
        for sub in model.modules():                       # load_state_dict writes parameters behind the fused update's back
            if not getattr(sub, "_ep24_pack_hook", False):
                sub.register_load_state_dict_post_hook(_weights_loaded)
                sub._ep24_pack_hook = True
        self.views = {}                                   # param -> (flat view, grad view, momentum view)
        for sub in model.modules():                       # plans / homes of parts that ran on their own before are stale now
            if sub is not model:
                sub.__dict__.pop("_ep24_home", None)
                sub.__dict__.pop("_ep24_sub", None)
        with torch.no_grad():
            for seg in self.order:
                o = seg.off
                for p in seg.params:
                    m = p.numel()
                    vs = []
                    for base in (self.flat, self.gflat, self.mflat):
                        v = base[o:o + m]
                        if p.dim() == 4:                  # physical [Cout][kh][kw][Cin], logical [Cout,Cin,kh,kw]
                            v = v.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
                        else:
                            v = v.view(p.shape)
                        vs.append(v)
                    vs[0].copy_(p.data)
                    p.data = vs[0]
                    p.grad = vs[1]
                    p._ep24_home = self
                    self.views[p] = tuple(vs)
                    o += m

    def _add(self, seg):
        (self.convs if isinstance(seg, ConvSeg) else self.vecs).append(seg)
        self.order.append(seg)
        for p in seg.params:
            self.by_param[p] = seg

    def bind_grads(self):
        """Undo optimizer.zero_grad(set_to_none=True): .grad must stay the flat view the kernels write."""
        for p, (_, g, _) in self.views.items():
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def pack(self, which=0):
        """fp32 masters -> bf16 [Cout][T][Cin] (forward) and [Cin][T][Cout_pad] (dgrad) copies; ``which`` = 1 / 2: only the first /
        only the second (a captured step packs the dgrad copy on the other stream beside the start of the forward pass)."""
        call("pack_weights_batched", ptr(self.flat), ptr(self.pack_desc), ptr(self.pack_prefix), ptr(self.pack_tprefix),
             len(self.convs), ptr(self.wf), ptr(self.wd), self.pack_total, self.pack_tiles, ptr(self.pack_chunk_seg),
             ptr(self.pack_tile_seg), which, stream_ptr())

    def mark_weights_changed(self):
        """Parameters were written by something other than the fused update: the packed forward copy is stale until the next pack."""
        self.wf_current = False

    def pack_rest_forward(self):
        """The forward copies the fused update does not write (padded-Cin convs: the Focus stem, a few KB): the packing kernel
        over a descriptor table of just those segments."""
        if not self.pack_rest:
            return
        desc, pref, tpref, n, total, tiles = self._rest_tables
        call("pack_weights_batched", ptr(self.flat), ptr(desc), ptr(pref), ptr(tpref), n, ptr(self.wf), ptr(self.wd), total, tiles,
             None, None, 1, stream_ptr())

    def zero_grad(self):
        call("memset_zero", ptr(self.gflat), self.numel * 4, stream_ptr())

    def sgd(self, lr, momentum, grad_scale=1.0):
        call("sgd_nesterov", ptr(self.flat), ptr(self.gflat), ptr(self.mflat), self.numel, float(lr), float(momentum),
             float(grad_scale), ptr(self.first_flag), stream_ptr())

    def sgd_hp(self, hp, ema_home=None, lo=0, hi=None, last=True):
        """The same update with lr / momentum / grad_scale (and the EMA decay) read from the device block ``hp``; with
        ``ema_home`` the EMA copy of the parameters and of the BatchNorm running statistics is advanced as well.
        ``lo`` / ``hi``: only elements [lo, hi) of the flat buffers (ep24.train updates the parameters whose gradients are
        complete while backward still runs); ``last``: this call finishes the step."""
        if ema_home is not None and (ema_home.numel, ema_home.bnumel) != (self.numel, self.bnumel):
            raise _lib.Ep24Error("ep24: the EMA model's parameter layout differs from the trained model's")
        hi = self.numel if hi is None else hi
        if hi > lo:
            call("sgd_nesterov_hp_range_pack", ptr(self.flat), ptr(self.gflat), ptr(self.mflat), lo, hi - lo, ptr(hp), ptr(self.first_flag),
                 ptr(ema_home.flat) if ema_home is not None else None, 1 if last else 0, ptr(self.wf_delta), ptr(self.wf), stream_ptr())
        if last and ema_home is not None and self.bnumel:
            call("ema_update", ptr(ema_home.bflat), ptr(self.bflat), self.bnumel, 0.0, 0.0, ptr(hp), stream_ptr())


def wgrad_group_splits(tiles, steps, cls, min_steps=120):
    """Pixel splits of the problems of one grouped weight-gradient launch (ep24_conv_wgrad_group_bf16).  tiles[i]: (Cout, tap, Cin)
    tiles of problem i, steps[i]: its 64-pixel steps, cls: the tile class (bit 0 / bit 1: 64-wide over Cout / Cin).  Every workgroup
    keeps >= min_steps steps (its prologue and 64-KB epilogue cost ~12), and the largest common split cap K is taken whose
    (tile, split) grid still fits the chip's resident workgroup slots - one round (2 / 3 / 4 workgroups per CU by the tile's LDS).
    -> (splits per problem, workgroups, slots)"""
    slots = 256 * (4 if cls == 3 else 2 if cls == 0 else 3)
    kmax = [max(1, st // max(8, int(min_steps))) for st in steps]
    for K in range(max(kmax), 0, -1):
        sp = [min(K, km) for km in kmax]
        wg = sum(t * q for t, q in zip(tiles, sp))
        if wg <= slots or K == 1:
            return sp, wg, slots


def _weights_loaded(module, _incompatible_keys):
    """load_state_dict post hook of every module of a model that lives in flat buffers (each module of the recursion calls its own
    hooks, so a sub-module's load counts): the packed forward copy of whichever home its parameters live in is stale."""
    for q in module.parameters(recurse=False):
        h = getattr(q, "_ep24_home", None)
        if h is not None:
            h.mark_weights_changed()


def _same_bn(a, b):
    """Two BatchNorm modules that share one launch must agree on everything but their parameters and statistics."""
    if (a.eps, a.momentum, a.num_features) != (b.eps, b.momentum, b.num_features):
        raise _lib.Ep24Error("ep24: merged units need identical BatchNorm eps / momentum / width (%s vs %s); run them separately with "
                             "ep24.options.set_options(model, PlanOptions(merge_csp=False, merge_head=False))" % ((a.eps, a.momentum), (b.eps, b.momentum)))


def csp_is_merged(m, opts):
    """CSP layers without shortcut bottlenecks (the neck's four and dark5's) run conv1 and conv2 - two 1x1 convs over the
    same input - as ONE GEMM with one BatchNorm launch: their weights, BN parameters and running statistics sit next to each
    other in the flat buffers (order conv2, conv1: the layout of the concatenation the block builds)."""
    return opts.merge_csp and len(m.m) > 0 and (opts.merge_csp_shortcut or not any(b.use_add for b in m.m))


def _units(m):
    """The BaseConv units of a conv block in execution order: a DWConv is its depthwise unit, then its 1x1 unit."""
    return (m.dconv, m.pconv) if isinstance(m, enn.DWConv) else (m,)


def head_is_merged(head, opts):
    """The first 3x3 conv of the class and of the regression branch as one GEMM - dense heads only (a depthwise first conv is a
    per-channel pass over the SAME input twice: nothing to merge)."""
    return opts.merge_head and not getattr(head, "depthwise", False)


def exec_order(model, opts=None):
    opts = opts if opts is not None else get_options(model)
    """Modules in the order the plan executes them (yolox.py:24-34 -> yolo_pafpn.py:83-124 -> yolo_head_24p.py:150-189);
    any other container (tests build single blocks) falls back to registration order."""
    def head_order(head):
        for k in range(len(head.stems)):
            yield head.stems[k]
            if head_is_merged(head, opts):            # the first 3x3 conv of the class and of the regression branch read the same tensor
                yield ("pair_merged", head.cls_convs[k][0], head.reg_convs[k][0])
                yield head.cls_convs[k][1]
                yield head.reg_convs[k][1]
            else:
                for blk in list(head.cls_convs[k]) + list(head.reg_convs[k]):
                    yield from _units(blk)
        yield head

    if isinstance(model, enn.YOLOXHead):              # a head run on its own (YOLOXHead.forward)
        yield from head_order(model)
        return
    def swapped(bb):
        """(conv, bn) units / pre-activation pieces of a swapped backbone in execution order (darknet.py:179-674)."""
        if isinstance(bb, enn.ResNet):
            first = True
            for conv, bn in bb.used_units():
                yield ("unit", conv, bn, first)
                first = False
        elif isinstance(bb, enn.VGG):
            first = True
            for stage in bb.stages():
                for m in stage:
                    if isinstance(m, enn.ConvBNReLU):
                        yield ("unit", m.conv, m.bn, first)
                        first = False
            yield ("unit", bb.conv_add.conv, bb.conv_add.bn, False)
        elif isinstance(bb, enn.DenseNet):
            yield ("unit", bb.stem[0].conv, bb.stem[0].bn, True)

            def block(blk):
                for lay in blk.denseblock:
                    for cb in lay.conv_block:
                        yield ("bn", cb.bn)
                        yield ("conv", cb.conv, False)

            def trans(t):
                yield ("bn", t.trans[0].bn)
                yield ("conv", t.trans[0].conv, False)
            yield from block(bb.D1)
            yield from trans(bb.T1)
            yield from block(bb.D2)
            yield ("unit", bb.baseconv1.conv, bb.baseconv1.bn, False)
            yield from trans(bb.T2)
            yield from block(bb.D3)
            yield ("unit", bb.baseconv2.conv, bb.baseconv2.bn, False)
            yield from trans(bb.T3)
            yield from block(bb.D4)

    def resnet_unused(bb):                                 # fc / baseconv1..3: parameters the reference never runs (darknet.py:311-330)
        used = {id(p) for conv, bn in bb.used_units() for p in (conv.weight, bn.weight, bn.bias)}
        for p in bb.parameters():
            if id(p) not in used:
                yield ("unused", p)

    if isinstance(model, (enn.ResNet, enn.VGG, enn.DenseNet)):     # a swapped backbone run on its own (its forward)
        yield from swapped(model)
        if isinstance(model, enn.ResNet):
            yield from resnet_unused(model)
        return
    if not isinstance(model, enn.YOLOX):
        skip = set()
        for m in model.modules():
            if m in skip:
                continue
            if isinstance(m, enn.CSPLayer) and csp_is_merged(m, opts):
                yield ("csp_merged", m)
                skip |= {m.conv1, m.conv2}
                continue
            if isinstance(m, enn.ConvBlock):              # tests build stand-alone stages of the swapped backbones
                yield ("bn", m.bn)
                yield ("conv", m.conv, False)
            elif isinstance(m, (enn.BaseConv_DN, enn.ConvBNReLU)):
                yield ("unit", m.conv, m.bn, m.conv.in_channels == 3)     # 3 input channels: an image stem (im2col rows x GEMM)
            elif isinstance(m, enn.ResBottleneck):
                if m.downsample is not None:
                    yield ("unit", m.downsample[0], m.downsample[1], False)
                for conv, bn in ((m.conv1, m.bn1), (m.conv2, m.bn2), (m.conv3, m.bn3)):
                    yield ("unit", conv, bn, False)
            else:
                yield m
        return

    def csp(m):
        if csp_is_merged(m, opts):
            yield ("csp_merged", m)
        else:
            yield m.conv1
            yield m.conv2
        for blk in m.m:
            yield blk.conv1
            yield from _units(blk.conv2)
        yield m.conv3

    neck, head = model.backbone, model.head
    bb = neck.backbone
    if isinstance(bb, (enn.ResNet, enn.VGG, enn.DenseNet)):
        yield from swapped(bb)
    else:
        yield bb.stem.conv
        for name in ("dark2", "dark3", "dark4"):
            seq = getattr(bb, name)
            yield from _units(seq[0])
            yield from csp(seq[1])
        yield from _units(bb.dark5[0])
        yield bb.dark5[1].conv1
        yield bb.dark5[1].conv2
        yield from csp(bb.dark5[2])
    yield neck.lateral_conv0
    yield from csp(neck.C3_p4)
    yield neck.reduce_conv1
    yield from csp(neck.C3_p3)
    yield from _units(neck.bu_conv2)
    yield from csp(neck.C3_n3)
    yield from _units(neck.bu_conv1)
    yield from csp(neck.C3_n4)
    yield from head_order(head)
    if isinstance(bb, enn.ResNet):
        yield from resnet_unused(bb)


def _drop_rate_of(layers):
    """The Dropout2d probability of a run of DenseLayers (darknet.py:569-577: ``nn.Dropout2d(self.drop_rate)``).  One buffer of keep
    factors is drawn per plan with one probability, so the layers that drop at all must agree; 0 when none does."""
    rates = sorted({float(l.drop_rate) for l in layers if float(l.drop_rate) > 0})
    if len(rates) > 1:
        raise _lib.Ep24Error("ep24: the DenseLayers of one plan must share their drop_rate (got %s)" % rates)
    return rates[0] if rates else 0.0


def param_home(model):
    """The flat parameter buffers a module lives in.  A module of a tree that has already been moved into flat buffers
    (a YOLOPAFPN inside a YOLOX that has run) shares its owner's home; a module on its own gets one of its own."""
    if getattr(model, "_ep24_home", None) is None:
        ps = list(model.parameters())
        owner = getattr(ps[0], "_ep24_home", None) if ps else None
        if owner is not None and all(getattr(q, "_ep24_home", None) is owner for q in ps):
            model._ep24_home = owner
        else:
            model._ep24_home = ParamHome(model)
    return model._ep24_home


# ------------------------------------------------------------------------------------------------ engine
class Engine:
    def __init__(self, model, batch, size, dtype=BF16):
        """``dtype=torch.float32`` is the PARITY MODE: the same plan (buffers, concat slots, residual aliasing, accumulate
        flags, flat parameters) on fp32 activations through the ``ep24_f32_*`` entry points (csrc/f32path.hip) - what the
        reference's fp32 training (train_24p.py:86-104) is compared with end to end; the product path is bf16."""
        _lib.require_gpu()
        if dtype not in (BF16, torch.float32):
            raise _lib.Ep24Error("ep24: activations are bfloat16 (product) or float32 (parity mode)")
        self.IH, self.IW = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
        if self.IH <= 0 or self.IW <= 0:
            raise IndexError("ep24: empty input size %s" % (size,))
        self.model, self.B, self.S = model, batch, self.IH     # S: the side of a square input (kept for callers that pass one)
        self.dtype, self.f32 = dtype, dtype == torch.float32
        self.home = param_home(model)
        self.dev = self.home.dev
        self.C = getattr(getattr(model, "head", None), "num_classes", 0)
        self.ncols = 27 + self.C
        self.fwd, self.bwd = [], []              # launch lists: (abi name, args)
        self.outputs = None
        self.unit_acts = {}                      # BaseConv module -> (input, raw conv output, activated output)
        self.fwd_eval = []                       # eval-mode forward (running-statistics BN, sigmoid head): SURVEY 8f N3
        self.bwd_writes = []                     # per backward launch: flat-gradient ranges it writes (for ep24.dp)
        self.bwd_gw, self.bwd_rd = [], []        # per backward launch: activation-gradient regions written / the one a BN reduce reads
        self._bwd_units = 0
        self._cur_tag = None
        self._force_side = False
        self._deferred = []
        self.bwd_tail_cut = None                 # index behind the reduce launch that precedes the last unit of backward
        self.bwd_join = None                     # index of the first backward entry that needs the parallel head levels joined
        self.bwd_par_end = None                  # entries [0, bwd_par_end) all run on the side lane
        # per-model plan options (ep24.options): no process-wide switches.  A submodule of a model that already lives in flat
        # buffers (SubEngine on model.backbone, a CSPLayer ...) has no options of its own: it inherits the home's, i.e. the root's
        self.options = model.__dict__.get("_ep24_options") or getattr(self.home, "options", None) or get_options(model)
        if self.options.layout() != self.home.options.layout():
            raise _lib.Ep24Error("ep24: this model's parameters are laid out for %r; the merge options cannot change afterwards" % (self.home.options,))
        self.parallel_head = self.options.parallel_head and not self.f32
        self._dz_elems = 0
        self._slab_floats, self._pending_reduce, self._keep = 0, [], []
        self.head_grads = []                     # per head level: (cells per image, stride, d_regobj, d_cls)
        self.skip_decode_bwd = False             # True: the loss wrote those rows itself, the head_decode_bwd entries are skipped
        self._wg_pending = {}                    # tile class -> weight-gradient problems waiting for their grouped launch
        self._wg_tail = False
        self._side = None
        self.use_side = not self.f32            # weight gradients on a second stream
        self.capture_side = self.options.capture_side   # also inside captured graphs (experimental)
        self._events = []
        self._bwd_builders = []
        self.pre_bn_inputs = {}                  # BatchNorm module -> the Act its pre-activation form reads (tests)
        # eval mode: BatchNorm folded into the conv (ep24_fold_bn + ep24_conv_fwd_infer_bf16): one launch per unit instead of two
        self.fold_bn_eval = self.options.fold_bn_eval
        self._fold_units, self._fold_w, self._fold_c = [], 0, 0
        self.dyn = {"origin": None, "d_origin": None}   # run-time pointers (incoming gradient, L1-branch buffers)
        self.origin = None                       # [B,A,26] raw regression outputs, filled while use_l1 is on
        self._stats_specs, self._sum_specs = [], []
        self.max_dz = 0
        self._build()

    # ---- small helpers ----------------------------------------------------------------------------
    def new_act(self, C, H, W, ld=None):
        return Act(Buf(self.dev, self.B * H * W, ld or C, self.dtype), 0, C, self.B, H, W)

    def _kopt(self, name, args):
        """conv_fwd_bf16 / conv_dgrad_bf16 with the plan's kernel options (PlanOptions.conv_kernel_opts != 0: the _ex entry points);
        bit 8 of the options is the weight gradient's bit 0 (its loader / consumer ring form, an A/B option)."""
        ko = self.options.conv_kernel_opts
        if ko & 0xFF and name in ("conv_fwd_bf16", "conv_dgrad_bf16"):
            return name + "_ex", tuple(args) + (ko & 0xFF,)
        if ko & 0x100 and name in ("conv_wgrad_slab_bf16", "side:conv_wgrad_slab_bf16"):
            return name + "_ex", tuple(args) + (1,)
        return name, tuple(args)

    def _wsplits(self, B, H, W, cin, cout, k, s):
        """Pixel splits of a weight-gradient launch (the slab it needs), for the kernel the plan's options select."""
        return _lib.lib().fn["ep24_conv_wgrad_splits_ex"](B, H, W, cin, cout, k, s, 1 if self.options.conv_kernel_opts & 0x100 else 0)

    def _f(self, name, *args, ev=None):
        """Append a forward launch; `ev` = the (name, args) that replaces it in the eval-mode list (default: the same)."""
        name, args = self._kopt(name, args)
        self.fwd.append((name, args))
        if ev is not False:                          # ev=False: a training-only launch (batch statistics, dropout)
            self.fwd_eval.append(ev if ev is not None else (name, args))

    def _b(self, name, args, writes=(), reads=None):
        name, args = self._kopt(name, args)
        if self._force_side and name[0] != "@" and not name.startswith("side:"):
            name = "side:" + name
        self.bwd.append((name, args))
        self.bwd_writes.append([(seg.off, seg.numel) for seg in writes])
        self.bwd_gw.append(list(_PENDING_GW))        # activation-gradient regions this entry writes
        del _PENDING_GW[:]
        self.bwd_rd.append(reads.gregion() if reads is not None else None)

    # ---- graph construction -------------------------------------------------------------------------
    def _build(self):
        """The whole network: CSPDarknet (or a swapped backbone) -> PAFPN -> the three head levels
        (yolox.py:24-34, yolo_pafpn.py:83-124, yolo_head_24p.py:143-210)."""
        m, B = self.model, self.B
        bb, neck, head = m.backbone.backbone, m.backbone, m.head
        if self.IH % 32 or self.IW % 32:
            raise IndexError("ep24: the input height and width must be multiples of 32 (three stride levels), got %dx%d" % (self.IH, self.IW))
        self.images = torch.zeros(B, 3, self.IH, self.IW, dtype=torch.float32, device=self.dev)
        x2, x1, x0 = self.build_neck_inputs(neck)
        pans = self.build_neck(neck, x2, x1, x0)
        self.build_head(head, pans)
        self._finalize()

    def neck_channels(self, bb):
        swapped = isinstance(bb, (enn.ResNet, enn.DenseNet, enn.VGG))
        if swapped and self.f32:
            raise NotImplementedError("ep24: the fp32 parity mode covers the CSPDarknet network (the BASELINE configuration)")
        return (256, 512, 1024) if swapped else tuple(_units(st[0])[-1].conv.out_channels for st in (bb.dark3, bb.dark4, bb.dark5))

    def focus_stem(self, focus):
        """Focus + its 3x3 BaseConv (network_blocks.py:188-210).  bf16: the space-to-depth image [B][H/2][W/2][16] and a conv that
        gathers the nine taps itself (no im2col buffer: it was 459 MB written and read twice per step at B = 20).  The fp32 parity
        mode keeps the im2col rows (K = 108 -> 112) x 1x1 GEMM form."""
        B = self.B
        if self.f32:
            rows = self.new_act(112, self.IH // 2, self.IW // 2)
            rows.needs_grad = False
            self._f("f32_stem_pack", ptr(self.images), rows.ptr(), 112, B, self.IH, self.IW)
            return self.unit(focus.conv, rows, stem=True)
        f16 = self.new_act(16, self.IH // 2, self.IW // 2)
        f16.needs_grad = False
        self._f("focus_pack", ptr(self.images), f16.ptr(), B, self.IH, self.IW)
        return self.unit(focus.conv, f16, stem="focus")

    def build_backbone(self, bb, out3=None, out4=None):
        """images -> (dark3, dark4, dark5) (darknet.py:165-177); ``out3`` / ``out4``: concat slots of the neck to write into."""
        B = self.B
        if isinstance(bb, (enn.ResNet, enn.DenseNet, enn.VGG)):      # BASELINE config 4 (yolox/models/yolo_pafpn.py:31-38)
            build = self.resnet if isinstance(bb, enn.ResNet) else self.vgg if isinstance(bb, enn.VGG) else self.densenet
            return build(bb, out3, out4)
        x = self.focus_stem(bb.stem)
        x = self.csp(bb.dark2[1], self.conv_block(bb.dark2[0], x))
        x2 = self.csp(bb.dark3[1], self.conv_block(bb.dark3[0], x), out=out3)
        x1 = self.csp(bb.dark4[1], self.conv_block(bb.dark4[0], x2), out=out4)
        x0 = self.csp(bb.dark5[2], self.spp(bb.dark5[1], self.conv_block(bb.dark5[0], x1)))
        return x2, x1, x0

    def build_neck_inputs(self, neck, feats=None):
        """Allocates the neck's concat buffers; the backbone (or, for a neck run on its own, nothing) fills dark3 / dark4 slots."""
        c3, c4, c5 = self.neck_channels(neck.backbone)
        H3, W3, H4, W4, H5, W5 = self.IH // 8, self.IW // 8, self.IH // 16, self.IW // 16, self.IH // 32, self.IW // 32
        # concat buffers of the neck; producers write straight into their slot (yolo_pafpn.py:100-124)
        self.cat_p4 = self.new_act(2 * c4, H4, W4)        # [up(fpn_out0) | dark4]
        self.cat_p3 = self.new_act(2 * c3, H3, W3)        # [up(fpn_out1) | dark3]
        self.cat_n3 = self.new_act(2 * c3, H4, W4)        # [bu_conv2(pan_out2) | fpn_out1]
        self.cat_n4 = self.new_act(2 * c4, H5, W5)        # [bu_conv1(pan_out1) | fpn_out0]
        self._nc = (c3, c4, c5)
        return self.build_backbone(neck.backbone, self.cat_p3.slice(c3, c3), self.cat_p4.slice(c4, c4))

    def build_neck(self, neck, x2, x1, x0):
        """(dark3, dark4, dark5) -> (pan_out2, pan_out1, pan_out0) (yolo_pafpn.py:91-124); x2 / x1 already sit in their concat slots."""
        c3, c4, c5 = self._nc
        cat_p4, cat_p3, cat_n3, cat_n4 = self.cat_p4, self.cat_p3, self.cat_n3, self.cat_n4
        fpn_out0 = self.unit(neck.lateral_conv0, x0, out=cat_n4.slice(c4, c4))
        self.up2(fpn_out0, cat_p4.slice(0, c4))
        f_out0 = self.csp(neck.C3_p4, cat_p4)
        fpn_out1 = self.unit(neck.reduce_conv1, f_out0, out=cat_n3.slice(c3, c3))
        self.up2(fpn_out1, cat_p3.slice(0, c3))
        pan_out2 = self.csp(neck.C3_p3, cat_p3)
        self.fwd_fork = len(self.fwd)                 # pan_out2 is complete: head level 0 can start (ep24.train runs it on a second lane)
        self.conv_block(neck.bu_conv2, pan_out2, out=cat_n3.slice(0, c3))
        pan_out1 = self.csp(neck.C3_n3, cat_n3)
        self.fwd_fork1 = len(self.fwd)                # pan_out1 is complete: head level 1 can start
        self.conv_block(neck.bu_conv1, pan_out1, out=cat_n4.slice(0, c4))
        pan_out0 = self.csp(neck.C3_n4, cat_n4)
        return pan_out2, pan_out1, pan_out0

    def build_head(self, head, feats):
        """Three feature maps -> outputs [B, A, 27 + C] fp32 and the anchor tables of the train-mode tuple (yolo_head_24p.py:143-237)."""
        B = self.B
        self.C = head.num_classes
        self.ncols = 27 + self.C
        self.A = sum(f.H * f.W for f in feats)
        self.outputs = torch.zeros(B, self.A, self.ncols, dtype=torch.float32, device=self.dev)
        a0 = 0
        self.levels = []
        self.fwd_head0 = None
        for k, feat in enumerate(feats):
            lo = len(self.fwd)
            self.head_level(head, k, feat, a0)
            if k == 0:
                self.fwd_head0 = (lo, len(self.fwd))
            if k == 1:
                self.fwd_head1 = (lo, len(self.fwd))
            self._cur_tag = None
            a0 += feat.H * feat.W
        # anchor tables of the train-mode tuple (yolo_head_24p.py:172-176)
        self.x_shifts, self.y_shifts, self.exp_strides = [], [], []
        for (H, W, s) in self.levels:
            yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            self.x_shifts.append(xv.reshape(1, -1).float().to(self.dev))
            self.y_shifts.append(yv.reshape(1, -1).float().to(self.dev))
            self.exp_strides.append(torch.full((1, H * W), float(s), device=self.dev))

    def _finalize(self):
        """Allocate the shared scratch, then build the backward list in reverse op order and resolve pointers."""
        # scratch shared by all layers (single stream => no overlap in time)
        self.stats = torch.zeros(max(sum(self._stats_specs), 4), dtype=torch.int64, device=self.dev)
        self.bnsums = torch.zeros(max(sum(self._sum_specs), 4), dtype=torch.int64, device=self.dev)
        # The backward of head levels 1 and 2 (40x40 and 20x20: kernels that leave most of the chip idle) depends on
        # nothing but the loss gradient, and nothing needs its results before the PAFPN backward: all of it goes to the
        # weight-gradient lane, where it runs next to the level-0 chain of the main lane; the main lane joins before
        # the first trunk entry.
        in_head = False
        n_builders = len(self._bwd_builders)
        for bi, b in enumerate(reversed(self._bwd_builders)):
            if not self._wg_tail and bi >= WGRAD_GROUP_TAIL * n_builders:
                self._wg_tail = True                      # from here on: one launch per layer, nothing left waiting at the end
                self._flush_wgrad()
            tag = getattr(b, "tag", None)
            is_head = tag is not None and tag[0] == "head"
            if self.parallel_head:
                if in_head and not is_head:
                    self._b("@main_wait_side", ("all",))
                    self.bwd_join = len(self.bwd)
                want = is_head and tag[1] >= 1
                if want and not self._force_side:
                    self._b("@side_wait_main", ())
                if self._force_side and not want:
                    self.bwd_par_end = len(self.bwd)
                    self._force_side = False
                    for f in self._deferred:              # the weight gradients of the side-lane chains: after the chains
                        f()
                    self._deferred = []
                self._force_side = want
            in_head = is_head
            if b is self._bwd_builders[0] and len(self._bwd_builders) > 1:
                # the last unit of backward (the stem): fold the pending weight-gradient slabs now, so that the final reduce launch
                # covers this unit only and everything else is complete before it (ep24.train updates those parameters meanwhile)
                self._flush_reduce()
                self.bwd_tail_cut = len(self.bwd)
            b()
        self._force_side = False
        for f in self._deferred:
            f()
        self._deferred = []
        self._flush_reduce()
        del _PENDING_GW[:]
        self.slab = torch.zeros(max(self._slab_floats, 4), dtype=torch.float32, device=self.dev)
        self.fold_w = torch.zeros(max(self._fold_w, 8), dtype=BF16, device=self.dev)
        self.fold_b = torch.zeros(max(self._fold_c, 4), dtype=torch.float32, device=self.dev)
        if self._fold_units:
            b0 = self.home.bflat.data_ptr()
            rows, pref, cpref, eps = [], [0], [0], []
            for seg, gam, bet, bn, woff, coff in self._fold_units:
                rows.append([seg.off, woff, seg.cout, seg.taps, seg.cin, seg.cin_pad, gam.off, bet.off,
                             (bn.running_mean.data_ptr() - b0) // 4, (bn.running_var.data_ptr() - b0) // 4, coff, 0])
                pref.append(pref[-1] + seg.numel)
                cpref.append(cpref[-1] + seg.cout)
                eps.append(float(bn.eps))
            t64 = dict(dtype=torch.int64, device=self.dev)
            self._fold_desc = (torch.tensor(rows, **t64), torch.tensor(pref, **t64), torch.tensor(cpref, **t64),
                               torch.tensor(eps, dtype=torch.float32, device=self.dev), len(rows), pref[-1], cpref[-1])
        # every layer keeps its own dz (the gradient w.r.t. the raw conv output): the weight-gradient lane may lag the
        # main lane by a whole segment without a write-after-read hazard (3.4 GB at -l / B=20; there are 288)
        self.dzbuf = torch.zeros(max(self._dz_elems, 8), dtype=self.dtype, device=self.dev)
        fw, bw, fe = [], [], []
        for lst, out in ((self.fwd, fw), (self.bwd, bw), (self.fwd_eval, fe)):
            for name, args in lst:
                out.append((name, tuple(a() if callable(a) else a for a in args)))   # Dyn stays for run time
        self.fwd, self.bwd, self.fwd_eval = fw, bw, fe

    def _add_builder(self, fn):
        fn.tag = self._cur_tag                    # which part of the network registered it (head level k / trunk)
        self._bwd_builders.append(fn)

    # ---- grouped weight gradients (round 5) ---------------------------------------------------------
    def _wg_plan(self, probs, cls):
        return wgrad_group_splits([pr["tiles"] for pr in probs], [pr["steps"] for pr in probs], cls, self.options.wgrad_group_steps)

    def _pend_wgrad(self, pr):
        fn = _lib.lib().fn
        cls = fn["ep24_conv_wgrad_tile_class"](pr["cin"], pr["cout"], pr["k"])
        tco, tci = (64 if cls & 1 else 128), (64 if cls & 2 else 128)
        pr["tiles"] = -(-pr["cin"] // tci) * -(-pr["cout"] // tco) * pr["k"] * pr["k"]
        OH, OW = (pr["H"] - 1) // pr["s"] + 1, (pr["W"] - 1) // pr["s"] + 1
        pr["steps"] = -(-(pr["B"] * OH * OW) // 64)
        lst = self._wg_pending.setdefault(cls, [])
        if lst and self._wg_plan(lst + [pr], cls)[1] > self._wg_plan(lst + [pr], cls)[2]:
            # with this problem the group would not fit one round even unsplit: a second round of a few LONG workgroups costs as much
            # as the first (the VGG backbone lost 12 % that way: four 512-channel layers as 576 workgroups of 500 steps on 512 slots,
            # profiles/r05_configs.txt) - what waits goes out now, the new problem starts the next group
            self._flush_wgrad(cls)
            lst = self._wg_pending.setdefault(cls, [])
        lst.append(pr)
        _sp, wg, slots = self._wg_plan(lst, cls)
        if len(lst) >= WGRAD_GROUP_MAX or wg >= WGRAD_GROUP_FILL * slots:
            self._flush_wgrad(cls)

    def _flush_wgrad(self, only=None):
        """Emit the grouped launch of every (or one) tile class with waiting problems, in the order the classes were first used."""
        for cls in ([only] if only is not None else list(self._wg_pending)):
            probs = self._wg_pending.pop(cls, None)
            if not probs:
                continue
            splits, _wg, _slots = self._wg_plan(probs, cls)
            rows = []
            for pr, sp in zip(probs, splits):
                seg = pr["seg"]
                soff = self._slab_floats
                self._slab_floats += sp * seg.numel
                rows.append((pr, sp, soff))
                self._pending_reduce.append((seg, sp, soff))
            keep = self._keep

            def table(rows=rows):                      # resolved with the other launch arguments in _finalize: a HOST int64 array
                t = torch.tensor([[pr["x"](), pr["ld_x"], pr["dz"](), pr["ld_dy"], self.slab.data_ptr() + 4 * soff, sp * pr["seg"].numel,
                                   pr["ld_dw"], pr["cout_valid"], pr["cin_valid"], pr["B"], pr["H"], pr["W"], pr["cin"], pr["cout"], pr["k"], pr["s"], sp]
                                  for pr, sp, soff in rows], dtype=torch.int64)
                keep.append(t)
                return t.data_ptr()
            self._b("@side_wait_main", ())
            self._b("side:conv_wgrad_group_bf16", (table, len(rows)))
            self._b("@side_record", (probs[-1]["idx"],))

    def _flush_reduce(self):
        """One reduce launch (side stream, behind the weight-gradient kernels it sums) for the pending layers."""
        self._flush_wgrad()                               # the slabs it folds must have been written
        if not self._pending_reduce:
            return
        rows = [[seg.off, seg.numel, splits, soff] for seg, splits, soff in self._pending_reduce]
        desc = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        self._keep.append(desc)
        self._b("side:wgrad_reduce", (ptr(desc), len(rows), max(r[1] for r in rows), ptr(self.home.gflat),
                                     (lambda: self.slab.data_ptr())), writes=tuple(seg for seg, _, _ in self._pending_reduce))
        self._pending_reduce = []

    def _stats_slot(self, C):
        off = sum(self._stats_specs)
        self._stats_specs.append(STATS_REPLICAS * 2 * C)
        return lambda: self.stats.data_ptr() + 8 * off

    def _bar_slot(self):
        """One zeroed word per unit for the grid-wide wait of ep24_bn_act_bwd_fused (it lives in the sums buffer: cleared with it)."""
        off = sum(self._sum_specs)
        self._sum_specs.append(2)
        return lambda: self.bnsums.data_ptr() + 8 * off

    def _sums_slot(self, C):
        """[STATS_REPLICAS][2][C] fixed-point sums of one BatchNorm backward (replica 0's two pointers; replica r is 2 C r further)."""
        off = sum(self._sum_specs)
        self._sum_specs.append(STATS_REPLICAS * 2 * C)
        return (lambda: self.bnsums.data_ptr() + 8 * off), (lambda: self.bnsums.data_ptr() + 8 * (off + C))

    # ---- ops ---------------------------------------------------------------------------------------
    def conv_block(self, mod, x, out=None, residual=None, x_single=False):
        """`Conv = DWConv if depthwise else BaseConv` of the reference's constructors: a BaseConv unit, or a DWConv = depthwise unit
        followed by a 1x1 unit (network_blocks.py:57-76), which takes the block's output slot and shortcut."""
        if isinstance(mod, enn.DWConv):
            return self.unit(mod.pconv, self.unit_dw(mod.dconv, x), out=out, residual=residual)
        return self.unit(mod, x, out=out, residual=residual, x_single=x_single)

    def unit_dw(self, mod, x):
        """A depthwise BaseConv (groups = channels): conv -> BN(batch stats) -> SiLU with the conv as HBM-bound elementwise kernels
        (csrc/dwconv.hip) over the fp32 master weights; BatchNorm + activation are the launches of every other unit."""
        if self.f32:
            raise NotImplementedError("ep24: the fp32 parity mode covers the dense network (the BASELINE configurations); depthwise variants run in bf16")
        home = self.home
        conv, bn, act = mod.conv, mod.bn, mod.act_code
        seg = home.by_param[conv.weight]
        gam, bet = home.by_param[bn.weight], home.by_param[bn.bias]
        k, s, C = conv.kernel_size[0], conv.stride[0], x.C
        assert (seg.cout, seg.cin, seg.taps) == (C, 1, 9), (seg.cout, seg.cin, seg.taps, C)
        B, H, W = x.B, x.H, x.W
        OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
        out, z = self.new_act(C, OH, OW), self.new_act(C, OH, OW)
        z.needs_grad = False
        M = B * OH * OW
        self.max_dz = max(self.max_dz, M * C)
        save = torch.zeros(2 * C, dtype=torch.float32, device=self.dev)
        stats = self._stats_slot(C)
        sum_g, sum_b = self._sums_slot(C)
        flat, gflat = home.flat, home.gflat
        w_p = ptr(flat, seg.off)
        # eval mode: the same conv without statistics, then BatchNorm from the running statistics (not folded: the weights are the masters)
        self._f("dwconv_fwd_bf16", x.ptr(), x.ld, w_p, z.ptr(), z.ld, stats, STATS_REPLICAS, B, H, W, C, k, s,
                ev=("dwconv_fwd_bf16", (x.ptr(), x.ld, w_p, z.ptr(), z.ld, None, 1, B, H, W, C, k, s)))
        self._f("bn_act_fwd", z.ptr(), z.ld, stats, STATS_REPLICAS, ptr(flat, gam.off), ptr(flat, bet.off), ptr(bn.running_mean),
                ptr(bn.running_var), ptr(bn.num_batches_tracked), None, ptr(save), out.ptr(), out.ld, None, 0, M, C, float(bn.eps),
                float(bn.momentum), act,
                ev=("bn_act_infer", (z.ptr(), z.ld, ptr(flat, gam.off), ptr(flat, bet.off), ptr(bn.running_mean), ptr(bn.running_var),
                                     out.ptr(), out.ld, None, 0, M, C, float(bn.eps), act)))
        self.unit_acts[mod] = (x, z, out)

        def build_bwd():
            assert out.gready(), "activation without a gradient producer"
            kk = self._bwd_units
            self._bwd_units += 1
            dzoff = self._dz_elems
            self._dz_elems += M * C
            dz = (lambda dzoff=dzoff: self.dzbuf.data_ptr() + 2 * dzoff)
            self._b("bn_act_bwd_reduce", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off), sum_g, sum_b,
                                          M, C, act, STATS_REPLICAS), reads=out)
            self._b("bn_act_bwd_apply", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off), sum_g, sum_b,
                                         ptr(gflat, gam.off), ptr(gflat, bet.off), dz, C, M, C, act, STATS_REPLICAS), writes=(gam, bet))
            splits = _lib.lib().fn["ep24_dwconv_wgrad_splits"](B, H, W, C, s)
            assert splits >= 1, splits

            def emit_wgrad():                         # side lane: per-workgroup partial sums, folded by the next reduce launch
                soff = self._slab_floats
                self._slab_floats += splits * seg.numel
                self._b("@side_wait_main", ())
                self._b("side:dwconv_wgrad_slab_bf16", (x.ptr(), x.ld, dz, C, (lambda soff=soff: self.slab.data_ptr() + 4 * soff),
                                                         splits * seg.numel, B, H, W, C, k, s))
                self._b("@side_record", (kk,))
                self._pending_reduce.append((seg, splits, soff))
                if len(self._pending_reduce) >= WGRAD_REDUCE_GROUP:
                    self._flush_reduce()

            if self._force_side:
                self._deferred.append(emit_wgrad)
            else:
                emit_wgrad()
            if x.needs_grad:
                acc = x.gwrite()
                self._b("dwconv_dgrad_bf16", (dz, C, w_p, x.gptr(), x.gld, acc, B, H, W, C, k, s))

        self._add_builder(build_bwd)
        return out

    def unit(self, mod, x, out=None, residual=None, stem=False, conv=None, bn=None, act=None, bn2=None, x_single=False, bn_in=False, below_in=False):
        """BaseConv: conv -> BN(batch stats) -> SiLU (+ residual) (network_blocks.py:50-51).  ``conv`` / ``bn`` / ``act``
        name the pieces of a unit that is not a BaseConv (the ResNet backbone: act 2 = ReLU, 0 = none).  ``x_single``: the
        caller states that this unit is the only consumer of ``x`` (a Bottleneck's 3x3 over its 1x1): its input gradient IS the
        dy of the unit that produced x, and a 3x3 stride-1 one then takes that unit's BatchNorm-backward sums in its epilogue
        (ep24_conv_dgrad_bnr_bf16) instead of a reduce launch re-reading dy and z."""
        home = self.home
        if act is None:                            # a BaseConv carries its activation (silu / relu / lrelu); merged pairs: the first module's
            act = getattr(mod, "act_code", 1)
        conv = mod.conv if conv is None else conv
        bn = mod.bn if bn is None else bn
        seg = home.by_param[conv.weight]
        gam, bet = home.by_param[bn.weight], home.by_param[bn.bias]
        focus = stem == "focus"                   # the Focus stem over the space-to-depth image (16 channels, 12 real): its own kernels
        k = k_ = 1 if stem else conv.kernel_size[0]
        s = 1 if stem else conv.stride[0]
        cin = x.C                                  # stem: im2col width 112 (108 real columns)
        assert (x.C == 16 and (x.ld, seg.cin) == (16, 108)) if focus else x.C == seg.cin_pad if stem else x.C == seg.cin, (x.C, seg.cin)
        cout = seg.cout
        B, H, W = x.B, x.H, x.W
        OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
        if out is None:
            out = self.new_act(cout, OH, OW)
        assert (out.H, out.W, out.C) == (OH, OW, cout)
        z = self.new_act(cout, OH, OW)
        z.needs_grad = False
        M = B * OH * OW
        self.max_dz = max(self.max_dz, M * cout)
        save = torch.zeros(2 * cout, dtype=torch.float32, device=self.dev)
        stats = self._stats_slot(cout)
        sum_g, sum_b = self._sums_slot(cout)
        flat, gflat = home.flat, home.gflat
        wf = ptr(home.wf, seg.wf_off)              # stem: master row [108] zero padded to the im2col width
        res_p = residual.ptr() if residual is not None else None
        res_ld = residual.ld if residual is not None else 0
        if self.f32:
            if bn2 is not None:
                self._f("incr_i64", ptr(bn2.num_batches_tracked), ev=False)
            return self._unit_f32(mod, conv, bn, seg, gam, bet, x, z, out, residual, res_p, res_ld, k_, s, act, save, sum_g)
        if self.fold_bn_eval:
            woff, coff = self._fold_w, self._fold_c
            self._fold_w += cout * seg.taps * seg.cin_pad
            self._fold_c += cout
            self._fold_units.append((seg, gam, bet, bn, woff, coff))
            fw, fb = (lambda woff=woff: self.fold_w.data_ptr() + 2 * woff), (lambda coff=coff: self.fold_b.data_ptr() + 4 * coff)
            if focus:
                assert residual is None
                ev_conv = ("stem_conv_fwd_infer_bf16", (x.ptr(), fw, seg.cin_pad, fb, act, out.ptr(), out.ld, B, H, W, cout))
            else:
                ev_conv = ("conv_fwd_infer_bf16", (x.ptr(), x.ld, fw, fb, act, res_p, res_ld, out.ptr(), out.ld, B, H, W, cin, cout, k, s))
        elif focus:
            ev_conv = ("stem_conv_fwd_bf16", (x.ptr(), wf, seg.cin_pad, z.ptr(), z.ld, None, 1, B, H, W, cout))
        else:
            ev_conv = ("conv_fwd_bf16", (x.ptr(), x.ld, wf, z.ptr(), z.ld, 0, 0, 0, None, None, 1, B, H, W, cin, cout, k, s))
        xf_ok = (not stem and k == 1 and s == 1 and act == 1 and cout % 8 == 0 and
                 _lib.lib().fn["ep24_conv1x1_xf_ok"](B, H, W, cin, cout) == 1)       # a shape of the transformed-A streaming kernel
        prev = getattr(x, "_bnf", None) if bn_in and self.options.fuse_bn_stream and xf_ok else None
        if prev is not None and prev[0] == len(self.fwd) - 1 and self.fwd[-1][0] == "bn_act_fwd" and prev[1][-1] == 1:
            # The BatchNorm pass that produced x is the launch just before this one and this unit streams x through registers: it makes
            # x itself (same expressions, same stored bytes) - the pass leaves the training list; the eval list keeps its own entries.
            (pz, pldz, pstats, preps, pgam, pbet, prm, prv, pnbt, pnbt2, psave, py, pldy, pres, pldres, pM, pC, peps, pmom, _pact) = prev[1]
            assert (py, pldy, pM, pC) == (x.ptr(), x.ld, M, cin)
            self.fwd.pop()
            self.fwd.append(("conv1x1_bnin_bf16", (pz, pldz, pstats, preps, pgam, pbet, prm, prv, pnbt, pnbt2, psave, py, pldy, pres, pldres,
                                                   peps, pmom, 1, wf, z.ptr(), z.ld, stats, STATS_REPLICAS, B, H, W, cin, cout)))
            self.fwd_eval.append(ev_conv)
            self.n_bnin = getattr(self, "n_bnin", 0) + 1
        elif focus:
            self._f("stem_conv_fwd_bf16", x.ptr(), wf, seg.cin_pad, z.ptr(), z.ld, stats, STATS_REPLICAS, B, H, W, cout, ev=ev_conv)
        else:
            self._f("conv_fwd_bf16", x.ptr(), x.ld, wf, z.ptr(), z.ld, 0, 0, 0, None, stats, STATS_REPLICAS, B, H, W, cin, cout, k, s, ev=ev_conv)
        bn_args = (z.ptr(), z.ld, stats, STATS_REPLICAS, ptr(flat, gam.off), ptr(flat, bet.off),
                   ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.num_batches_tracked),
                   ptr(bn2.num_batches_tracked) if bn2 is not None else None, ptr(save), out.ptr(), out.ld,
                   res_p, res_ld, M, cout, float(bn.eps), float(bn.momentum), act)
        self._f("bn_act_fwd", *bn_args,
                ev=False if self.fold_bn_eval else
                ("bn_act_infer", (z.ptr(), z.ld, ptr(flat, gam.off), ptr(flat, bet.off), ptr(bn.running_mean),
                                  ptr(bn.running_var), out.ptr(), out.ld, res_p, res_ld, M, cout, float(bn.eps), act)))
        out._bnf = (len(self.fwd) - 1, bn_args)        # where this output's BatchNorm pass sits (a consumer with bn_in may absorb it)
        if residual is not None:
            residual.alias_grad(out)
        self.unit_acts[mod if mod is not None else conv] = (x, z, out)
        info = dict(z=z, save=save, gam=gam, bet=bet, sum_g=sum_g, sum_b=sum_b, act=act, cout=cout, fused=0)
        # one launch for both BatchNorm-backward passes (PlanOptions.fuse_bn_bwd): trunk units only - while the head levels' backward
        # runs on the second lane its ring kernels hold whole CUs' LDS, and a workgroup that cannot be placed keeps the others waiting
        one_launch = bool(self.options.fuse_bn_bwd) and not (self._cur_tag is not None and self._cur_tag[0] == "head") and cout <= 2048
        bar = self._bar_slot() if one_launch else None
        if residual is None:                       # with a residual the incoming gradient is shared with the shortcut: not this unit's alone
            out.bn_info, out.bn_c0 = info, 0
        out.bn_any = info                          # ... which a consumer that states `below_in` knows how to complete

        def build_bwd():
            assert out.gready(), "activation without a gradient producer"
            k = self._bwd_units                       # position in backward execution order
            self._bwd_units += 1
            dgrad_done = False
            dzoff = self._dz_elems
            self._dz_elems += M * cout
            dz = (lambda dzoff=dzoff: self.dzbuf.data_ptr() + 2 * dzoff)
            assert info["fused"] in (0, cout), "BatchNorm-backward sums fused for a part of the channels only"
            if one_launch and not info["fused"]:
                self._b("bn_act_bwd_fused", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off), sum_g, sum_b,
                                             ptr(gflat, gam.off), ptr(gflat, bet.off), dz, cout, M, cout, act, STATS_REPLICAS, bar),
                        writes=(gam, bet), reads=out)
            else:
                if not info["fused"]:                 # else: the consumer's input-gradient epilogue has produced the two sums
                    self._b("bn_act_bwd_reduce", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off),
                                                  ptr(flat, bet.off), sum_g, sum_b, M, cout, act, STATS_REPLICAS), reads=out)
                # The apply pass and the input gradient of a 1x1 unit of the streaming kernel as ONE launch (PlanOptions.fuse_bn_dgrad,
                # ep24_conv1x1_dgrad_bnbwd_bf16): dz is made on the way to the MFMAs and stored for the weight gradient.
                if xf_ok and self.options.fuse_bn_dgrad and x.needs_grad and not info["fused"] and seg.cout_pad == cout:
                    acc = x.gwrite()
                    self._b("conv1x1_dgrad_bnbwd_bf16", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off),
                                                          sum_g, sum_b, ptr(gflat, gam.off), ptr(gflat, bet.off), dz, cout, act, STATS_REPLICAS,
                                                          ptr(home.wd, seg.wd_off), x.gptr(), x.gld, acc, B, H, W, cin, cout), writes=(gam, bet))
                    self.n_bnbwd = getattr(self, "n_bnbwd", 0) + 1
                    dgrad_done = True
                else:
                    self._b("bn_act_bwd_apply", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off),
                                                 ptr(flat, bet.off), sum_g, sum_b, ptr(gflat, gam.off), ptr(gflat, bet.off),
                                                 dz, cout, M, cout, act, STATS_REPLICAS), writes=(gam, bet), reads=out if info["fused"] else None)
            # weight gradient on the side stream: it only needs dz and the saved input, and nothing on the main
            # stream needs its result before the optimizer, so it overlaps the dgrad and the next layer's BN passes
            # partial sums of the pixel splits go to this layer's slab slice with plain stores; a reduce launch every
            # few layers folds them into the flat gradient in a fixed order (no atomics: bitwise reproducible)
            splits = _lib.lib().fn["ep24_stem_conv_wgrad_splits"](B, H, W, cout) if focus else \
                self._wsplits(B, H, W, cin, cout, k_, s)
            assert splits >= 1, splits

            def emit_wgrad():
                if self.options.group_wgrad and not focus and not self._wg_tail and not (self.options.conv_kernel_opts & 0x100):
                    # grouped with the other layers of its tile class (csrc/conv_wgrad.hip wgrad_group_kernel); its slab is
                    # allotted when the group is launched (the split count is the group's)
                    self._pend_wgrad(dict(x=x.ptr, ld_x=x.ld, dz=dz, ld_dy=cout, ld_dw=seg.taps * seg.cin, cout_valid=cout, cin_valid=seg.cin,
                                          B=B, H=H, W=W, cin=cin, cout=cout, k=k_, s=s, seg=seg, idx=k))
                    if len(self._pending_reduce) >= WGRAD_REDUCE_GROUP:
                        self._flush_reduce()
                    return
                soff = self._slab_floats
                self._slab_floats += splits * seg.numel
                self._b("@side_wait_main", ())
                slab_p = (lambda soff=soff: self.slab.data_ptr() + 4 * soff)
                if focus:
                    self._b("side:stem_conv_wgrad_slab_bf16", (x.ptr(), dz, cout, slab_p, splits * seg.numel, B, H, W, cout))
                else:
                    self._b("side:conv_wgrad_slab_bf16", (x.ptr(), x.ld, dz, cout, slab_p,
                                                          splits * seg.numel, seg.taps * seg.cin, cout, seg.cin, B, H, W, cin, cout, k_, s))
                self._b("@side_record", (k,))
                self._pending_reduce.append((seg, splits, soff))
                if len(self._pending_reduce) >= WGRAD_REDUCE_GROUP:
                    self._flush_reduce()

            # inside the head levels that run on the side lane the weight gradients wait until the chain is through, so
            # that the main lane's join is not held up by work nothing depends on (every layer owns its dz)
            if self._force_side:
                self._deferred.append(emit_wgrad)
            else:
                emit_wgrad()
            if x.needs_grad and not dgrad_done:
                acc = x.gwrite()
                below = getattr(x, "bn_info", None) if x_single else None
                if (below is not None and not acc and k_ == 3 and s == 1 and self.options.fuse_bn_reduce and x._alias is None and
                        x.C % 8 == 0 and x.gld % 8 == 0 and below["z"].ld % 8 == 0):
                    o, ct = x.bn_c0, below["cout"]
                    zb, sv, gb, bb = below["z"], below["save"], below["gam"], below["bet"]
                    self._b("conv_dgrad_bnr_bf16", (dz, cout, ptr(home.wd, seg.wd_off), x.gptr(), x.gld, B, H, W, cin, seg.cout_pad, k_,
                                                    zb.ptr() + 2 * o, zb.ld, ptr(sv, o), ptr(sv, ct + o), ptr(flat, gb.off + o), ptr(flat, bb.off + o),
                                                    (lambda f=below["sum_g"], o=o: f() + 8 * o), (lambda f=below["sum_b"], o=o: f() + 8 * o),
                                                    2 * ct, STATS_REPLICAS, below["act"]))
                    below["fused"] += x.C
                elif (below_in and self.options.fuse_bn_reduce_stream and k_ == 1 and s == 1 and getattr(x, "bn_any", None) is not None and
                      not x.bn_any["fused"] and x.bn_any["act"] == 1 and x.bn_any["cout"] == cin and x.bn_any["z"].ld % 4 == 0 and
                      _lib.lib().fn["ep24_conv_kernel_for"](1, B, H, W, cin, seg.cout_pad, 1, 1, 0, 0) == 2):
                    # d(x) is complete with this launch and is the dy of the unit below: its reduce pass rides on the rows being stored
                    bl = x.bn_any
                    zb, sv, ct = bl["z"], bl["save"], bl["cout"]
                    self._b("conv1x1_dgrad_bnr_bf16", (dz, cout, ptr(home.wd, seg.wd_off), x.gptr(), x.gld, acc, B, H, W, cin, seg.cout_pad,
                                                        zb.ptr(), zb.ld, ptr(sv, 0), ptr(sv, ct), ptr(flat, bl["gam"].off), ptr(flat, bl["bet"].off),
                                                        bl["sum_g"], bl["sum_b"], 2 * ct, STATS_REPLICAS, 1))
                    bl["fused"] += cin
                    self.n_bnr_stream = getattr(self, "n_bnr_stream", 0) + 1
                else:
                    self._b("conv_dgrad_bf16", (dz, cout, ptr(home.wd, seg.wd_off), x.gptr(), x.gld, acc, B, H, W, cin,
                                                seg.cout_pad, k_, s))

        self._add_builder(build_bwd)
        return out


    def _unit_f32(self, mod, conv, bn, seg, gam, bet, x, z, out, residual, res_p, res_ld, k, s, act, save, sums):
        """The unit in the fp32 parity mode: same buffers and gradient bookkeeping, weights read in place from the fp32
        master ([Cout][T][Cin] = element (co, t, ci) at co*T*Cin + t*Cin + ci), everything on one lane."""
        home = self.home
        flat, gflat = home.flat, home.gflat
        B, H, W = x.B, x.H, x.W
        cin, cout, T = seg.cin, seg.cout, seg.taps     # stem: cin = 108 real columns of the 112-wide rows, one tap
        M = out.M
        wco, wt = T * cin, cin
        w_p, gw_p = ptr(flat, seg.off), ptr(gflat, seg.off)
        self._f("f32_conv", x.ptr(), x.ld, w_p, wco, wt, z.ptr(), z.ld, 0, 0, None, 0, B, H, W, cin, cout, k, s, 0, ev=False)
        self._f("f32_bn_act_fwd", z.ptr(), z.ld, ptr(flat, gam.off), ptr(flat, bet.off), ptr(bn.running_mean), ptr(bn.running_var),
                ptr(bn.num_batches_tracked), ptr(save), out.ptr(), out.ld, res_p, res_ld, M, cout, float(bn.eps), float(bn.momentum),
                act, ev=False)
        if residual is not None:
            residual.alias_grad(out)
        self.unit_acts[mod if mod is not None else conv] = (x, z, out)

        def build_bwd():
            assert out.gready(), "activation without a gradient producer"
            self._bwd_units += 1
            dzoff = self._dz_elems
            self._dz_elems += M * cout
            dz = (lambda dzoff=dzoff: self.dzbuf.data_ptr() + 4 * dzoff)
            self._b("f32_bn_act_bwd", (out.gptr(), out.gld, z.ptr(), z.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off), sums,
                                       ptr(gflat, gam.off), ptr(gflat, bet.off), dz, cout, M, cout, act), writes=(gam, bet), reads=out)
            self._b("f32_conv_wgrad", (x.ptr(), x.ld, dz, cout, gw_p, wco, wt, B, H, W, cin, cout, k, s), writes=(seg,))
            if x.needs_grad:
                acc = x.gwrite()
                self._b("f32_conv", (dz, cout, w_p, wco, wt, x.gptr(), x.gld, 0, 0, None, acc, B, H, W, cin, cout, k, s, 1))

        self._add_builder(build_bwd)
        return out

    def csp(self, mod, x, out=None):
        """CSPLayer: cat(m(conv1(x)), conv2(x)) -> conv3 (network_blocks.py:179-185)."""
        if csp_is_merged(mod, self.options):
            return self.csp_merged(mod, x, out)
        h = self.home.by_param[mod.conv1.conv.weight].cout
        cat = self.new_act(2 * h, x.H, x.W)
        n = len(mod.m)
        t = self.unit(mod.conv1, x, out=cat.slice(0, h) if n == 0 else None)
        self.unit(mod.conv2, x, out=cat.slice(h, h))
        for i, blk in enumerate(mod.m):
            last = i == n - 1
            u = self.unit(blk.conv1, t, bn_in=i > 0, below_in=i > 0)
            t = self.conv_block(blk.conv2, u, out=cat.slice(0, h) if last else None, residual=t if blk.use_add else None, x_single=True)
        return self.unit(mod.conv3, cat, out=out)

    def relu(self, y):
        """In-place ReLU on an activation that is a sum ("out += identity; out = relu(out)", darknet.py:266-268): the
        backward masks the incoming gradient in place with the stored output."""
        self._f("relu_fwd", y.ptr(), y.ld, y.M, y.C)

        def build_bwd():
            assert y.gready(), "activation without a gradient producer"
            _PENDING_GW.append(y.gregion())
            self._b("relu_bwd", (y.gptr(), y.gld, y.ptr(), y.ld, y.M, y.C))

        self._add_builder(build_bwd)
        return y

    def maxpool3s2(self, x, out=None):
        """nn.MaxPool2d(3, 2, 1) (darknet.py:303)."""
        OH, OW = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        y = self.new_act(x.C, OH, OW) if out is None else out
        assert (y.H, y.W, y.C) == (OH, OW, x.C)
        idx = torch.zeros(y.M * x.C, dtype=torch.uint8, device=self.dev)
        self._f("maxpool3s2_fwd", x.ptr(), x.ld, y.ptr(), y.ld, ptr(idx), x.B, x.H, x.W, x.C)

        def build_bwd():
            assert y.gready()
            acc = x.gwrite()
            self._b("maxpool3s2_bwd", (y.gptr(), y.gld, ptr(idx), x.gptr(), x.gld, acc, x.B, x.H, x.W, x.C))

        self._add_builder(build_bwd)
        return y

    def maxpool2(self, x, out=None):
        """nn.MaxPool2d(kernel_size=2, stride=2) (darknet.py:481)."""
        y = self.new_act(x.C, x.H // 2, x.W // 2) if out is None else out
        assert (y.H, y.W, y.C) == (x.H // 2, x.W // 2, x.C)
        idx = torch.zeros(y.M * x.C, dtype=torch.uint8, device=self.dev)
        self._f("maxpool2_fwd", x.ptr(), x.ld, y.ptr(), y.ld, ptr(idx), x.B, x.H, x.W, x.C)

        def build_bwd():
            assert y.gready()
            acc = x.gwrite()
            self._b("maxpool2_bwd", (y.gptr(), y.gld, ptr(idx), x.gptr(), x.gld, acc, x.B, x.H, x.W, x.C))

        self._add_builder(build_bwd)
        return y

    def vgg(self, bb, out3, out4):
        """vgg19() (darknet.py:447-513): the first conv (3 -> 64 at full resolution) as im2col rows (27 -> 32 columns) x GEMM, then
        plain conv-BN-ReLU units and 2x2 max pools; dark3 / dark4 are the pooled outputs of stages 3 / 4, dark5 = conv_add."""
        B, IH, IW = self.B, self.IH, self.IW
        rows = self.new_act(32, IH, IW)
        rows.needs_grad = False
        self._f("im2col_bf16", ptr(self.images), rows.ptr(), 32, B, 3, IH, IW, 3, 1, 1)
        x, feats = rows, []
        for si, stage in enumerate(bb.stages()):
            for m in stage:
                if isinstance(m, enn.ConvBNReLU):
                    x = self.unit(None, x, stem=x is rows, conv=m.conv, bn=m.bn, act=2)
                else:
                    x = self.maxpool2(x, out=out3 if si == 2 else out4 if si == 3 else None)
            feats.append(x)
        x0 = self.unit(None, feats[4], conv=bb.conv_add.conv, bn=bb.conv_add.bn, act=2)
        return feats[2], feats[3], x0

    def res_block(self, blk, x, out=None):
        """ResNet bottleneck (darknet.py:247-271).  The downsample unit is placed first so that, in backward, its 1x1
        stride-2 input gradient (which reaches the even pixels only) accumulates onto conv1's complete one."""
        idn = x
        if blk.downsample is not None:
            idn = self.unit(None, x, conv=blk.downsample[0], bn=blk.downsample[1], act=0)
        t = self.unit(None, x, conv=blk.conv1, bn=blk.bn1, act=2)
        t = self.unit(None, t, conv=blk.conv2, bn=blk.bn2, act=2)
        y = self.unit(None, t, out=out, conv=blk.conv3, bn=blk.bn3, act=0, residual=idn)
        return self.relu(y)

    def resnet(self, bb, out3, out4):
        """The swapped backbone (darknet.py:389-416): stem conv 7x7/2 as im2col rows x GEMM, BN + ReLU, max pool, four
        stages; returns dark3 / dark4 / dark5 (layer2 / layer3 / layer4)."""
        B, IH, IW = self.B, self.IH, self.IW
        rows = self.new_act(152, IH // 2, IW // 2)           # 7*7*3 = 147 columns -> 152
        rows.needs_grad = False
        self._f("im2col_bf16", ptr(self.images), rows.ptr(), 152, B, 3, IH, IW, 7, 2, 3)
        x = self.unit(None, rows, stem=True, conv=bb.conv1, bn=bb.bn1, act=2)
        x = self.maxpool3s2(x)
        feats = []
        for li, layer in enumerate((bb.layer1, bb.layer2, bb.layer3, bb.layer4)):
            for bi, blk in enumerate(layer):
                last = bi == len(layer) - 1
                out = out3 if (last and li == 1) else out4 if (last and li == 2) else None
                x = self.res_block(blk, x, out)
            feats.append(x)
        return feats[1], feats[2], feats[3]

    # ---- DenseNet pieces (darknet.py:515-674) ------------------------------------------------------------------
    def conv_raw(self, conv, x, out):
        """A convolution whose output is stored as it is (no BatchNorm behind it: DenseNet's come in front) into ``out``,
        typically a channel slice of a block's concatenation."""
        home = self.home
        seg = home.by_param[conv.weight]
        k, s = conv.kernel_size[0], conv.stride[0]
        B, H, W, cin, cout = x.B, x.H, x.W, x.C, seg.cout
        assert x.C == seg.cin and (out.H, out.W, out.C) == ((H - 1) // s + 1, (W - 1) // s + 1, cout)
        wf = ptr(home.wf, seg.wf_off)
        self._f("conv_fwd_bf16", x.ptr(), x.ld, wf, out.ptr(), out.ld, 0, 0, 0, None, None, 1, B, H, W, cin, cout, k, s)
        self.unit_acts[conv] = (x, None, out)

        def build_bwd():
            assert out.gready(), "activation without a gradient producer"
            splits = self._wsplits(B, H, W, cin, cout, k, s)
            soff = self._slab_floats
            self._slab_floats += splits * seg.numel
            idx = self._bwd_units
            self._bwd_units += 1
            # the chunk of the concatenation's gradient this conv owns is final here (every later consumer has run) and
            # nothing writes those columns again, so the weight-gradient lane can read it in place
            self._b("@side_wait_main", ())
            self._b("side:conv_wgrad_slab_bf16", (x.ptr(), x.ld, out.gptr(), out.gld, (lambda soff=soff: self.slab.data_ptr() + 4 * soff),
                                                  splits * seg.numel, seg.taps * seg.cin, cout, seg.cin, B, H, W, cin, cout, k, s))
            self._b("@side_record", (idx,))
            self._pending_reduce.append((seg, splits, soff))
            if len(self._pending_reduce) >= WGRAD_REDUCE_GROUP:
                self._flush_reduce()
            if x.needs_grad:
                acc = x.gwrite()
                self._b("conv_dgrad_bf16", (out.gptr(), out.gld, ptr(home.wd, seg.wd_off), x.gptr(), x.gld, acc, B, H, W, cin,
                                            seg.cout_pad, k, s))

        self._add_builder(build_bwd)
        return out

    def pre_bn(self, bn, x, bstats, ld_stats):
        """Pre-activation BatchNorm + ReLU over ``x`` (a prefix of a block's concatenation): the batch statistics are the
        block's (``bstats``, one entry per channel of the whole concatenation, filled as chunks are produced); the layer
        gathers its prefix.  In backward the input gradient ACCUMULATES into the concatenation's gradient."""
        home = self.home
        gam, bet = home.by_param[bn.weight], home.by_param[bn.bias]
        C, M = x.C, x.M
        a = self.new_act(C, x.H, x.W)
        self.pre_bn_inputs[bn] = x
        lstats = self._stats_slot(C)
        save = torch.zeros(2 * C, dtype=torch.float32, device=self.dev)
        sum_g, sum_b = self._sums_slot(C)
        flat, gflat = home.flat, home.gflat
        self._f("stats_gather", bstats, ld_stats, lstats, C, STATS_REPLICAS, ev=False)
        self._f("bn_act_fwd", x.ptr(), x.ld, lstats, STATS_REPLICAS, ptr(flat, gam.off), ptr(flat, bet.off), ptr(bn.running_mean),
                ptr(bn.running_var), ptr(bn.num_batches_tracked), None, ptr(save), a.ptr(), a.ld, None, 0, M, C, float(bn.eps),
                float(bn.momentum), 2,
                ev=("bn_act_infer", (x.ptr(), x.ld, ptr(flat, gam.off), ptr(flat, bet.off), ptr(bn.running_mean),
                                     ptr(bn.running_var), a.ptr(), a.ld, None, 0, M, C, float(bn.eps), 2)))

        def build_bwd():
            assert a.gready(), "activation without a gradient producer"
            acc = x.gwrite()
            self._b("bn_act_bwd_reduce", (a.gptr(), a.gld, x.ptr(), x.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off),
                                          sum_g, sum_b, M, C, 2, STATS_REPLICAS), reads=a)
            self._b("bn_act_bwd_apply_acc" if acc else "bn_act_bwd_apply",
                    (a.gptr(), a.gld, x.ptr(), x.ld, ptr(save), ptr(flat, gam.off), ptr(flat, bet.off), sum_g, sum_b,
                     ptr(gflat, gam.off), ptr(gflat, bet.off), x.gptr(), x.gld, M, C, 2, STATS_REPLICAS), writes=(gam, bet))

        self._add_builder(build_bwd)
        return a

    def chan_dropout(self, x, layer):
        """nn.Dropout2d(0.3) of a dense layer (darknet.py:574-576), training only: the keep factors of the step live in
        self.drop_keep[layer] (drawn by draw_dropout())."""
        keep = self.drop_keep[layer]
        self._f("chanscale", x.ptr(), x.ld, ptr(keep), x.B, x.H * x.W, x.C, ev=False)

        def build_bwd():
            assert x.gready()
            _PENDING_GW.append(x.gregion())
            self._b("chanscale", (x.gptr(), x.gld, ptr(keep), x.B, x.H * x.W, x.C))

        self._add_builder(build_bwd)

    def dense_block(self, blk, cat, c0, bstats, base):
        """DenseBlock (darknet.py:582-597): ``cat`` [M, c0 + 32 n] holds the block input in its first c0 channels (its
        statistics already in ``bstats``); every layer appends 32 channels and their statistics."""
        ct = cat.C
        for i, lay in enumerate(blk.denseblock):
            ci = c0 + 32 * i
            cb1, cb2 = lay.conv_block
            a = self.pre_bn(cb1.bn, cat.slice(0, ci), bstats, ct)
            b = self.unit(None, a, conv=cb1.conv, bn=cb2.bn, act=2)
            chunk = cat.slice(ci, 32)
            self.conv_raw(cb2.conv, b, chunk)
            if lay.drop_rate > 0:
                self.chan_dropout(chunk, base + i)
            self._f("colstats", chunk.ptr(), chunk.ld, (lambda ci=ci: bstats() + 8 * ci), ct, chunk.M, 32, ev=False)
        return cat

    def densenet(self, bb, out3, out4):
        B, IH, IW = self.B, self.IH, self.IW
        rows = self.new_act(152, IH // 2, IW // 2)
        rows.needs_grad = False
        self._f("im2col_bf16", ptr(self.images), rows.ptr(), 152, B, 3, IH, IW, 7, 2, 3)
        x = self.unit(None, rows, stem=True, conv=bb.stem[0].conv, bn=bb.stem[0].bn, act=2)
        n_layers = sum(len(b.denseblock) for b in (bb.D1, bb.D2, bb.D3, bb.D4))
        self.drop_keep = torch.ones(n_layers, B, 32, dtype=torch.float32, device=self.dev)
        self.drop_p = _drop_rate_of([l for b in (bb.D1, bb.D2, bb.D3, bb.D4) for l in b.denseblock])
        H, W = IH // 4, IW // 4
        base, cat, feats = 0, None, []
        for bi, (blk, tr) in enumerate(((bb.D1, bb.T1), (bb.D2, bb.T2), (bb.D3, bb.T3), (bb.D4, None))):
            c0 = 64 if bi == 0 else cat.C // 2
            ncat = self.new_act(c0 + 32 * len(blk.denseblock), H, W)
            bstats = self._stats_slot(ncat.C)
            head = ncat.slice(0, c0)
            if bi == 0:
                self.maxpool3s2(x, out=head)
            else:
                t = self.new_act(c0, 2 * H, 2 * W)
                self.conv_raw(tr_prev.trans[0].conv, self.pre_bn(tr_prev.trans[0].bn, cat, prev_stats, cat.C), t)
                self.avgpool2(t, head)
            self._f("colstats", head.ptr(), head.ld, bstats, ncat.C, head.M, c0, ev=False)
            cat = self.dense_block(blk, ncat, c0, bstats, base)
            base += len(blk.denseblock)
            feats.append(cat)
            tr_prev, prev_stats = tr, bstats
            H, W = H // 2, W // 2
        c3 = self.unit(None, feats[1], out=out3, conv=bb.baseconv1.conv, bn=bb.baseconv1.bn, act=2)
        c4 = self.unit(None, feats[2], out=out4, conv=bb.baseconv2.conv, bn=bb.baseconv2.bn, act=2)
        return c3, c4, feats[3]

    def avgpool2(self, x, y):
        """nn.AvgPool2d(2, 2) into ``y`` (the head of the next block's concatenation)."""
        self._f("avgpool2_fwd", x.ptr(), x.ld, y.ptr(), y.ld, x.B, x.H, x.W, x.C)

        def build_bwd():
            assert y.gready()
            acc = x.gwrite()
            self._b("avgpool2_bwd", (y.gptr(), y.gld, x.gptr(), x.gld, acc, x.B, x.H, x.W, x.C))

        self._add_builder(build_bwd)
        return y

    def draw_dropout(self):
        """New Dropout2d keep factors for the next training forward (torch's generator on the device: no host sync)."""
        if getattr(self, "drop_keep", None) is not None and not getattr(self, "fixed_dropout", False):
            self.drop_keep.bernoulli_(1.0 - self.drop_p).div_(1.0 - self.drop_p)

    def csp_merged(self, mod, x, out=None):
        """CSP layer without shortcuts with conv1 and conv2 as one GEMM: P = [m(x_1) | x_2 | x_1]; the merged unit writes
        [x_2 | x_1] (columns h..3h), the bottleneck chain reads x_1 and ends in columns 0..h, conv3 reads columns 0..2h."""
        h = mod.conv1.conv.out_channels
        _same_bn(mod.conv2.bn, mod.conv1.bn)
        P = self.new_act(3 * h, x.H, x.W)
        both = P.slice(h, 2 * h)
        self.unit(None, x, out=both, conv=mod.conv2.conv, bn=mod.conv2.bn, act=mod.conv2.act_code, bn2=mod.conv1.bn)
        xa, za, ya = self.unit_acts.pop(mod.conv2.conv)
        self.unit_acts[mod.conv2] = (xa, za.slice(0, h), ya.slice(0, h))      # per-module views (tests walk unit_acts)
        self.unit_acts[mod.conv1] = (xa, za.slice(h, h), ya.slice(h, h))
        t = x1 = P.slice(2 * h, h)
        n = len(mod.m)
        if any(blk.use_add for blk in mod.m):
            # with shortcuts d(x_1) shares storage with the bottleneck chain's gradient (residual aliasing), which is not where
            # the merged unit reads it: one copy into columns 2h..3h of the gradient, after the chain's backward
            raw = P.slice(2 * h, h)

            def build_copy():
                assert x1._alias is not None and x1._groot().buf.gwritten
                acc = raw.gwrite()
                self._b("f32_rows_copy" if self.f32 else "rows_copy", (x1.gptr(), x1.gld, raw.gptr(), raw.gld, acc, x1.M, h))

            self._add_builder(build_copy)
        for i, blk in enumerate(mod.m):
            u = self.unit(blk.conv1, t, bn_in=i > 0, below_in=i > 0)
            t = self.conv_block(blk.conv2, u, out=P.slice(0, h) if i == n - 1 else None, residual=t if blk.use_add else None, x_single=True)
        return self.unit(mod.conv3, P.slice(0, 2 * h), out=out)

    def spp(self, mod, x):
        """SPPBottleneck: conv1 -> cat(x, pool5, pool9, pool13) -> conv2 (network_blocks.py:139-144)."""
        h = self.home.by_param[mod.conv1.conv.weight].cout
        cat = self.new_act(4 * h, x.H, x.W)
        t = self.unit(mod.conv1, x, out=cat.slice(0, h))
        y5, y9, y13 = cat.slice(h, h), cat.slice(2 * h, h), cat.slice(3 * h, h)
        idx = torch.zeros(3 * t.M * h, dtype=torch.int32 if self.f32 else torch.uint8, device=self.dev)
        if self.f32:
            self._f("f32_spp_fwd", t.ptr(), t.ld, y5.ptr(), y9.ptr(), y13.ptr(), cat.ld, ptr(idx), t.B, t.H, t.W, h, ev=False)
        else:
            scratch = torch.zeros(9 * t.M * h, dtype=torch.uint8, device=self.dev)
            self._keep.append(scratch)
            self._f("spp_fwd", t.ptr(), t.ld, y5.ptr(), y9.ptr(), y13.ptr(), cat.ld, ptr(idx), t.B, t.H, t.W, h, ptr(scratch))

        def build_bwd():
            assert y5.gready() and y9.gready() and y13.gready()
            acc = t.gwrite()
            self._b("f32_spp_bwd" if self.f32 else "spp_bwd", (y5.gptr(), y9.gptr(), y13.gptr(), y5.gld, ptr(idx), t.gptr(), t.gld, acc,
                                                                t.B, t.H, t.W, h))

        self._add_builder(build_bwd)
        return self.unit(mod.conv2, cat)

    def up2(self, x, y):
        pre = "f32_" if self.f32 else ""
        self._f(pre + "upsample2_fwd", x.ptr(), x.ld, y.ptr(), y.ld, x.B, x.H, x.W, x.C, ev=False if self.f32 else None)

        def build_bwd():
            assert y.gready()
            acc = x.gwrite()
            self._b(pre + "upsample2_bwd", (y.gptr(), y.gld, x.gptr(), x.gld, acc, x.B, x.H, x.W, x.C))

        self._add_builder(build_bwd)

    def head_level(self, head, k, feat, a0):
        """stem -> {cls branch -> cls_preds, reg branch -> reg_preds + obj_preds} -> decode (yolo_head_24p.py:150-189)."""
        home, B, C = self.home, self.B, self.C
        H, W, s = feat.H, feat.W, float(head.strides[k])
        self.levels.append((H, W, s))
        self._cur_tag = ("head", k)
        x = self.unit(head.stems[k], feat)
        if head_is_merged(head, self.options):
            c0, r0 = head.cls_convs[k][0], head.reg_convs[k][0]
            hc = c0.conv.out_channels
            _same_bn(c0.bn, r0.bn)
            both = self.unit(None, x, conv=c0.conv, bn=c0.bn, act=c0.act_code, bn2=r0.bn, x_single=True)   # [class branch | regression branch], one GEMM (N = 2h)
            xa, za, ya = self.unit_acts.pop(c0.conv)
            self.unit_acts[c0] = (xa, za.slice(0, hc), ya.slice(0, hc))
            self.unit_acts[r0] = (xa, za.slice(hc, hc), ya.slice(hc, hc))
            cf = self.unit(head.cls_convs[k][1], both.slice(0, hc), x_single=True)      # each branch owns its half of the merged unit's channels
            rf = self.unit(head.reg_convs[k][1], both.slice(hc, hc), x_single=True)
        else:
            cf = self.conv_block(head.cls_convs[k][1], self.conv_block(head.cls_convs[k][0], x), x_single=True)
            rf = self.conv_block(head.reg_convs[k][1], self.conv_block(head.reg_convs[k][0], x), x_single=True)
        ro_seg = home.by_param[head.reg_preds[k].weight]
        ro_b = home.by_param[head.reg_preds[k].bias]
        cl_seg = home.by_param[head.cls_preds[k].weight]
        cl_b = home.by_param[head.cls_preds[k].bias]
        hch = x.C
        M = B * H * W
        out = self.outputs
        flat, gflat = home.flat, home.gflat
        if self.f32:
            return self._head_preds_f32(rf, cf, ro_seg, ro_b, cl_seg, cl_b, hch, M, a0, H, W, s)
        self._f("conv_fwd_bf16", rf.ptr(), rf.ld, ptr(home.wf, ro_seg.wf_off), ptr(out), self.ncols, 1, self.A, a0,
                ptr(flat, ro_b.off), None, 1, B, H, W, hch, 27, 1, 1)
        self._f("conv_fwd_bf16", cf.ptr(), cf.ld, ptr(home.wf, cl_seg.wf_off), ptr(out, 27), self.ncols, 1, self.A, a0,
                ptr(flat, cl_b.off), None, 1, B, H, W, hch, C, 1, 1)
        self._f("head_decode_fwd", ptr(out), B, self.A, a0, H, W, s, self.ncols, Dyn(self.dyn, "origin"),
                ev=("head_decode_eval", (ptr(out), B, self.A, a0, H, W, s, self.ncols)))
        ldc = _r8(C)
        d_ro = torch.zeros(M * 32, dtype=BF16, device=self.dev)
        d_cl = torch.zeros(M * ldc, dtype=BF16, device=self.dev)
        self.head_grads.append((H * W, s, d_ro, d_cl))           # what ep24_loss_grad_decode writes directly (ep24.train)

        def build_bwd():
            dout = Dyn(self.dyn, "dout")
            self._b("head_decode_bwd", (dout, ptr(out), ptr(d_ro), ptr(d_cl), B, self.A, a0, H, W, s, self.ncols,
                                        Dyn(self.dyn, "d_origin")))
            # bias and weight gradients of the prediction convs: side lane, slab partials folded by the next reduce launch
            fn = _lib.lib().fn

            def emit_pred_grads():
                self._b("@side_wait_main", ())
                for dsrc, ld_d, bseg, n in ((d_ro, 32, ro_b, 27), (d_cl, ldc, cl_b, C)):
                    sp = fn["ep24_colsum_splits"](M)
                    soff = self._slab_floats
                    self._slab_floats += sp * n + (-(sp * n)) % 4
                    self._b("side:colsum_slab", (ptr(dsrc), ld_d, (lambda soff=soff: self.slab.data_ptr() + 4 * soff), M, n))
                    self._pending_reduce.append((bseg, sp, soff))
                for feat_in, dsrc, ld_d, wseg, n, npad in ((rf, d_ro, 32, ro_seg, 27, 32), (cf, d_cl, ldc, cl_seg, C, ldc)):
                    sp = self._wsplits(B, H, W, hch, npad, 1, 1)
                    soff = self._slab_floats
                    self._slab_floats += sp * wseg.numel
                    self._b("side:conv_wgrad_slab_bf16", (feat_in.ptr(), feat_in.ld, ptr(dsrc), ld_d,
                                                          (lambda soff=soff: self.slab.data_ptr() + 4 * soff), sp * wseg.numel, hch, n, hch,
                                                          B, H, W, hch, npad, 1, 1))
                    self._pending_reduce.append((wseg, sp, soff))

            if self._force_side:
                self._deferred.append(emit_pred_grads)
            else:
                emit_pred_grads()
            self._b("conv_dgrad_bf16", (ptr(d_ro), 32, ptr(home.wd, ro_seg.wd_off), rf.gptr(), rf.gld, rf.gwrite(), B, H,
                                        W, hch, 32, 1, 1))
            self._b("conv_dgrad_bf16", (ptr(d_cl), ldc, ptr(home.wd, cl_seg.wd_off), cf.gptr(), cf.gld, cf.gwrite(), B, H,
                                        W, hch, ldc, 1, 1))

        self._add_builder(build_bwd)

    def _head_preds_f32(self, rf, cf, ro_seg, ro_b, cl_seg, cl_b, hch, M, a0, H, W, s):
        """Prediction convs + decode of one level in the fp32 parity mode (weights / biases read in place from the master)."""
        home, B, C, out = self.home, self.B, self.C, self.outputs
        flat, gflat = home.flat, home.gflat
        self._f("f32_conv", rf.ptr(), rf.ld, ptr(flat, ro_seg.off), hch, hch, ptr(out), self.ncols, self.A, a0, ptr(flat, ro_b.off), 0,
                B, H, W, hch, 27, 1, 1, 0, ev=False)
        self._f("f32_conv", cf.ptr(), cf.ld, ptr(flat, cl_seg.off), hch, hch, ptr(out, 27), self.ncols, self.A, a0, ptr(flat, cl_b.off), 0,
                B, H, W, hch, C, 1, 1, 0, ev=False)
        self._f("head_decode_fwd", ptr(out), B, self.A, a0, H, W, s, self.ncols, Dyn(self.dyn, "origin"), ev=False)
        ldc = _r8(C)
        d_ro = torch.zeros(M * 32, dtype=torch.float32, device=self.dev)
        d_cl = torch.zeros(M * ldc, dtype=torch.float32, device=self.dev)

        def build_bwd():
            self._b("f32_head_decode_bwd", (Dyn(self.dyn, "dout"), ptr(out), ptr(d_ro), ptr(d_cl), B, self.A, a0, H, W, s, self.ncols,
                                            Dyn(self.dyn, "d_origin")))
            self._b("f32_colsum", (ptr(d_ro), 32, ptr(gflat, ro_b.off), M, 27), writes=(ro_b,))
            self._b("f32_colsum", (ptr(d_cl), ldc, ptr(gflat, cl_b.off), M, C), writes=(cl_b,))
            self._b("f32_conv_wgrad", (rf.ptr(), rf.ld, ptr(d_ro), 32, ptr(gflat, ro_seg.off), hch, hch, B, H, W, hch, 27, 1, 1), writes=(ro_seg,))
            self._b("f32_conv_wgrad", (cf.ptr(), cf.ld, ptr(d_cl), ldc, ptr(gflat, cl_seg.off), hch, hch, B, H, W, hch, C, 1, 1), writes=(cl_seg,))
            self._b("f32_conv", (ptr(d_ro), 32, ptr(flat, ro_seg.off), hch, hch, rf.gptr(), rf.gld, 0, 0, None, rf.gwrite(), B, H, W, hch, 27, 1, 1, 1))
            self._b("f32_conv", (ptr(d_cl), ldc, ptr(flat, cl_seg.off), hch, hch, cf.gptr(), cf.gld, 0, 0, None, cf.gwrite(), B, H, W, hch, C, 1, 1, 1))

        self._keep += [d_ro, d_cl]
        self._add_builder(build_bwd)

    # ---- execution ----------------------------------------------------------------------------------
    def _run(self, lst):
        """Launch a list on the current stream.  Entries named ``side:<fn>`` go to a second stream; the control
        entries ``@side_wait_main`` / ``@side_record(k)`` / ``@main_wait_side(k)`` order the two with events.  The
        call returns with the side stream joined.  Under hipGraph capture everything stays on the capturing stream:
        a captured graph with ~240 cross-stream edges faulted at replay on ROCm 7.2, so
        ep24.train replays graphs for the forward/loss and update phases and runs backward through this path."""
        main = torch.cuda.current_stream()
        s_main = main.cuda_stream
        capturing = torch.cuda.is_current_stream_capturing()
        lanes = self.use_side and (not capturing or self.capture_side)
        keep = self._events if capturing else []      # events that are edges of a graph under capture must outlive it
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        side, s_side = self._side, self._side.cuda_stream
        fn = _lib.lib().fn
        events, used_side = {}, False
        for name, args in lst:
            if name[0] == "@":
                if not lanes:
                    continue
                if name == "@side_wait_main":
                    ev = torch.cuda.Event()
                    keep.append(ev)
                    ev.record(main)
                    side.wait_event(ev)
                    used_side = True
                elif name == "@side_record":
                    ev = torch.cuda.Event()
                    keep.append(ev)
                    ev.record(side)
                    events[args[0]] = ev
                elif name == "@main_wait_side":
                    if args[0] == "all":
                        if used_side:
                            main.wait_stream(side)
                    else:
                        ev = events.get(args[0])
                        if ev is not None:
                            main.wait_event(ev)
                continue
            s = s_main
            if name.startswith("side:"):
                name = name[5:]
                if lanes:
                    s = s_side
                    used_side = True
            if name == "head_decode_bwd" and self.skip_decode_bwd:
                continue
            rc = fn["ep24_" + name](*[a.get() if isinstance(a, Dyn) else a for a in args], s)
            if rc != 0:
                raise _lib.Ep24Error("ep24_%s failed (%d): %s" % (name, rc, _lib.lib().last_error()))
        if used_side:
            main.wait_stream(side)

    def lane_lists(self, lo, hi):
        """Backward entries [lo, hi) split into the main lane and the weight-gradient lane (control entries dropped).
        The side entries of a segment depend only on main entries of the same or earlier segments."""
        main, side = [], []
        for name, args in self.bwd[lo:hi]:
            if name[0] == "@":
                continue
            if name.startswith("side:"):
                side.append((name[5:], args))
            else:
                main.append((name, args))
        return main, side

    def run_lane(self, lst):
        """Launch plain entries on the current stream (capturable)."""
        s = stream_ptr()
        fn = _lib.lib().fn
        for name, args in lst:
            if name == "head_decode_bwd" and self.skip_decode_bwd:
                continue
            rc = fn["ep24_" + name](*[a.get() if isinstance(a, Dyn) else a for a in args], s)
            if rc != 0:
                raise _lib.Ep24Error("ep24_%s failed (%d): %s" % (name, rc, _lib.lib().last_error()))

    def decode_levels(self):
        """HOST table for ep24_loss_grad_decode: per head level (cells per image, the stride's float32 bits, d_regobj, d_cls)."""
        import struct
        if getattr(self, "_decode_levels", None) is None:
            rows = [[hw, struct.unpack("<I", struct.pack("<f", float(s)))[0], d_ro.data_ptr(), d_cl.data_ptr()] for hw, s, d_ro, d_cl in self.head_grads]
            self._decode_levels = torch.tensor(rows, dtype=torch.int64)
        return self._decode_levels

    def zero_step_buffers(self):
        s = stream_ptr()
        call("memset_zero", ptr(self.stats), self.stats.numel() * 8, s)
        call("memset_zero", ptr(self.bnsums), self.bnsums.numel() * 8, s)

    def set_use_l1(self, on):
        """use_l1 of the head (yolo_head_24p.py:179-188): the decode launches also keep the raw regression outputs in
        self.origin.  Launch lists are unchanged (the pointers are run-time arguments); captured graphs hold the
        pointer values, so ep24.train re-captures when this flips."""
        if on and self.origin is None:
            self.origin = torch.zeros(self.B, self.A, 26, dtype=torch.float32, device=self.dev)
        self.dyn["origin"] = self.origin.data_ptr() if on else None
        if not on:
            self.dyn["d_origin"] = None

    def forward(self, images=None):
        """Train-mode forward into self.outputs ([B,A,27+C] fp32, decoded)."""
        if images is not None:
            self.images.copy_(images)
        if not torch.cuda.is_current_stream_capturing():
            self.draw_dropout()                      # a captured step draws before it replays (ep24.train)
        self.zero_step_buffers()
        if not self.f32:
            self.home.pack()
        self._run(self.fwd)
        return self.outputs

    def forward_eval(self, images=None):
        """Eval-mode forward (BatchNorm with running statistics, sigmoid on obj / class): decoded [B,A,27+C] fp32, the
        tensor the reference's YOLOXHead returns with decode_in_inference (yolo_head_24p.py:190-210)."""
        if self.f32:
            raise NotImplementedError("ep24: the fp32 parity mode runs the training-mode plan only")
        if images is not None:
            self.images.copy_(images)
        self.home.pack()
        self.fold()
        self._run(self.fwd_eval)
        return self.outputs

    def fold(self):
        """Fold every unit's BatchNorm (running statistics) into its packed weights and a bias: two launches for the network."""
        if self._fold_units:
            d, pf, cp, eps, n, tot, totc = self._fold_desc
            call("fold_bn", ptr(self.home.flat), ptr(self.home.bflat), ptr(d), ptr(pf), ptr(cp), ptr(eps), n, tot, totc,
                 ptr(self.fold_w), ptr(self.fold_b), stream_ptr())

    def backward(self, dout, d_origin=None):
        """Accumulates parameter gradients into the flat gradient buffer; dout [B,A,27+C] fp32 contiguous, d_origin
        [B,A,26] the gradient of the L1 branch with respect to the raw regression outputs (or None)."""
        self.dyn["dout"] = dout.data_ptr()
        self.dyn["d_origin"] = None if d_origin is None else d_origin.data_ptr()
        self.skip_decode_bwd = False              # the eager API hands over the dense gradient: the decode backward launches run
        self._run(self.bwd)

    # ---- nn.Module / autograd entry -----------------------------------------------------------------
    def run_module_forward(self, x, train):
        if not train:
            # inference (show_24p.py): the reference calls model.eval() first, so BatchNorm uses its running statistics
            if self.model.training:
                raise NotImplementedError("ep24: model(x, train=False) on a model in training mode (batch-statistics BN with "
                                          "the eval head) is not implemented - call model.eval() first, as show_24p.py does")
            with torch.no_grad():
                return self.forward_eval(x).clone()
        use_l1 = bool(getattr(self.model.head, "use_l1", False))
        self.set_use_l1(use_l1)
        out, origin = _NetFn.apply(x, self, use_l1, *list(self.home.views.keys()))
        origin_preds, a0 = [], 0
        if use_l1:                                   # per-level [B, H*W, 26] slices, yolo_head_24p.py:179-188
            for H, W, _ in self.levels:
                origin_preds.append(origin[:, a0:a0 + H * W])
                a0 += H * W
        return self.x_shifts, self.y_shifts, self.exp_strides, out, origin_preds


# ------------------------------------------------------------------------------------------------ sub-plans
def _to_act(act, x):
    """NCHW tensor -> the Act's NHWC slot."""
    v = act.buf.t.view(act.buf.rows, act.buf.ld)[:, act.c0:act.c0 + act.C]
    v.copy_(x.permute(0, 2, 3, 1).reshape(-1, act.C))


def _from_act(act, grad=False):
    """The Act's (gradient's) NHWC slot -> NCHW fp32 tensor."""
    if grad:
        r = act._groot()
        v = r.buf.grad().view(r.buf.rows, r.buf.ld)[:, r.c0:r.c0 + r.C]
    else:
        v = act.buf.t.view(act.buf.rows, act.buf.ld)[:, act.c0:act.c0 + act.C]
    return v.reshape(act.B, act.H, act.W, act.C).permute(0, 3, 1, 2).float().contiguous()


class SubEngine(Engine):
    """Launch plan of ONE module of the tree run on its own - what ``BaseConv.forward`` / ``CSPLayer.forward`` /
    ``CSPDarknet.forward`` / ``YOLOPAFPN.forward`` / ``YOLOXHead.forward`` of the reference compute
    (network_blocks.py:50-51,179-185,139-144, darknet.py:165-177, yolo_pafpn.py:83-124, yolo_head_24p.py:143-210).
    Same building blocks, buffers and kernels as the whole-network plan; inputs / outputs cross as NCHW fp32 tensors."""

    IMAGE_KINDS = ("focus", "darknet", "pafpn", "backbone", "stem_unit")

    def __init__(self, mod, kind, shapes, batch, dtype=BF16):
        self.kind, self.shapes = kind, shapes                 # shapes: (C, H, W) of every input tensor
        size = (shapes[0][1], shapes[0][2]) if kind in self.IMAGE_KINDS else (32, 32)
        super().__init__(mod, batch, size, dtype)

    def _build(self):
        mod, kind, B = self.model, self.kind, self.B
        self.inputs, self.outs = [], []
        if kind in self.IMAGE_KINDS:
            self.images = torch.zeros(B, 3, self.IH, self.IW, dtype=torch.float32, device=self.dev)
        else:
            for (C, H, W) in self.shapes:
                self.inputs.append(self.new_act(C, H, W))
        x = self.inputs[0] if self.inputs else None
        home = self.home
        if kind == "baseconv":
            if len(home.by_param[mod.conv.weight].params) != 1:
                raise NotImplementedError("ep24: this BaseConv runs as one merged unit with its sibling (conv1 / conv2 of a CSP layer, the "
                                          "first convs of the head branches): call the enclosing module, or build the model with PlanOptions(merge_csp=False, merge_head=False)")
            self.outs = [self.unit_dw(mod, x) if mod.conv.groups > 1 else self.unit(mod, x)]
        elif kind == "dwconv":                                # DWConv.forward (network_blocks.py:73-76)
            self.outs = [self.conv_block(mod, x)]
        elif kind == "focus":
            self.outs = [self.focus_stem(mod)]
        elif kind == "bottleneck":
            self.outs = [self.conv_block(mod.conv2, self.unit(mod.conv1, x), residual=x if mod.use_add else None, x_single=True)]
        elif kind == "csp":
            self.outs = [self.csp(mod, x)]
        elif kind == "spp":
            self.outs = [self.spp(mod, x)]
        elif kind == "darknet":
            self.outs = list(self.build_backbone(mod))
        elif kind == "pafpn":
            self.outs = list(self.build_neck(mod, *self.build_neck_inputs(mod)))
        elif kind == "head":
            self.build_head(mod, self.inputs)
        elif kind == "backbone":                              # resnet50() / densenet121() / vgg19() on their own (darknet.py:389-674)
            self.outs = list(self.build_backbone(mod))
        elif kind == "stem_unit":                             # a conv-BN-ReLU unit over images: im2col rows x GEMM, as in the full plans
            kk, st, pd = mod.conv.kernel_size[0], mod.conv.stride[0], mod.conv.padding[0]
            cols = _r8(kk * kk * 3)
            OH, OW = (self.IH + 2 * pd - kk) // st + 1, (self.IW + 2 * pd - kk) // st + 1
            rows = self.new_act(cols, OH, OW)
            rows.needs_grad = False
            self._f("im2col_bf16", ptr(self.images), rows.ptr(), cols, B, 3, self.IH, self.IW, kk, st, pd)
            self.outs = [self.unit(None, rows, stem=True, conv=mod.conv, bn=mod.bn, act=2)]
        elif kind == "unit_relu":                             # BaseConv_DN / ConvBNReLU (darknet.py:432-445,518-529)
            self.outs = [self.unit(None, x, conv=mod.conv, bn=mod.bn, act=2)]
        elif kind == "resblock":                              # ResBottleneck (darknet.py:247-271)
            self.outs = [self.res_block(mod, x)]
        elif kind in ("convblock", "transition", "denselayer", "denseblock"):
            self.outs = [self._dense_piece(mod, kind)]
        else:
            raise NotImplementedError(kind)
        for o in self.outs:
            o.gwrite()                                    # the caller is the consumer: it provides d(out)
        self._finalize()

    def _dense_piece(self, mod, kind):
        """ConvBlock / Transition / DenseLayer / DenseBlock of the DenseNet backbone on their own (darknet.py:532-597).  The input sits
        in the head of a concatenation buffer, its batch statistics come from one ``colstats`` launch, and the pieces are the ones
        the full plan uses (pre-activation BatchNorm over a prefix of the concatenation, raw convs into their channel slots)."""
        C0, H, W = self.shapes[0]
        nl = len(mod.denseblock) if kind == "denseblock" else (1 if kind == "denselayer" else 0)
        cat = self.new_act(C0 + 32 * nl, H, W)
        head = cat.slice(0, C0)
        self.inputs = [head]                                  # replaces the stand-alone input buffer: the input IS the head slot
        bstats = self._stats_slot(cat.C)
        self.drop_keep = torch.ones(max(nl, 1), self.B, 32, dtype=torch.float32, device=self.dev)
        self.drop_p = _drop_rate_of(list(mod.denseblock) if kind == "denseblock" else [mod] if kind == "denselayer" else [])
        self._f("colstats", head.ptr(), head.ld, bstats, cat.C, head.M, C0, ev=False)
        if kind == "denseblock":                              # x -> cat(x, layer_0(x), layer_1(cat), ...)
            if not any(l.drop_rate > 0 for l in mod.denseblock):
                self.drop_keep = None
            return self.dense_block(mod, cat, C0, bstats, 0)
        if kind == "denselayer":                              # x -> the 32 new channels (after Dropout2d in training mode)
            blk = torch.nn.Module()
            blk.denseblock = [mod]
            if mod.drop_rate <= 0:
                self.drop_keep = None
            self.dense_block(blk, cat, C0, bstats, 0)
            return cat.slice(C0, 32)
        cb = mod.trans[0] if kind == "transition" else mod
        t = self.new_act(cb.conv.out_channels, H, W)
        self.conv_raw(cb.conv, self.pre_bn(cb.bn, head, bstats, cat.C), t)
        if kind == "convblock":
            return t
        out = self.new_act(t.C, H // 2, W // 2)
        return self.avgpool2(t, out)

    def load_inputs(self, tensors):
        if self.kind in self.IMAGE_KINDS:
            self.images.copy_(tensors[0])
        else:
            for a, x in zip(self.inputs, tensors):
                _to_act(a, x)


class _SubFn(torch.autograd.Function):
    """A sub-plan as one autograd node (inputs: the module's input tensors, then its parameters)."""

    @staticmethod
    def forward(ctx, eng, n_in, *args):
        ctx.eng, ctx.n_in = eng, n_in
        eng.load_inputs(args[:n_in])
        eng.forward()
        if eng.kind == "head":
            return (eng.outputs.clone(),)
        return tuple(_from_act(o) for o in eng.outs)

    @staticmethod
    def backward(ctx, *grads):
        eng = ctx.eng
        eng.home.bind_grads()
        if eng.kind == "head":
            eng.backward(grads[0].contiguous())
        else:
            for o, g in zip(eng.outs, grads):
                r = o._groot()
                v = r.buf.grad().view(r.buf.rows, r.buf.ld)[:, r.c0:r.c0 + r.C]
                v.copy_(g.permute(0, 2, 3, 1).reshape(-1, o.C))
            eng.backward(torch.zeros(1, device=eng.dev))
        gin = tuple(_from_act(a, grad=True) for a in eng.inputs) if eng.kind not in SubEngine.IMAGE_KINDS else (None,)
        return (None, None) + gin + (None,) * (len(ctx.needs_input_grad) - 2 - len(gin))


def run_submodule(mod, kind, tensors, train=True):
    """``forward`` of a module of the tree on its own.  tensors: NCHW tensors on the GPU.  Training mode (batch-statistics
    BatchNorm, differentiable) unless ``mod.training`` is off, which runs the eval-mode list (running statistics, no grad)."""
    _lib.require_gpu()
    for x in tensors:
        if not x.is_cuda or x.dim() != 4:
            raise _lib.Ep24Error("ep24: %s.forward takes [B,C,H,W] tensors on the GPU (no CPU fallback on the product path)" % type(mod).__name__)
    dtype = getattr(mod, "compute_dtype", BF16)
    key = (kind, tuple(tuple(x.shape) for x in tensors), dtype)
    cache = mod.__dict__.setdefault("_ep24_sub", {})
    if key not in cache:
        cache[key] = SubEngine(mod, kind, [tuple(x.shape[1:]) for x in tensors], int(tensors[0].shape[0]), dtype)
    eng = cache[key]
    if kind == "head":
        use_l1 = bool(getattr(mod, "use_l1", False)) and train
        eng.set_use_l1(use_l1)
    if not mod.training or (kind == "head" and not train):
        with torch.no_grad():
            eng.load_inputs(tensors)
            eng.forward_eval()
            return (eng.outputs.clone(),) if kind == "head" else tuple(_from_act(o) for o in eng.outs)
    return _SubFn.apply(eng, len(tensors), *tensors, *list(eng.home.views.keys()))


class _NetFn(torch.autograd.Function):
    """The whole network as ONE autograd node: forward / backward are the engine's launch lists."""

    @staticmethod
    def forward(ctx, x, eng, use_l1, *params):
        ctx.eng, ctx.use_l1 = eng, use_l1
        out = eng.forward(x.float().contiguous()).clone()
        origin = eng.origin.clone() if use_l1 else out.new_zeros(0)
        if not use_l1:
            ctx.mark_non_differentiable(origin)
        return out, origin

    @staticmethod
    def backward(ctx, dout, d_origin):
        eng = ctx.eng
        eng.home.bind_grads()
        eng.backward(dout.contiguous(), d_origin.contiguous() if ctx.use_l1 and d_origin is not None else None)
        # parameter gradients were accumulated in place into the flat buffer that every p.grad views
        return (None, None, None) + (None,) * len(eng.home.views)
