"""Per-model plan options.

Everything that changes WHICH kernels a plan launches, how its parameters are laid out or how a captured step is cut into
lanes is an explicit, immutable option object attached to the model it applies to - never a process-wide switch or an
environment variable read inside the library: two engines in one process (a test that compares two plans, two threads that
each build a model) cannot see each other's settings.

    from ep24.options import PlanOptions, set_options
    set_options(model, PlanOptions(merge_csp=False))      # before the model's first forward / TrainStep

``PlanOptions.parse("merge_csp=0,forward_lanes=3")`` is the text form tools and ``bench.py --plan`` use for A/B runs.
"""
from dataclasses import dataclass, fields, replace
from typing import ClassVar, Optional, Tuple


@dataclass(frozen=True)
class PlanOptions:
    # ---- parameter layout + launch lists (ep24.engine) ----
    merge_csp: bool = True             # conv1 + conv2 of a CSP layer as ONE GEMM with one BatchNorm launch
    merge_csp_shortcut: bool = True    # ... also in the backbone's CSP layers with shortcuts (one gradient copy each)
    merge_head: bool = True            # the first 3x3 conv of the head's class / regression branches as one GEMM (N = 2h)
    parallel_head: bool = True         # backward of head levels 1, 2 on the weight-gradient lane
    capture_side: bool = False         # cross-stream edges inside captured graphs (experimental; faulted on ROCm 7.2)
    fold_bn_eval: bool = True          # eval mode: BatchNorm folded into the conv (one launch per unit)
    conv_kernel_opts: int = 0          # bit 0: 3x3 stride-1 layers through the generic tiled kernel, bit 1: 8-byte epilogue stores
    wgrad_group_steps: int = 100       # 64-pixel steps a workgroup of a grouped weight-gradient launch keeps at least (swept again at the end of round 5,
                                       # when the lane had become co-critical: 60 / 75 / 90 / 100 / 120 / 160 / 250 -> 19.94 / 19.55 / 19.41 / 19.44 / 19.44 / 19.58 / 19.77 ms
                                       # for the headline network; 90 costs the VGG backbone 7 % (484 against 521 - 523 images/s at 100 / 120): 100)
    group_wgrad: bool = True           # weight gradients of a backward segment's layers as grouped launches per tile class (round 5)
    fuse_bn_reduce: bool = False       # a 3x3 stride-1 input gradient that is the only consumer of the unit below takes that unit's
                                       # BatchNorm-backward sums in its epilogue (no reduce launch for it).  Measured slower in the step
                                       # (899 against 912 images/s, same box): kept for A/B runs only
    fuse_bn_bwd: bool = False          # the two BatchNorm-backward passes of a trunk unit as ONE launch with a grid-wide wait (round 5, A/B)
    fuse_bn_stream: bool = True        # forward: a Bottleneck's first conv, when it is a 1x1 unit of the streaming kernel (Cin, Cout <= 128), makes its
                                       # input from the previous Bottleneck's raw output - no BatchNorm launch in front of it (round 5)
    fuse_bn_dgrad: bool = True         # backward: the input gradient of such a 1x1 unit makes dz itself - no BatchNorm-backward apply launch (round 5)
    fuse_bn_reduce_stream: bool = True # backward: the input gradient of a Bottleneck's 1x1 conv on the streaming kernel (128 < Cout <= 256: the 40x40 level) also
                                       # takes the BatchNorm-backward sums of the unit below - no reduce launch for that unit (round 5; sums' fp32 order differs)
    fuse_loss_decode: bool = True      # TrainStep: loss gradient and the head's decode backward in one pass (round 5; not with the L1 branch)
    # ---- captured step (ep24.train.TrainStep) ----
    parallel_forward: bool = True      # level-0 head chain on a second forward lane
    forward_lanes: int = 2             # 2: one more stream beside the main lane; 3: head level 1 on a stream of its own
    head1_side: bool = True            # with two lanes: head level 1 follows head level 0 on the second lane (round 5; False: on the main lane)
    cost_on_lanes: bool = True         # SimOTA's pw / cost rows of head levels 0, 1 on the forward lane that produced them (round 5)
    tail_cuts: bool = True             # last 4 % of backward: a graph cut in front of an input gradient that follows its layer's weight gradient (round 5)
    bwd_cuts: Optional[Tuple[float, ...]] = None   # fractions of the backward list where its graph segments are cut (None: default)
    chunked_update: bool = True        # the optimizer update in pieces on the weight-gradient lane, each as soon as its gradients are complete

    LAYOUT_FIELDS: ClassVar[tuple] = ("merge_csp", "merge_csp_shortcut", "merge_head")     # these decide the flat parameter layout

    def with_(self, **kw):
        return replace(self, **kw)

    def layout(self):
        return tuple(getattr(self, k) for k in self.LAYOUT_FIELDS)

    @classmethod
    def parse(cls, text):
        """"key=value,key=value" -> PlanOptions (booleans as 0/1; bwd_cuts as u<N> for N uniform segments or a:b:c fractions)."""
        kw = {}
        types = {f.name: f.type for f in fields(cls)}
        for item in filter(None, (text or "").split(",")):
            k, _, v = item.partition("=")
            k = k.strip()
            if k not in types:
                raise ValueError("unknown plan option %r (known: %s)" % (k, ", ".join(sorted(types))))
            if k == "bwd_cuts":
                kw[k] = (tuple((i + 1) / int(v[1:]) for i in range(int(v[1:]) - 1)) if v[0] == "u"
                         else tuple(float(x) for x in v.split(":")))
            elif k in ("forward_lanes", "conv_kernel_opts", "wgrad_group_steps"):
                kw[k] = int(v)
            else:
                kw[k] = v.strip() not in ("0", "false", "False", "")
        return cls(**kw)


DEFAULT = PlanOptions()


def set_options(model, options):
    """Attach ``options`` to ``model`` (the root module a plan is built for).  The merge options decide the flat parameter
    layout, so they must be set before the model's parameters move into flat buffers (its first forward / TrainStep); the other
    options apply to plans built afterwards (plans already built keep theirs)."""
    home = model.__dict__.get("_ep24_home")
    if home is not None and home.options.layout() != options.layout():
        from ._lib import Ep24Error
        raise Ep24Error("ep24: the merge options must be set before the model first runs (its parameters already live in flat "
                        "buffers laid out for %r)" % (home.options,))
    model.__dict__["_ep24_options"] = options
    return model


def get_options(model):
    return model.__dict__.get("_ep24_options", DEFAULT)
