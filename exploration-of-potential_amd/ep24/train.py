"""One training step of the 24p detector as replayable hipGraphs + a two-stream backward, plus the optimizer mirror.

Reference step body (yolox_24p/train_24p.py:80-111): zero_grad -> model(images, train=True) ->
Loss_Function.forward -> backward -> SGD.step -> .item() (+53 TensorBoard D2H syncs).  Here the whole body -
gradient clearing, weight packing, forward list, SimOTA + loss + loss gradient, backward list, (gradient
all-reduce), fused SGD - is a fixed sequence of launches on static buffers, captured once and replayed;
the host only copies the next batch into the static input buffers and reads the loss when it wants to.
"""
import torch

from . import _lib, loss as eloss
from .engine import param_home


class SGD(torch.optim.Optimizer):
    """``torch.optim.SGD(params, lr, momentum, nesterov=True)`` semantics (yolox_24p/exp/yolox_base.py:120-124,
    no weight decay) as ONE fused kernel over the model's flat parameter / gradient / momentum buffers."""

    def __init__(self, params, lr, momentum=0.9, nesterov=True, model=None):
        if not nesterov or model is None:
            raise NotImplementedError("ep24.SGD implements the reference's nesterov SGD over an ep24 model")
        super().__init__(list(params), dict(lr=lr, momentum=momentum, nesterov=True))
        self.model = model

    def zero_grad(self, set_to_none=False):
        home = param_home(self.model)
        home.bind_grads()
        home.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        g = self.param_groups[0]
        param_home(self.model).sgd(g["lr"], g["momentum"], grad_scale)

    def state_dict(self):
        home = param_home(self.model)
        sd = super().state_dict()
        # momentum buffers in torch.optim.SGD's layout so reference-style checkpoints round-trip
        order = [p for grp in self.param_groups for p in grp["params"]]
        sd["state"] = {i: {"momentum_buffer": home.views[p][2].detach().clone()} for i, p in enumerate(order)}
        return sd

    def load_state_dict(self, sd):
        home = param_home(self.model)
        order = [p for grp in self.param_groups for p in grp["params"]]
        for i, st in sd.get("state", {}).items():
            if st.get("momentum_buffer") is not None:
                home.views[order[int(i)]][2].copy_(st["momentum_buffer"])
                home.first_flag.zero_()
        for grp, new in zip(self.param_groups, sd["param_groups"]):
            grp["lr"], grp["momentum"] = new["lr"], new["momentum"]


class TrainStep:
    """Captured training step.  ``step(images, labels)`` returns the device tensor ``result[64]`` of
    ``ep24_loss_finalize`` (loss at [0]); nothing synchronises unless the caller reads it.

    ``reducer`` (ep24.dp.GradReducer or None) averages the flat gradient across ranks between backward and
    SGD: the backward list is cut where a gradient bucket is complete and the bucket's all-reduce is issued on
    the communication stream while the rest of backward still runs (no collective inside a graph).
    """

    def __init__(self, model, loss_fn, lr, momentum=0.9, batch=None, size=640, reducer=None, use_graph=True,
                 graph_backward=True, ema=None, use_l1=False):
        _lib.require_gpu()
        self.model, self.loss_fn = model, loss_fn
        self.eng = model.engine(batch, size)
        self.home = self.eng.home
        self.lr, self.momentum = lr, momentum
        self.reducer = reducer
        # the schedule / EMA / L1 pieces the reference carries for long runs.  lr, momentum, 1/world and
        # the EMA decay of the step sit in a device block the captured update reads, so none of them re-captures a graph
        self.ema = ema                                   # ep24.ema.ModelEMA or None: fused into the SGD launch
        self.ema_home = ema.homes(model)[1] if ema is not None else None
        self.hp = torch.zeros(8, dtype=torch.float32, device=self.eng.dev)
        self._hp_dirty = True
        self.use_l1 = False
        eng = self.eng
        self.ws = loss_fn.workspace(eng.B, eng.A, eng.dev)
        self.state = loss_fn._state
        self.xs = torch.cat(eng.x_shifts, 1)[0].contiguous()
        self.ys = torch.cat(eng.y_shifts, 1)[0].contiguous()
        self.st = torch.cat(eng.exp_strides, 1)[0].contiguous()
        self.labels = torch.zeros(eng.B, 50, 51, dtype=torch.float32, device=eng.dev)
        self.graphs = None
        self._side = None
        self.use_graph = use_graph
        # backward runs on two lanes (main: BN backward / dgrad / glue, side: weight gradients).  graph_backward=True
        # replays each lane as a chain of captured segments (no launch gaps: 28.7 ms/step at YOLOX-l / B=20);
        # False launches the same lanes from the host with per-layer events (29.4 ms)
        self.graph_backward = graph_backward
        opts = self.eng.options                          # per-model plan options (ep24.options.PlanOptions)
        self.parallel_forward = graph_backward and opts.parallel_forward
        self.forward_lanes = int(opts.forward_lanes)
        self.chunked_update = bool(opts.chunked_update)
        self.world = 1 if reducer is None else reducer.world
        if reducer is not None:
            reducer.attach(self.home, eng)
        if use_l1:
            self.set_use_l1(True)

    def set_lr(self, lr):
        """Learning rate of the following steps (``LRScheduler.update_lr(iter)``); no graph is re-captured."""
        if lr != self.lr:
            self.lr = lr
            self._hp_dirty = True

    def set_use_l1(self, on):
        """The reference's ``use_l1`` switch on head and loss (yolo_head_24p.py:128, losses.py:163; its yolox ancestor
        turns it on for the last no-aug epochs).  The L1 buffers become launch arguments, so the graphs are captured
        again on the next step."""
        on = bool(on)
        if on == self.use_l1:
            return
        self.use_l1 = on
        self.eng.set_use_l1(on)
        self.loss_fn.use_l1 = on
        if getattr(self.model, "head", None) is not None:
            self.model.head.use_l1 = on
        self.graphs = None

    def _push_hparams(self):
        d, omd = self.ema.next_decay() if self.ema is not None else (0.0, 0.0)
        if self._hp_dirty or self.ema is not None:
            _lib.call("set_hparams", _lib.ptr(self.hp), float(self.lr), float(self.momentum), 1.0 / self.world, float(d),
                      float(omd), _lib.stream_ptr())
            self._hp_dirty = False

    # the three phases, each a pure launch sequence on the current stream
    # The forward pass as two lanes too: from the point where the 80x80 PAFPN output is complete, the level-0 head chain
    # (the big kernels) runs on the second lane next to the rest of the neck and the level-1/2 heads (40x40 / 20x20 kernels
    # that leave most of the chip idle).  Three graphs, two events, no cross-stream edge inside a graph.
    def _fwd_split(self):
        eng = self.eng
        if getattr(eng, "fwd_head0", None) is None or not self.parallel_forward:
            return None
        lo, hi = eng.fwd_head0
        fork = eng.fwd_fork
        # Head level 1 off the main lane from the point its input (pan_out1) is complete: on a third stream (forward_lanes = 3) or,
        # the default since round 5, behind head level 0 on the second lane.  Timed events in a plain run (profiles/r05_fwd_lanes.txt)
        # showed the main lane - rest of the neck + head levels 1, 2 - finishing 0.84 ms AFTER the level-0 head, the opposite of what
        # the profiler's trace had suggested for two rounds.
        if (self.forward_lanes >= 3 or self.eng.options.head1_side) and getattr(eng, "fwd_head1", None) is not None:
            f1, (lo1, hi1) = eng.fwd_fork1, eng.fwd_head1
            return eng.fwd[:fork], eng.fwd[fork:f1], eng.fwd[lo:hi], eng.fwd[f1:lo] + eng.fwd[hi1:], eng.fwd[lo1:hi1]
        return eng.fwd[:fork], eng.fwd[fork:lo] + eng.fwd[hi:], eng.fwd[lo:hi]

    def _phase_pre_side(self):
        """The label-only part of SimOTA (candidate masks) on the second lane beside the start of the forward pass: that lane is idle
        until the level-0 head forks.  (The candidate kernel used to deviate next to MFMA kernels: a v_pk_mul_f32 with swapped
        operand halves emitted by the SLP vectoriser, rooted in round 3 - tools/hazard_probe.hip; the library
        is built without it, tests/test_gpu_hazard.py.)"""
        eloss.assign_candidates(self.ws, self.labels, self.xs, self.ys, self.st)

    def _phase_pre_backward(self):
        """What only backward needs, on the second lane while the main lane runs the loss (a chain of small kernels that leave most
        of the chip idle): the gradient clear and the transposed weight copy of the input-gradient kernels.  At the start of the
        step the same two kernels shared the memory system with the stem (profiles/r03_stream_gaps.txt)."""
        self.home.zero_grad()
        self.home.pack(2)

    def _phase_forward_head(self):
        eng = self.eng
        if not torch.cuda.is_current_stream_capturing():
            eng.draw_dropout()
        eng.zero_step_buffers()
        # the packed forward copy of the weights is kept current by the fused update (round 5: csrc/elementwise.hip sgd_kernel;
        # step() packs everything once when something else has written parameters); only the segments the update cannot write
        # (padded-Cin convs: the Focus stem) are packed here - a few KB instead of 217 MB of masters at the head of the main lane
        self.home.pack_rest_forward()
        eng.run_lane(self._fwd_split()[0])

    def _loss_grad(self, origin):
        """The loss gradient.  Default (round 5): fused with the head's decode backward - the rows the prediction convs' backward reads
        are written directly (ep24_loss_grad_decode) and the plan's head_decode_bwd entries are skipped; with the L1 branch, or
        PlanOptions(fuse_loss_decode=False), the dense fp32 gradient + the three head_decode_bwd launches."""
        eng = self.eng
        fused = bool(eng.options.fuse_loss_decode) and not self.use_l1 and len(eng.head_grads) > 0
        eng.skip_decode_bwd = fused
        if fused:
            eloss.loss_grad_decode(self.ws, eng.outputs, self.labels, eng.decode_levels())
        else:
            eloss.loss_grad(self.ws, eng.outputs, self.labels, None, origin, (self.xs, self.ys, self.st))
        eng.dyn["dout"] = self.ws.dout.data_ptr()
        eng.dyn["d_origin"] = self.ws.d_origin.data_ptr() if self.use_l1 else None

    def _phase_loss(self):
        eng = self.eng
        origin = eng.origin if self.use_l1 else None
        eloss.assign_and_reduce(self.ws, eng.outputs, self.labels, self.xs, self.ys, self.st, self.state, origin, candidates_done=True,
                                cost_done=getattr(self, "_cost_done", ()))
        self._loss_grad(origin)

    def _phase_forward(self):
        eng = self.eng
        self.home.zero_grad()
        eng.forward()
        origin = eng.origin if self.use_l1 else None
        eloss.assign_and_reduce(self.ws, eng.outputs, self.labels, self.xs, self.ys, self.st, self.state, origin)
        self._loss_grad(origin)

    def _phase_backward(self, lo, hi):
        eng = self.eng
        eng._run(eng.bwd[lo:hi])

    def _phase_update(self, lo=0, hi=None, last=True):
        self.home.sgd_hp(self.hp, self.ema_home, lo, hi, last)

    def _early_update_cut(self, segs):
        """Where the flat parameter buffer is cut for the two-part update.  The buffer is in execution order, so backward completes it
        from the tail; when the main lane has finished its last segment, everything except what the weight-gradient lane's LAST
        segment still writes (the stem's weight gradient and the final reduce launch: the head of the buffer) is complete once that
        lane has finished the segment before.  Returns (element offset, index of that segment) or None."""
        eng = self.eng
        if self.reducer is not None or len(segs) < 3:
            return None                                       # with a reducer the update follows the last all-reduce
        k = len(segs) - 2
        lo_late = segs[k + 1][0]
        late = [w for i in range(lo_late, len(eng.bwd)) if eng.bwd[i][0].startswith("side:") for w in eng.bwd_writes[i]]
        if not late:
            return None
        cut = max(off + cnt for off, cnt in late)
        cut = (cut + 3) // 4 * 4
        if cut <= 0 or cut >= self.home.numel:
            return None
        return cut, k

    def _update_chunks(self, segs, k_early):
        """The update in pieces on the weight-gradient lane (that lane idles behind the main one for half the backward pass, and the
        update is pure HBM traffic - 1.5 GB per step with the EMA copy - that used to fall into the step's HBM-bound tail next to the
        stem's weight gradient).  Returns ({segment index: (lo, hi)}, lo of the lowest piece): once the lane has run segment i, the
        elements [lo, hi) of the flat buffers can be updated - every gradient in the range was written by an entry of a segment <= i
        (both lanes: the lane starts a segment when the main lane has finished it) or is written by nobody (alignment padding).
        Segments from ``k_early`` on stay with the two-part update at the end of the step."""
        eng, n = self.eng, self.home.numel
        min_chunk = max(n // 16, 4096)                            # at most 16 pieces: a launch per piece
        wr = [[w for i in range(lo, hi) for w in eng.bwd_writes[i]] for lo, hi in segs]
        done, pos = [], 0
        for off, cnt in sorted(w for ws in wr for w in ws):       # regions nobody writes are complete from the start
            if off > pos:
                done.append((pos, off - pos))
            pos = max(pos, off + cnt)
        if pos < n:
            done.append((pos, n - pos))
        chunks, prev = {}, n
        for k in range(k_early):
            done += wr[k]
            x = n
            for off, cnt in sorted(done, key=lambda w: -(w[0] + w[1])):     # lowest x with [x, n) covered
                if off + cnt < x:
                    break
                x = min(x, off)
            x = (x + 3) // 4 * 4
            if prev - x >= min_chunk:
                chunks[k] = (x, prev)
                prev = x
        return chunks, prev

    def _segments(self):
        """Cut points of the backward list.  The weight-gradient lane runs one segment behind the main lane, so the last
        segments are short (what is left of the lane after the main lane has finished is exposed); with a reducer its
        bucket boundaries are cut points too."""
        n = len(self.eng.bwd)
        fr = (0.12, 0.24, 0.36, 0.48, 0.58, 0.68, 0.76, 0.83, 0.89, 0.93, 0.96, 0.98, 0.99, 0.995)
        if self.eng.options.bwd_cuts is not None:
            fr = self.eng.options.bwd_cuts
        cuts = {0, n} | {int(n * f) for f in fr}
        ready = {}
        if self.reducer is not None:
            rc = self.reducer.cuts(self.eng)
            cuts |= set(rc)
            ready = {c: i for i, c in enumerate(rc[1:])}          # cut index -> reducer segment that ends there
        if getattr(self.eng, "bwd_tail_cut", None):            # the reduce launch in front of the last unit closes a segment:
            cuts.add(self.eng.bwd_tail_cut)                    # the side lane runs it while the main lane does that unit's BatchNorm
        if self.eng.options.tail_cuts:
            # At the END of backward what the side lane still holds is exposed.  A weight gradient waits for the main lane to finish the
            # SEGMENT it is in, i.e. also for the input gradient of its own layer that follows it in the list (dark2's stride-2 conv:
            # 0.39 ms, during which the side lane sat idle and the stem's weight gradient then queued behind this one).  In the last
            # entries a cut goes in front of every such input gradient: the weight gradient starts when its dy is there.
            bwd = self.eng.bwd
            for i in range(max(int(n * 0.96), 2), n):
                if bwd[i][0].startswith("conv_dgrad") and bwd[i - 1][0] == "@side_record" and bwd[i - 2][0].startswith("side:"):
                    cuts.add(i)
        join = self.eng.bwd_join
        if join is not None:                                   # head levels 1, 2 run on the side lane up to here
            cuts.add(join)
            # (Cutting the trunk's first graph short - 2 / 8 launches, or a chain of 2, 4, 8, 16 - so that its submission would not
            # delay the first kernel behind the join was tried: the main lane's wait there stayed 0.12 - 0.17 ms in two of three
            # traces and the step time did not move, 923 against 925 images/s same box.  The wait is the host waking up, not the
            # size of what it submits.)
            if self.eng.bwd_par_end:
                cuts.add(self.eng.bwd_par_end)
        cuts = sorted(cuts)
        return list(zip(cuts[:-1], cuts[1:])), ready

    def _capture(self):
        eng = self.eng
        # warm-up outside capture (lazy code-object loads); it must not count as a training step, so the
        # stateful pieces it touches (BN running statistics, the loss's "last loss" weights) are restored
        keep = [b.clone() for b in self.model.buffers()] + [self.state.clone()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._phase_forward()
            self._phase_backward(0, len(eng.bwd))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, k in zip(list(self.model.buffers()) + [self.state], keep):
                b.copy_(k)
        pool = [None]

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread polls the events of collectives in flight (the broadcast of
            # ep24.dp.GradReducer.attach, the previous step's buckets); in the default global mode such a query from ANY
            # thread while this one captures is an error that kills the rank
            with torch.cuda.graph(g, pool=pool[0], capture_error_mode="thread_local"):
                fn()
            pool[0] = g.pool()
            return g

        split = self._fwd_split()
        if split is None:
            self.g_fwd = capture(self._phase_forward)
        else:
            # SimOTA's pw / cost rows of a head level on the lane that produced the level (round 5): the second lane has 0.3 ms to spare
            # at the end of the forward pass and the rows of levels 0 and 1 are 95 % of the loss path's longest kernel
            self._cost_done = []
            lane_cost = {}
            if eng.options.cost_on_lanes and len(eng.head_grads) == 3:
                a1 = eng.head_grads[0][0]
                lane_cost[2] = (0, a1)                                        # split[2]: head level 0
                if len(split) == 5 and self.forward_lanes < 3:                # (a third stream is not ordered behind the candidate masks)
                    lane_cost[4] = (a1, a1 + eng.head_grads[1][0])            # split[4]: head level 1, behind level 0 on the same lane
                self._cost_done = sorted(lane_cost.values())

            def lane_fn(i, l):
                eng.run_lane(l)
                if i in lane_cost:
                    eloss.assign_cost_range(self.ws, eng.outputs, self.labels, *lane_cost[i])
            lanes = [capture(lambda i=i, l=l: lane_fn(i, l)) for i, l in enumerate(split) if i >= 1]
            self.g_pre = capture(self._phase_pre_side)
            self.g_pre_bwd = capture(self._phase_pre_backward)
            self.g_fwd = (capture(self._phase_forward_head),) + tuple(lanes) + (capture(self._phase_loss),)
            if self._side is None:
                self._side = torch.cuda.Stream(device=eng.dev)
            if len(split) == 5 and getattr(self, "_side2", None) is None:
                self._side2 = torch.cuda.Stream(device=eng.dev) if self.forward_lanes >= 3 else self._side
        self.g_upd = capture(self._phase_update)
        self.g_upd_early = None
        self._cuts = [0, len(eng.bwd)] if self.reducer is None else self.reducer.cuts(eng)
        self.g_bwd = None
        if self.graph_backward:
            # two lanes, each a chain of graphs: the main lane (BN backward, dgrad, glue) and the weight-gradient lane,
            # which starts a segment when the main lane has finished it - one event per segment, none inside a graph
            segs, ready = self._segments()
            self.g_bwd = []
            early = self._early_update_cut(segs)
            chunks, upd_hi = self._update_chunks(segs, early[1]) if early is not None and self.chunked_update else ({}, None)
            self.update_chunks = dict(chunks)
            for si, (lo, hi) in enumerate(segs):
                main, side = eng.lane_lists(lo, hi)
                gm = capture(lambda: eng.run_lane(main)) if main else None
                piece = chunks.get(si)

                def side_work(side=side, piece=piece):
                    eng.run_lane(side)
                    if piece is not None:                     # the parameters whose gradients this segment has completed
                        self._phase_update(piece[0], piece[1], False)

                gs = capture(side_work) if side or piece is not None else None
                par = eng.bwd_par_end is not None and hi <= eng.bwd_par_end      # a segment that runs on the side lane only
                self.g_bwd.append((gm, gs, ready.get(hi), lo == eng.bwd_join, par))
            if self._side is None:
                self._side = torch.cuda.Stream(device=eng.dev)
            # Two-part update: when the main lane is through, the weight-gradient lane still has its last segment to run (the
            # stem's 2 M-pixel weight gradient and the final reduce, ~0.25 ms during which the main lane used to wait, then 0.18 ms
            # of SGD).  The parameters above the cut are updated in that window, the few below it afterwards.
            if early is not None:
                cut, k = early
                self.g_upd_early = (capture(lambda: self._phase_update(cut, upd_hi, False)), capture(lambda: self._phase_update(0, cut, True)), k)
                if chunks:                                    # what is left if the early part does not run
                    self.g_upd = capture(lambda: self._phase_update(0, upd_hi, True))
        self.graphs = True

    def step(self, images=None, labels=None):
        eng = self.eng
        if images is not None:
            eng.images.copy_(images, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        if self.graphs is None and self.use_graph:
            self._capture()
        if self.use_graph and not self.home.wf_current:
            # first step, a loaded checkpoint, parameters written from outside: the whole forward copy from the masters, once
            self.home.pack(1)
            self.home.wf_current = True
        self._push_hparams()
        eng.draw_dropout()                               # DenseNet backbone: this step's Dropout2d factors (device RNG)
        if not self.use_graph:
            self._phase_forward()
            self._phase_backward(0, len(eng.bwd))
            if self.reducer is not None:
                self.reducer.reduce_all()
            self._phase_update()
            return self.ws.result
        if isinstance(self.g_fwd, tuple):
            # the second lane starts the step too: candidate masks (the loss waits for this lane's later work anyway)
            main, side = torch.cuda.current_stream(), self._side
            ev0 = torch.cuda.Event()
            ev0.record(main)
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                self.g_pre.replay()
        if isinstance(self.g_fwd, tuple) and len(self.g_fwd) == 6:
            g1, g_main_a, g_side, g_main_b, g_side2, g_loss = self.g_fwd
            main, side, side2 = torch.cuda.current_stream(), self._side, self._side2
            probe = getattr(self, "probe", None)      # tools/fwd_lanes_probe.py: timed events at the lane boundaries (outside a profiler)

            def mark(name, stream):
                if probe is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(stream)
                    probe[name] = e
            mark("start", main)
            g1.replay()
            mark("fork", main)
            ev = torch.cuda.Event()
            ev.record(main)
            g_main_a.replay()                         # the main lane's graphs first (see the backward loop)
            eva = torch.cuda.Event()
            eva.record(main)
            g_main_b.replay()
            mark("main_end", main)
            side.wait_event(ev)
            mark("side_begin", side)
            with torch.cuda.stream(side):
                g_side.replay()                       # head level 0
            side2.wait_event(eva)                     # head level 1 (side2 is the same stream unless forward_lanes = 3)
            with torch.cuda.stream(side2):
                g_side2.replay()
            mark("side_end", side2)
            main.wait_stream(side)
            main.wait_stream(side2)
            mark("loss_begin", main)
            g_loss.replay()
            mark("loss_end", main)
            with torch.cuda.stream(side):
                self.g_pre_bwd.replay()
            ev_pb = torch.cuda.Event()
            ev_pb.record(side)
            main.wait_event(ev_pb)
        elif isinstance(self.g_fwd, tuple):
            g1, g_main, g_side, g_loss = self.g_fwd
            main, side = torch.cuda.current_stream(), self._side
            probe = getattr(self, "probe", None)      # tools/fwd_lanes_probe.py: timed events at the lane boundaries (outside a profiler)

            def mark(name, stream):
                if probe is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(stream)
                    probe[name] = e
            mark("start", main)
            g1.replay()
            mark("fork", main)
            ev = torch.cuda.Event()
            ev.record(main)
            # the main lane's graph is enqueued first (see the backward loop).  Round 5, tools/fwd_lanes_probe.py: timed events in a
            # plain run put the side lane's start 20 us behind the fork (rocprofv3's queue interception shows it 1.2 ms late - the "558
            # us hole" of profiles/r04_stream_gaps.txt is the profiler's, not the runtime's); a plain dispatch between the graph and the
            # event delayed it by 1.3 ms, the side lane's graph first changed nothing (profiles/r05_fwd_lanes.txt)
            g_main.replay()
            side.wait_event(ev)
            mark("side_begin", side)
            with torch.cuda.stream(side):
                g_side.replay()
            mark("main_end", main)
            mark("side_end", side)
            ev2 = torch.cuda.Event()
            ev2.record(side)
            main.wait_event(ev2)
            mark("loss_begin", main)
            g_loss.replay()
            mark("loss_end", main)
            with torch.cuda.stream(side):
                self.g_pre_bwd.replay()                # beside the loss; backward starts behind both
            ev_pb = torch.cuda.Event()
            ev_pb.record(side)
            main.wait_event(ev_pb)
        else:
            self.g_fwd.replay()
        did_early = False
        if self.g_bwd is not None:
            main, side = torch.cuda.current_stream(), self._side
            # Host order matters: a graph launch into a stream that is still waiting on an event can hold the host, so
            # the main lane's graph of segment i+1 is enqueued BEFORE the side lane's graph of segment i.
            pending, par_done = None, None

            bprobe = getattr(self, "probe", None)

            def bmark(key, stream):                          # tools/fwd_lanes_probe.py: where each lane is, segment by segment
                if bprobe is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(stream)
                    bprobe.setdefault(key, []).append(e)
            bmark("bwd_main", main)

            def launch_side(p):
                gs, ready, ev, par = p
                side.wait_event(ev)
                bmark("bwd_side_begin", side)
                with torch.cuda.stream(side):
                    if gs is not None:
                        gs.replay()
                    if ready is not None:
                        self.reducer.bucket_ready(ready)      # recorded on the side lane: it has waited for the main one
                bmark("bwd_side", side)
                if par:
                    e2 = torch.cuda.Event()
                    e2.record(side)
                    return e2
                return None

            seg_done = []                                    # per segment: event on the side lane behind its graph
            early = self.g_upd_early

            def launch_side_ev(p):
                e = launch_side(p)
                if early is not None:
                    e3 = torch.cuda.Event()
                    e3.record(side)
                    seg_done.append(e3)
                return e

            for gm, gs, ready, join, par in self.g_bwd:
                if join:
                    # Only a pending segment that IS one of the side-lane head chains has to be enqueued before the wait (its event is
                    # what the main lane waits for).  Anything else (the weight gradients of the level-0 head) goes in behind the
                    # main lane's next graph as always: enqueueing it first held the host until the main lane had reached that
                    # segment's event, and the trunk's first graph arrived ~0.18 ms late (profiles/r03_stream_gaps.txt).
                    if pending is not None and pending[3]:
                        par_done = launch_side_ev(pending) or par_done
                        pending = None
                    if par_done is not None:
                        main.wait_event(par_done)          # the head levels that ran on the side lane (not its later work)
                if join:
                    bmark("bwd_join", main)
                if gm is not None:
                    gm.replay()
                ev = torch.cuda.Event()
                ev.record(main)
                bmark("bwd_main", main)
                if pending is not None:
                    par_done = launch_side_ev(pending) or par_done
                pending = (gs, ready, ev, par)
            if pending is not None:
                launch_side_ev(pending)
                if early is not None and len(seg_done) > early[2] + 1:
                    # the main lane is through and the weight-gradient lane has its last segment left: the early part of the
                    # update runs on the main lane meanwhile (it needs that lane's segments up to the one before)
                    main.wait_event(seg_done[early[2]])
                    early[0].replay()
                    did_early = True
            main.wait_stream(side)
            bmark("bwd_end", main)
        else:
            for i, (lo, hi) in enumerate(zip(self._cuts[:-1], self._cuts[1:])):
                self._phase_backward(lo, hi)
                if self.reducer is not None:
                    self.reducer.bucket_ready(i)
        if self.reducer is not None:
            self.reducer.wait()
        if did_early:
            self.g_upd_early[1].replay()                     # the parameters below the cut; finishes the step
        else:
            self.g_upd.replay()
        if getattr(self, "probe", None) is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            self.probe["end"] = e
        return self.ws.result
