"""Data parallelism: one process per GPU, gradient SUM all-reduce over RCCL (xGMI), overlapped with backward.

The reference's 24p trainer is single-device; its only description of data parallelism is the legacy Megvii
trainer (yolox_24p/core/trainer.py:163 DDP(broadcast_buffers=False), core/launch.py:118-124 one process per GPU,
NCCL).  The semantics kept here: identical replicas, per-rank BatchNorm statistics, per-rank num_fg
normalisation, gradients averaged over ranks.

MI355X form: all gradients already live in ONE flat fp32 buffer (ep24.engine.ParamHome), laid out in
execution order, so backward completes it from the tail.  The buffer is cut into a few large buckets
(4 MB, 32 MB, then 96 MB pieces from the head of the buffer: see plan()); the backward launch list
is cut where a bucket's last writer has run, and each bucket's all-reduce is issued on a communication stream
while the next backward segment (its own hipGraph) keeps the compute stream busy.  No collective sits inside
a captured graph.  Works unchanged with the gloo backend on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, bucket_bytes=32 << 20, first_bucket_bytes=4 << 20, comm_dtype=None):
        if not dist.is_initialized():
            raise RuntimeError("ep24.dp: torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_bytes = bucket_bytes
        # the head of the buffer (stem, dark2: few parameters) is what backward finishes LAST, and the all-reduce of that
        # bucket is the only one nothing overlaps: it is kept small so that the exposed tail is a fraction of a 32 MB ring pass
        self.first_bucket_bytes = min(first_bucket_bytes, bucket_bytes)
        # wire format of the buckets: None / torch.float32 = the gradients as they are (the reference's DDP), torch.bfloat16 =
        # cast to bf16, SUM-reduce, cast back (half the xGMI bytes: 108 instead of 217 MB per step for YOLOX-l; the sum of
        # `world` bf16 values is rounded once more than the fp32 sum): a constructor argument, no environment switch
        self.comm_dtype = comm_dtype if comm_dtype in (torch.bfloat16,) else None
        self._wire = None
        self.flat = None
        self.buckets = []          # (lo, hi) element ranges in readiness order
        self._cuts = None
        self._works = []
        self._comm = None
        self._events = []

    # ---- planning ---------------------------------------------------------------------------------
    def plan(self, flat, writes):
        """flat: the gradient buffer; writes: per backward op, a list of (offset, numel) ranges it writes.
        Chooses buckets and the backward indices after which each bucket is complete."""
        self.flat = flat
        n = flat.numel()
        # bucket sizes follow the time backward leaves for them: the buffer is in execution order, so its head completes last.
        # [first 4 MB | one bucket_bytes bucket | the rest in 3 x bucket_bytes pieces]: the late buckets are small (their
        # all-reduce has little backward left to hide under), the early ones - most of the parameters sit in dark5, the
        # neck and the head, whose gradients are complete before half of backward has run - are few and large (xGMI is
        # point to point: few large collectives; every collective and every extra segment cut costs ~0.08 ms of launches)
        per = max(1, self.bucket_bytes // flat.element_size())
        first = max(1, self.first_bucket_bytes // flat.element_size())
        bounds = [0]
        for size in (first, per):
            if bounds[-1] + size < n:
                bounds.append(bounds[-1] + size)
        bounds += list(range(bounds[-1] + 3 * per, n, 3 * per)) + [n]
        if self.comm_dtype is not None:                    # the cast kernels move 4 elements per lane: 16-byte aligned cuts
            bounds = sorted({b // 4 * 4 if b != n else n for b in bounds})
        ranges = list(zip(bounds[:-1], bounds[1:]))
        last = [-1] * len(ranges)
        for i, ws in enumerate(writes):
            for off, cnt in ws:
                for j, (lo, hi) in enumerate(ranges):
                    if off < hi and off + cnt > lo:
                        last[j] = max(last[j], i)
        order = sorted(range(len(ranges)), key=lambda j: last[j])
        self.buckets = [ranges[j] for j in order]
        self._ready_after = [last[j] for j in order]
        # one backward segment per distinct readiness point: a bucket is complete right after its last writer
        cuts = sorted({r + 1 for r in self._ready_after if r >= 0} | {0, len(writes)})
        self._cuts = cuts
        segs = list(zip(cuts[:-1], cuts[1:]))
        self._seg_buckets = [[] for _ in segs]
        for k, r in enumerate(self._ready_after):
            s_idx = 0 if r < 0 else next(i for i, (a, b) in enumerate(segs) if a <= r < b)
            self._seg_buckets[s_idx].append(k)
        if flat.is_cuda:
            self._comm = torch.cuda.Stream()

    def attach(self, home, eng):
        """Plan the buckets over this model's flat gradient and make the replicas identical once: rank 0's parameters,
        momentum and BatchNorm running statistics go to every rank (the legacy trainer relied on equal seeds,
        core/trainer.py:163; a rank that loaded a checkpoint would silently diverge)."""
        self.plan(home.gflat, eng.bwd_writes)
        if self.world > 1:
            for buf in (home.flat, home.mflat, home.bflat, home.first_flag):
                broadcast_parameters(buf, 0, self.group)

    def cuts(self, eng=None):
        return list(self._cuts)

    # ---- execution --------------------------------------------------------------------------------
    def _launch(self, k):
        lo, hi = self.buckets[k]
        view = self.flat[lo:hi]
        if self.flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                if self.comm_dtype is not None:
                    from ._lib import call, ptr, stream_ptr
                    if self._wire is None:
                        self._wire = torch.empty(self.flat.numel(), dtype=self.comm_dtype, device=self.flat.device)
                    wire = self._wire[lo:hi]
                    call("cast_f32_bf16", ptr(view), ptr(wire), hi - lo, stream_ptr())
                    dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
                    call("cast_bf16_f32", ptr(wire), ptr(view), hi - lo, stream_ptr())
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def bucket_ready(self, segment):
        """Call right after backward segment `segment` has been enqueued."""
        for k in self._seg_buckets[segment]:
            self._launch(k)

    def wait(self):
        """Make the compute stream (or the host, for CPU tensors) wait for every outstanding all-reduce."""
        if self.flat.is_cuda:
            torch.cuda.current_stream().wait_stream(self._comm)
        for w in self._works:
            w.wait()
        self._works = []

    def reduce_all(self):
        for s in range(len(self._seg_buckets)):
            self.bucket_ready(s)
        self.wait()


def broadcast_parameters(flat, src=0, group=None):
    """One-time replica synchronisation (the legacy trainer relies on identical seeds; this makes it explicit)."""
    dist.broadcast(flat, src=src, group=group)
