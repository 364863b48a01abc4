#!/usr/bin/env python3
"""Headline benchmark: training images/sec of YOLOX-l-24p, 640x640, bf16, batch 20 per GPU (BASELINE.json config 2;
--gpus N shards a global batch of 20*N over N ranks = config 3).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = zero grads -> forward (131 MFMA convs + BN/SiLU) -> SimOTA + 24-circle loss -> backward -> (RCCL gradient
all-reduce) -> fused SGD, on synthetic images / labels already resident in HBM (SURVEY.md 8d generator).
Rank 0 prints ONE JSON line with the whole-job images/s, the roofline of the dominant kernel (HIP events around
each of its launches in an instrumented eager step of the same workload) and, at N=1, a CPU baseline: the
oracle's fp32 train step timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "exploration-of-potential_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

TRAIN_GFLOP_PER_IMAGE = 466.2       # 3 x 2 x 77.694 GMAC, BASELINE.md section 3
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md


def conv_shape(name, args):
    """(B, H, W, Cin, Cout, k, s) of a conv launch entry (the _ex forms carry kernel_opts as one more trailing argument)."""
    if name.endswith("_ex"):
        args = args[:-1]
    if name.startswith("conv_dgrad_bnr"):               # (dy, ld, wt, dx, ld, B, H, W, Cin, Cout_k, ksize, ...): stride 1
        return tuple(args[5:11]) + (1,)
    return args[6:13] if name.startswith("conv_dgrad") else args[-7:]


def group_rows(args):
    """The (B, H, W, Cin, Cout, k, s) of the problems of a grouped weight-gradient launch: its first argument is the address of a HOST
    int64 table [n][17] (include/ep24.h, ep24_conv_wgrad_group_bf16), n the second."""
    import ctypes
    ptr, n = args[0], args[1]
    t = (ctypes.c_int64 * (17 * n)).from_address(ptr)
    return [tuple(int(t[17 * i + j]) for j in range(9, 16)) for i in range(n)]


def conv_flops(args):
    B, H, W, Cin, Cout, k, s = args
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    return 2.0 * B * OH * OW * Cin * Cout * k * k


def conv_bytes(args):
    """Algorithmic HBM bytes of one conv launch: every operand element once (bf16 activations / packed weights)."""
    B, H, W, Cin, Cout, k, s = args
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    return 2.0 * (B * H * W * Cin + B * OH * OW * Cout + Cout * k * k * Cin)


def pmc_traffic(prefix):
    """Average HBM-side bytes per launch of the kernels whose name starts with `prefix`, from the committed PMC passes
    (profiles/*_pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied there)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    n = b = 0
    for name, v in d["kernels"].items():
        if name.startswith(prefix):
            n += v["launches"]
            b += v["traffic_bytes"] * v["launches"]
    return (round(b / n) if n else None), "%s @ %s" % (os.path.basename(files[-1]), d.get("git_head", "unstamped"))


# device-kernel name prefixes behind each member of the dominant family in a rocprofv3 kernel list
REPLAY_PREFIX = {"conv_ring_kernel": "conv_ring_kernel", "conv_ring_generic_kernel": "conv_ring_generic_kernel", "conv_patch_kernel": "conv_patch_kernel", "igemm_dma_kernel": "igemm_dma_",
                 "conv_wreg_kernel": "conv_wreg_kernel"}


def replayed_ms_per_step(prefix):
    """ms per step the kernels starting with `prefix` take in the graph-replayed, two-lane run: from the committed rocprofv3
    --kernel-trace --stats summary (profiles/*_kernel_meta.json, stamped with the commit it was taken at)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_meta.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    ms = sum(v["ms_per_step"] for k, v in d["kernels"].items() if k.startswith(prefix))
    return (ms or None), "%s @ %s" % (os.path.basename(files[-1]), d.get("git_head", "unstamped"))


def instrumented_step(ts):
    """Eager pass over the same launch lists with an event pair round every conv launch."""
    from ep24 import _lib, loss as eloss
    eng = ts.eng
    fn = _lib.lib().fn
    rec = []
    all_rec = [] if os.environ.get("EP24_LAYER_TABLE") else None

    def run(lst):
        s = _lib.stream_ptr()
        for name, args in lst:
            if name[0] == "@":                      # stream-ordering entries: the instrumented pass is serial
                continue
            if name.startswith("side:"):
                name = name[5:]
            a = [x.get() if hasattr(x, "get") else x for x in args]
            timed = name.startswith("conv_")
            if all_rec is not None and not timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn["ep24_" + name](*a, s)
                e1.record()
                tail = a[6:11] if name == "conv1x1_dgrad_bnr_bf16" else a[-5:]        # (that entry's shape B, H, W, Cin, Cout sits in front of the unit below's arguments)
                all_rec.append((name, tuple(v for v in tail if isinstance(v, int) and abs(v) < (1 << 24)), e0, e1))
                continue
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn["ep24_" + name](*a, s)
            assert rc == 0, (name, _lib.lib().last_error())
            if timed:
                e1.record()
                if name == "conv_wgrad_group_bf16":
                    shapes = group_rows(a)
                    rec.append((name, sum(conv_flops(sh) for sh in shapes), e0, e1, sum(conv_bytes(sh) for sh in shapes)))
                else:
                    rec.append((name, conv_flops(conv_shape(name, args)), e0, e1, conv_bytes(conv_shape(name, args))))

    ts.home.zero_grad()
    eng.zero_step_buffers()
    ts.home.pack()
    run(eng.fwd)
    eloss.assign_and_reduce(ts.ws, eng.outputs, ts.labels, ts.xs, ts.ys, ts.st, ts.state)
    eloss.loss_grad(ts.ws, eng.outputs, ts.labels)
    eng.dyn["dout"] = ts.ws.dout.data_ptr()
    run(eng.bwd)
    torch.cuda.synchronize()
    if os.environ.get("EP24_LAYER_TABLE"):
        rows = {}
        for (name, fl, e0, e1, _by), (_, args) in zip(rec, [x for x in list(eng.fwd) + list(eng.bwd) if x[0].replace("side:", "").startswith("conv_")]):
            if name == "conv_wgrad_group_bf16":          # a grouped launch: its time goes to its problems by their share of the FLOPs
                shapes = group_rows(args)
                for sh in shapes:
                    r = rows.setdefault(("conv_wgrad_grouped",) + tuple(sh), [0, 0.0, 0.0])
                    r[0] += 1
                    r[1] += conv_flops(sh)
                    r[2] += e0.elapsed_time(e1) * conv_flops(sh) / fl
                continue
            key = (name,) + tuple(conv_shape(name, args))
            r = rows.setdefault(key, [0, 0.0, 0.0])
            r[0] += 1
            r[1] += fl
            r[2] += e0.elapsed_time(e1)
        with open(os.environ["EP24_LAYER_TABLE"], "w") as fh:
            fh.write("kernel B H W Cin Cout k s : launches  ms_total  TFLOP/s\n")
            for key, (n, fl, ms) in sorted(rows.items(), key=lambda kv: -kv[1][2]):
                fh.write("%-18s %s : %3d %8.3f %8.1f\n" % (key[0], " ".join("%4d" % v for v in key[1:]), n, ms, fl / ms / 1e9))
            other = {}
            for name, key, e0, e1 in all_rec:
                r = other.setdefault((name,) + key, [0, 0.0])
                r[0] += 1
                r[1] += e0.elapsed_time(e1)
            fh.write("\nother kernels (name, trailing int args) : launches ms_total\n")
            for key, (n, ms) in sorted(other.items(), key=lambda kv: -kv[1][1]):
                fh.write("%-22s %-40s : %3d %8.3f\n" % (key[0], " ".join(str(v) for v in key[1:]), n, ms))
    fam = {}
    def kernel_of(name, args):
        """Which device kernel a conv launch runs: the library's own dispatch rule (ep24_conv_kernel_for), not a copy of it."""
        if name.startswith("conv_wgrad"):
            return "wgrad_kernel"
        B, H, W, Cin, Cout, k, s = conv_shape(name, args)
        fwd = name.startswith("conv_fwd")
        if name.endswith("_ex") and (args[-1] & 1) and k == 3 and s == 1:
            return "igemm_dma_kernel"                      # kernel_opts bit 0: the tiled kernel was asked for
        kid = fn["ep24_conv_kernel_for"](0 if fwd else 1, B, H, W, Cin, Cout, k, s, int(bool(fwd and args[5] != 0)), int(bool(fwd and args[8] is not None)))
        assert kid >= 0, (name, _lib.lib().last_error())
        if kid == 3 and ((name.endswith("_ex") and (args[-1] & 8)) or name.startswith("conv_dgrad_bnr")):
            kid = 1                                        # kernel_opts bit 3 / the fused BatchNorm sums: the 8-wave halo-patch kernel
        if kid == 6 and not fwd and args[5]:
            kid = 0                                        # an ACCUMULATING input gradient of such a layer stays with the tiled kernel
        if name.endswith("_ex") and (((args[-1] & 16) and kid == 0) or (args[-1] & 64) or (args[-1] & 512)):      # bit 4: the ring without a patch; bit 6: the narrow ring; bit 9: no weights-in-registers kernel
            kid = fn["ep24_conv_kernel_for_ex"](0 if fwd else 1, B, H, W, Cin, Cout, k, s, int(bool(fwd and args[5] != 0)), int(bool(fwd and args[8] is not None)), args[-1])
        return ("igemm_dma_kernel", "conv_patch_kernel", "igemm_stream_kernel", "conv_ring_kernel", "conv_ring_generic_kernel", "conv_ring_kernel", "conv_wreg_kernel")[kid]

    convs = [x for x in list(eng.fwd) + list(eng.bwd) if x[0].replace("side:", "").startswith("conv_")]
    for (name, fl, e0, e1, by), (_, args) in zip(rec, convs):
        kern = kernel_of(name, args)
        f = fam.setdefault(kern, dict(flops=0.0, ms=0.0, launches=0, bytes=0.0))
        f["flops"] += fl
        f["bytes"] += by
        f["ms"] += e0.elapsed_time(e1)
        f["launches"] += 1
    return fam


def cpu_baseline(steps=3):
    """Oracle (CPU restatement, fp32) train step of YOLOX-l-24p at B=1, 640x640, 5 GTs on the host cores, timed at several
    thread counts (a batch-1 fp32 step does not scale to a whole socket: oversubscribing it is not a baseline); the best one
    is reported with the thread count that achieved it."""
    from ep24 import synth
    from oracle import model as om
    from oracle.loss import LossOracle
    torch.manual_seed(0)
    net = om.Net(1.0, 1.0)
    net.train()
    lf = LossOracle(80)
    params = list(net.parameters())
    bufs = [None] * len(params)
    x = synth.make_images(1, 640, seed=1)
    labels = synth.make_labels(1, 5, seed=2)

    def one():
        for p in params:
            p.grad = None
        tup = lf(net(x, train=True), labels)
        tup[0].backward()
        om.sgd_nesterov_step(params, bufs, 0.01)
        return float(tup[0])

    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    sweep = {}
    for nt in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), default_threads}):
        torch.set_num_threads(nt)
        one()                               # warm-up at this thread count
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        sweep[nt] = round(steps / (time.perf_counter() - t0), 4)
    torch.set_num_threads(default_threads)
    best = max(sweep, key=sweep.get)
    return dict(value=sweep[best], unit="images/s", cores=best, kind="port", host_cpus=ncpu,
                threads_sweep={str(k): v for k, v in sweep.items()},
                sample="oracle fp32 train step (fwd+SimOTA loss+bwd+SGD), YOLOX-l-24p, B=1, 640x640, 5 GTs, %d steps after 1 warm-up "
                       "per thread count; best of the sweep" % steps)


def self_launch(n):
    """Run this script as n ranks under torch.distributed.run (rendezvous on 127.0.0.1, a free port) and relay the output."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=20, help="per-GPU batch (BASELINE config 2: 20)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--gts", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--long-run", action="store_true", help="non-default: also run the SURVEY 8f N2 pieces inside the step (L1 "
                    "branch on, ModelEMA fused into the update, a yoloxwarmcos rate pushed every step)")
    ap.add_argument("--fisheye", action="store_true", help="non-default: BASELINE config 5 - every step first warps its uint8 source images "
                    "(and masks) with the sector warp on the GPU and letterboxes the results into the network input")
    ap.add_argument("--backbone", default="darknet", choices=["darknet", "resnet", "densenet", "vgg"], help="non-default: BASELINE config 4 (backbone swap)")
    ap.add_argument("--depthwise", action="store_true", help="non-default: the depthwise variants (DWConv in backbone, neck and head; in no BASELINE configuration)")
    ap.add_argument("--width", type=float, default=1.0, help="non-default: channel multiplier (tests run a small network through the same path)")
    ap.add_argument("--depth", type=float, default=1.0, help="non-default: depth multiplier")
    ap.add_argument("--eager-backward", action="store_true", help="launch the two backward lanes from the host instead of replaying captured segments")
    ap.add_argument("--plan", default="", help="non-default: per-model plan options for A/B runs, e.g. 'merge_csp=0,forward_lanes=3' (ep24.options.PlanOptions.parse)")
    ap.add_argument("--dp-wire", default="fp32", choices=["bf16", "fp32"], help="wire format of the gradient all-reduce at --gpus > 1: fp32 as the reference's DDP and "
                    "the trainer's default; bf16 (half the xGMI bytes) is the opt-in of train_24p.py --dp-wire bf16")
    ap.add_argument("--bucket-mb", type=int, default=32, help="all-reduce bucket size at --gpus > 1")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (the reference's launcher also spawns its own ranks,
        # yolox_24p/core/launch.py:82-96).  Decided before anything touches the GPU; the ranks are child processes of a
        # torch.distributed.run child (never an exec), rank 0's JSON line is relayed, a failing rank fails the run.
        sys.exit(self_launch(a.gpus))

    import torch.distributed as dist
    from ep24 import dp, loss as eloss, nn as enn, synth, train as etrain

    world = a.gpus
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # rehearsal hooks (one-GPU box): EP24_REHEARSE=1 puts every rank on cuda:0, uses gloo for the collective and gives
    # all ranks the same batch, so the N-rank loss must equal the 1-rank loss (tests the whole data-parallel step)
    rehearse = os.environ.get("EP24_REHEARSE") == "1"
    if rehearse:
        local = 0
    if world > 1:
        if int(os.environ.get("WORLD_SIZE", 1)) != world:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s (start it plainly, or with torch.distributed.run --nproc-per-node %d)"
                     % (world, os.environ.get("WORLD_SIZE"), world))
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # EP24_NCCL_SOLO=1 (one-GPU box): a one-rank RCCL group, so that the reducer path - buckets, communication stream,
    # the nccl backend's stream semantics next to the captured segments - runs exactly as it will on 8 GPUs
    solo = world == 1 and os.environ.get("EP24_NCCL_SOLO") == "1"
    if solo:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29613", rank=0, world_size=1, device_id=dev)

    # Which device every rank drives, gathered over the process group itself: the record that the collective backend saw
    # `world` DISTINCT GPUs (core/launch.py:118-124 of the reference pins one device per rank the same way).
    pr = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local, "device_index": dev.index, "name": pr.name,
          "uuid": str(getattr(pr, "uuid", "")), "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
          "pid": os.getpid()}
    ranks_info = [me]
    if dist.is_initialized():
        ranks_info = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_info, me)

    torch.manual_seed(0)                                     # identical replicas on every rank
    model = enn.YOLOX(enn.YOLOPAFPN(a.depth, a.width, backbone_type=a.backbone, depthwise=a.depthwise), enn.YOLOXHead(80, a.width, depthwise=a.depthwise))
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):            # init_yolo, exp/yolox_base.py:58-62
            mod.eps, mod.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    model.to(dev)
    if a.plan:
        from ep24.options import PlanOptions, set_options
        set_options(model, PlanOptions.parse(a.plan))
    lf = eloss.Loss_Function(80)
    reducer = dp.GradReducer(bucket_bytes=a.bucket_mb << 20, comm_dtype=torch.bfloat16 if a.dp_wire == "bf16" else None) if (world > 1 or solo) else None
    ema, sched = None, None
    if a.long_run:
        from ep24.ema import ModelEMA
        from ep24.schedule import LRScheduler
        ema = ModelEMA(model, 0.9998)
        sched = LRScheduler("yoloxwarmcos", 0.001, 100, 300, warmup_epochs=5, warmup_lr_start=0, no_aug_epochs=100, min_lr_ratio=0.05)
    ts = etrain.TrainStep(model, lf, lr=0.001, momentum=0.9, batch=a.batch, size=a.size, reducer=reducer,
                          use_graph=not a.no_graph, graph_backward=not a.eager_backward, ema=ema, use_l1=a.long_run)
    if sched is not None:
        plain_step, it = ts.step, [0]

        def step_with_schedule(*args):
            it[0] += 1
            ts.set_lr(sched.update_lr(it[0]))
            return plain_step(*args)
        ts.step = step_with_schedule
    # this rank's shard of the synthetic global batch (weak scaling: per-GPU work fixed)
    shard = 0 if rehearse else rank
    images = synth.make_images(a.batch, a.size, seed=1 + shard).to(dev)
    labels = synth.make_labels(a.batch, a.gts, size=a.size, seed=1000 + shard).to(dev)
    ts.eng.images.copy_(images)
    ts.labels.copy_(labels)

    if a.fisheye:
        from ep24 import input as ein, sector as esec
        dist_op = esec.Image_Distortion(str(dev))
        g8 = torch.Generator().manual_seed(77 + shard)
        src = [(torch.rand(a.size, a.size, 3, generator=g8) * 255).to(torch.uint8).to(dev) for _ in range(a.batch)]
        msk = [torch.zeros(a.size, a.size, 3, dtype=torch.uint8, device=dev) for _ in range(a.batch)]
        for mk in msk:
            mk[a.size // 4: a.size // 2, a.size // 4: a.size // 2] = 255
        thetas = [30 + 60 * i // max(a.batch - 1, 1) for i in range(a.batch)]          # Theta in [30, 90], SURVEY 8d C5
        plain_step2 = ts.step

        def step_with_warp(*args):
            warped, _masks, _boxes = dist_op.distort_batch(src, msk, thetas)      # the public batch form of sector_distort
            ein.preproc_batch(warped, (a.size, a.size), device=str(dev), out=ts.eng.images)
            return plain_step2(*args)
        ts.step = step_with_warp

    for _ in range(a.warmup):
        ts.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ts.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss = float(ts.ws.result[0])
    ips = world * a.batch * a.steps / dt

    out = None
    if rank == 0:
        fam = instrumented_step(ts)
        # The dominant kernel family: the conv forward / input-gradient gather-GEMM behind ep24_conv_fwd_bf16 / ep24_conv_dgrad_bf16.
        # Since round 2 it has two tilings - the halo-patch form (3x3 stride-1 layers with >= 200 tiles of 256 x 128: conv_ring_kernel
        # since round 4, conv_patch_kernel before and on request) and igemm_dma_kernel (everything else) - over the same 183 launches
        # that were one kernel in round 1; the line carries the
        # family and, under "members", each kernel by itself (their average launch durations are what rocprofv3 reports).
        members = [k for k in ("conv_ring_kernel", "conv_ring_generic_kernel", "conv_patch_kernel", "igemm_dma_kernel", "conv_wreg_kernel") if k in fam]
        dom = " + ".join(members)
        f = {key: sum(fam[k][key] for k in members) for key in ("flops", "ms", "launches", "bytes")}
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
        tr = [(pmc_traffic(REPLAY_PREFIX[k]), fam[k]["launches"]) for k in members]
        traffic = round(sum(t[0] * n for (t, n) in tr if t[0]) / max(sum(n for (t, n) in tr if t[0]), 1)) if any(t[0] for t, _ in tr) else None
        traffic_src = tr[0][0][1]
        # the family's time in the replayed run: a member's launches may run as several device kernels (the six stride-2 input
        # gradients of igemm_dma_kernel's share are igemm_dma_multi_kernel launches)
        reps = [replayed_ms_per_step(REPLAY_PREFIX[k]) for k in members]
        default_cfg = (a.batch, a.size, a.gts, a.backbone, a.width, a.depth) == (20, 640, 10, "darknet", 1.0, 1.0) and not (a.fisheye or a.long_run or a.no_graph or a.plan or a.depthwise)
        rep_ms = sum(r[0] for r in reps) if (default_cfg and all(r[0] for r in reps)) else None     # the committed profile is of the default workload
        rep_src = reps[0][1]
        from ep24 import _lib as _l
        ring_timeouts = int(_l.lib().fn["ep24_conv_ring_timeouts"]())
        if ring_timeouts != 0:
            sys.exit("bench.py: %d bounded waits of the ring kernels gave up (ep24_conv_ring_timeouts): results are not valid" % ring_timeouts)
        out = {
            "metric": "training images/sec, YOLOX-l 24p 640x640 bf16", "value": round(ips, 2), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "YOLOX-l-24p (%s+PAFPN+24p head) train step, %dx%d, batch %d/GPU, %d GTs/img, "
                                   "SimOTA + 24-circle GIoU loss, SGD nesterov" % ({"darknet": "CSPDarknet53", "resnet": "resnet50 backbone swap", "densenet": "densenet121 backbone swap", "vgg": "vgg19 backbone swap"}[a.backbone],
                                                                                   a.size, a.size, a.batch, a.gts),
                       "global_batch": a.batch * world, "parallelism": "dp%d" % world, "hip_graph": ("none" if a.no_graph else "fwd+loss, update; backward launched on 2 streams" if a.eager_backward
                                     else "forward and backward as two lanes of captured segments, loss, update"),
                       **({"long_run": "use_l1 + fused ModelEMA + yoloxwarmcos per step"} if a.long_run else {}),
                       **({"depthwise": "DWConv in backbone, neck and head (network_blocks.py:57-76); step_mfma_frac / roofline refer to the dense network's FLOPs and do not apply"} if a.depthwise else {}),
                       **({"fisheye": "sector warp of image + mask (Theta 30..90) and letterbox of every image inside the timed step"} if a.fisheye else {})},
            "loss": round(loss, 4),
            # bounded waits of the loader / consumer ring kernels that gave up in this process (0 unless their hand-off protocol is
            # broken: wrong numbers would follow, so the benchmark refuses to report a rate beside them)
            "ring_timeouts": ring_timeouts,
            "ranks": ranks_info,
            "distributed": ({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                             "distinct_devices": len({(r["uuid"], r["pci_bus_id"]) for r in ranks_info}),
                             "dp_wire": a.dp_wire, "bucket_mb": a.bucket_mb, "rehearsal_on_one_device": rehearse}
                            if dist.is_initialized() else None),
            "step_mfma_frac": round(ips / world * {"darknet": TRAIN_GFLOP_PER_IMAGE, "resnet": 290.7, "densenet": 388.0, "vgg": 1196.0}[a.backbone] * (a.size / 640.0) ** 2 / 1e3
                                    / MFMA_BF16_PEAK_TFLOPS, 4),      # swaps: 3 x 2 x 48.45 / 64.67 GMAC
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                         "frac_note": "achieved = the family's conv FLOPs over the SUM of its launch durations, HIP events round every launch of an "
                                      "instrumented serial pass of the same launch lists in this run",
                         "frac_replayed": (round(f["flops"] / (rep_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if rep_ms else None),
                         "frac_replayed_note": "the same FLOPs over the family's ms per step in the graph-replayed two-lane run, where kernels of the two "
                                               "lanes overlap and stretch each other (rocprofv3 --kernel-trace --stats: %s)" % rep_src,
                         "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC, %s)" % traffic_src,
                         "algorithmic_bytes_per_launch": round(f["bytes"] / f["launches"]),
                         "algorithmic_gflop_per_launch": round(f["flops"] / f["launches"] / 1e9, 3),
                         "launches_per_step": f["launches"], "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2),
                         "members": {k: {"achieved": round(fam[k]["flops"] / (fam[k]["ms"] * 1e-3) / 1e12, 2),
                                         "frac": round(fam[k]["flops"] / (fam[k]["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                         "launches_per_step": fam[k]["launches"], "avg_launch_us": round(fam[k]["ms"] * 1e3 / fam[k]["launches"], 2),
                                         "ms_per_step": round(fam[k]["ms"], 3)} for k in members},
                         "families": {k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "ms_per_step": round(v["ms"], 3),
                                          "launches": v["launches"]} for k, v in fam.items()}},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif solo:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
