"""Training entry point with the reference's flags (yolox_24p/train_24p.py:180-211):

    cd exploration-of-potential_amd/yolox_24p
    python train_24p.py -f load_train/yolox_24p_l_train.py -b 20 -l 0.01
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_24p.py -f ... -b 20 -l 0.01

The step body of the reference ``Trainer.train`` (:80-111: zero_grad, forward, Loss_Function, backward, SGD step) runs
as ONE captured hipGraph (``ep24.train.TrainStep``); ``--no-graph`` runs the same kernels through the reference-style
eager API (model(...), loss_func.forward(...), loss.backward(), optimizer.step()).  Scalars are logged from a single
packed device->host copy every ``--log-interval`` steps instead of ~54 per-step ``add_scalar`` syncs (:115-137).
Under torch.distributed.run every rank trains on its own shard (DistributedSampler in Exp.get_data_loader), rank 0's
parameters are broadcast once, and gradients are averaged over RCCL (ep24.dp).

The reference carries three more pieces that its 24p trainer never switches on; they are opt-in
here and all of them run inside the captured step: ``--sched`` follows ``exp.get_lr_scheduler`` (yoloxwarmcos,
exp/yolox_base.py:155-167) per iteration, ``--ema`` keeps a ``ModelEMA`` copy (utils/ema.py) that is saved with the
checkpoint, ``--l1`` turns ``use_l1`` on for head and loss from epoch ``exp.L1_epoch`` on (exp/yolox_base.py:38).
"""
import argparse
import os
import time

import _path  # noqa: F401
import torch

from exp import get_exp
from models import Loss_Function
from utils import save_checkpoint

try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:                                         # tensorboard is optional
    SummaryWriter = None


class Trainer:
    def __init__(self, exp, args):
        self.exp, self.args = exp, args
        self.max_epoch = exp.max_epoch
        self.L1_epoch = exp.L1_epoch
        self.start_device, self.numb_device = args.start_device, args.devices
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        local = int(os.environ.get("LOCAL_RANK", args.start_device))
        dev = torch.device(args.device)
        if dev.type != "cuda":
            # SURVEY 5 lists --device for the reference's plumbing run (BASELINE config 1).  This build is the GPU path only:
            # there is no CPU fallback behind the C ABI, and the trainer says so instead of silently running something else.
            from ep24._lib import Ep24Error
            raise Ep24Error("train_24p.py --device %s: the ep24 path runs on an MI355X only (no CPU fallback); the CPU "
                            "restatement lives in oracle/ and is test infrastructure" % args.device)
        self.device = torch.device("cuda", dev.index if dev.index is not None else local)
        self.input_size = exp.input_size
        self.file_name = os.path.join(exp.output_dir, exp.exp_name)
        self.current_step = 0                             # global step (epoch * max_iter + iter): schedule / TensorBoard axis
        self.run_steps = 0                                # steps of THIS run: what --steps limits
        os.makedirs(self.file_name, exist_ok=True)
        if not args.synthetic:
            print("note: the COCO24P loader of the reference reads hard-coded paths (datasets/coco24p.py:19-20) and is out of "
                  "scope; the synthetic source with the same label layout is used (--synthetic)")
        # Round 5: the raw uint8 source is the DEFAULT (0.98 of bench.py against 0.91 for ready-made fp32 canvases from four loader
        # processes, profiles/r04_trainer.json; one host core per rank instead of five: profiles/r05_trainer_8sets.json).
        # --fp32-batches restores the reference's form of the source; --no-prefetch implies it (raw batches are letterboxed by the
        # prefetcher's side stream).
        if args.raw_u8 and (args.no_prefetch or args.fp32_batches):
            raise SystemExit("train_24p.py: --raw-u8 batches are letterboxed by the prefetcher (drop --no-prefetch / --fp32-batches)")
        args.raw_u8 = not (args.fp32_batches or args.no_prefetch)
        self.train_loader = exp.get_data_loader(args.batch_size, raw_u8=bool(args.raw_u8), workers=args.loader_workers,
                                                 pin=None if args.loader_pin is None else bool(args.loader_pin))
        self.loss_func = Loss_Function(exp.num_classes)
        self.loss_func.draw = False

    def train(self):
        from ep24 import dp
        from ep24.train import TrainStep
        args, exp = self.args, self.exp
        torch.cuda.set_device(self.device)
        reducer = None
        if self.world > 1:
            torch.distributed.init_process_group("nccl", device_id=self.device)
            # fp32 buckets like the reference's DDP (core/trainer.py:163) unless asked: --dp-wire bf16 halves the xGMI bytes
            reducer = dp.GradReducer(comm_dtype=torch.bfloat16 if args.dp_wire == "bf16" else None)
        torch.manual_seed(0)                              # identical replicas on every rank (made explicit by the reducer's broadcast)
        model = exp.get_model()
        model.to(self.device)
        self.model = model
        self.optimizer = exp.get_optimizer(args.learn_rate)
        self.max_iter = len(self.train_loader)
        # -c / --resume / -e: the reference's parser accepts them and its trainer never reads them (train_24p.py:180-211); here
        # they load what save_ckpt wrote.  The checkpoint goes in AFTER get_model() (which re-applies the bias prior on every
        # call, exp/yolox_base.py:70-71) so the loaded predictor biases survive; --resume also restores momentum and epoch.
        self.start_epoch, ck, resumed_step = 0, None, None
        if args.ckpt:
            from utils import load_ckpt
            ck = torch.load(args.ckpt, map_location="cpu")
            load_ckpt(model, ck.get("model", ck))
            if args.resume:
                if "optimizer" in ck:
                    self.optimizer.load_state_dict(ck["optimizer"])
                self.start_epoch = int(ck.get("start_epoch", 0))
                if "global_step" in ck:                   # written by this trainer: exact also when --steps cut an epoch short
                    resumed_step = int(ck["global_step"])
        if args.start_epoch is not None:
            self.start_epoch, resumed_step = args.start_epoch, None
        # progress = epoch * max_iter + iter, as the reference's schedulers / loggers count it: a resumed run continues the
        # learning-rate schedule and the TensorBoard axis where the checkpoint stopped instead of restarting the warm-up
        self.current_step = self.resumed_step(self.start_epoch, self.max_iter) if resumed_step is None else resumed_step
        # iterations of the start epoch that the checkpointed run had already done (a run cut short by --steps inside an epoch)
        skip_iters = min(max(self.current_step - self.start_epoch * self.max_iter, 0), self.max_iter)
        # ... and the DATA position with it: the loader's first pass leaves those batches out (the epoch's first batches
        # were trained twice and its tail never ran)
        if skip_iters and hasattr(self.train_loader.sampler, "start"):
            self.train_loader.sampler.start = skip_iters * args.batch_size
        if args.throughput_json and args.steps and args.steps <= args.throughput_window:
            print("train_24p.py: --throughput-window %d clamped to --steps %d - 1" % (args.throughput_window, args.steps))
            args.throughput_window = max(args.steps - 1, 1)
        self.tblogger = SummaryWriter(self.file_name) if (SummaryWriter and self.rank == 0) else None
        self.lr_scheduler = exp.get_lr_scheduler(args.learn_rate, self.max_iter) if args.sched else None
        self.ema_model = None
        if args.ema:
            from utils import ModelEMA
            self.ema_model = ModelEMA(model, 0.9998)
            if ck is not None and args.resume and "ema_model" in ck:
                self.ema_model.ema.load_state_dict(ck["ema_model"])
                self.ema_model.updates = int(ck.get("ema_updates", 0))
        step_fn = None
        if not args.no_graph:
            step_fn = TrainStep(model, self.loss_func, lr=args.learn_rate, momentum=exp.momentum, batch=args.batch_size,
                                size=tuple(self.input_size), reducer=reducer, ema=self.ema_model)
        print("Training start... (rank %d/%d, %s)" % (self.rank, self.world, "captured step" if step_fn else "eager API"))
        done = False
        self.epoch = self.start_epoch
        self.epoch_complete = False
        tp_t0 = tp_seen = tp_dt = None                    # --throughput-json: a synchronised window over the run's last steps
        for epoch in range(self.start_epoch, self.max_epoch):
            self.epoch = epoch
            model.train()
            if hasattr(self.train_loader.sampler, "set_epoch"):
                self.train_loader.sampler.set_epoch(epoch)
            t0, seen = time.time(), 0
            if args.l1 and epoch >= self.L1_epoch and not self.loss_func.use_l1:
                model.head.use_l1 = self.loss_func.use_l1 = True
                if step_fn is not None:
                    step_fn.set_use_l1(True)
            self.epoch_complete = False
            for it, (images, labels) in enumerate(self.batches()):
                if epoch == self.start_epoch and it + skip_iters >= self.max_iter:
                    self.epoch_complete = True            # that was the rest of a resumed partial epoch
                    break
                if args.throughput_json and args.steps and self.run_steps == args.steps - args.throughput_window:
                    torch.cuda.synchronize()
                    tp_t0, tp_seen = time.perf_counter(), 0
                    host = {"loader_and_upload": 0.0, "step_enqueue": 0.0, "rest": 0.0}
                    pf0 = getattr(self, "prefetcher", None)
                    pf_mark = (pf0.t_loader, pf0.t_upload) if pf0 is not None else None
                    t_mark = tp_t0
                elif tp_t0 is not None:                   # host time of the window by phase: what the loop's own thread spends where
                    now = time.perf_counter()
                    host["loader_and_upload"] += now - t_mark
                    t_mark = now
                self.current_step += 1
                self.run_steps += 1
                if self.lr_scheduler is not None:
                    lr = self.lr_scheduler.update_lr(self.current_step)
                    for group in self.optimizer.param_groups:
                        group["lr"] = lr
                    if step_fn is not None:
                        step_fn.set_lr(lr)
                images, labels = exp.preprocess(images, labels, self.input_size)
                if step_fn is not None:
                    res = step_fn.step(images, labels)
                else:
                    self.optimizer.zero_grad()
                    loss_all = self.loss_func.forward(model(images, train=True), labels)
                    loss_all[0].backward()
                    if reducer is not None:
                        if reducer.flat is None:
                            eng = model.engine(images.shape[0], (images.shape[2], images.shape[3]))
                            reducer.attach(eng.home, eng)
                        reducer.reduce_all()
                    self.optimizer.step(grad_scale=1.0 / self.world)
                    if self.ema_model is not None:
                        self.ema_model.update(model)
                    res = self.loss_func._ws.result
                seen += images.shape[0]
                if tp_t0 is not None:
                    tp_seen += images.shape[0]
                    now = time.perf_counter()
                    host["step_enqueue"] += now - t_mark
                    t_mark = now
                if self.current_step % args.log_interval == 0:
                    self.TB_data(res, seen * self.world / (time.time() - t0))
                    self.check_ring_guard()               # the log line has synchronised already
                if tp_t0 is not None:
                    now = time.perf_counter()
                    host["rest"] += now - t_mark
                    t_mark = now
                if args.steps and self.run_steps >= args.steps:
                    if tp_t0 is not None:                 # the window ends HERE: the checkpoint written below is not training time
                        torch.cuda.synchronize()          # (rounds 3-5 measured it inside the window: ~0.25 s, 1.2 ms per step of 200)
                        tp_dt = time.perf_counter() - tp_t0
                    done = True
                    break
            else:
                self.epoch_complete = True
            if self.current_step >= (epoch + 1) * self.max_iter:
                self.epoch_complete = True                # --steps ended the run exactly on the epoch's last iteration
            if self.rank == 0:
                self.save_ckpt("last_epoch")
            if done:
                break
        if tp_t0 is not None:
            if tp_dt is None:                             # the epochs ran out before --steps did
                torch.cuda.synchronize()
                tp_dt = time.perf_counter() - tp_t0
            dt = tp_dt
            if self.rank == 0:
                import json
                rec = {"images_per_s": round(tp_seen * self.world / dt, 2), "ms_per_step": round(dt / max(args.throughput_window, 1) * 1e3, 3),
                       "window_steps": args.throughput_window, "run_steps": self.run_steps, "batch_per_gpu": args.batch_size, "world": self.world,
                       "input_size": list(self.input_size), "prefetch": not args.no_prefetch, "raw_u8": bool(args.raw_u8),
                       "captured_step": step_fn is not None, "log_interval": args.log_interval, "exp_file": args.exp_file,
                       "loader_workers": self.train_loader.num_workers, "loader_pin": bool(self.train_loader.pin_memory),
                       "host_ms_per_step": {k: round(v / max(args.throughput_window, 1) * 1e3, 3) for k, v in host.items()}}
                pf1 = getattr(self, "prefetcher", None)
                if pf1 is not None and pf_mark is not None:      # of loader_and_upload: the loader's own work / the (blocking) upload enqueue
                    rec["host_ms_per_step"]["loader_only"] = round((pf1.t_loader - pf_mark[0]) / max(args.throughput_window, 1) * 1e3, 3)
                    rec["host_ms_per_step"]["upload_enqueue_blocking"] = round((pf1.t_upload - pf_mark[1]) / max(args.throughput_window, 1) * 1e3, 3)
                with open(args.throughput_json, "w") as fh:
                    json.dump(rec, fh)
                print("throughput %s" % json.dumps(rec))
        self.check_ring_guard()
        if self.world > 1:
            torch.distributed.destroy_process_group()

    @staticmethod
    def ring_timeouts():
        from ep24 import _lib
        return _lib.lib().fn["ep24_conv_ring_timeouts"]()

    def check_ring_guard(self):
        """A bounded wait of a loader / consumer ring kernel that gives up carries on with whatever is in LDS (by design: a protocol
        error must never hang the GPU), so the convolutions of that launch are wrong.  Never seen outside development builds; read at
        every log interval, BEFORE every checkpoint and at the end of the run, so that a run with a broken hand-off stops without
        overwriting its last good checkpoint (the reference's cadence: train_24p.py:139-154)."""
        n = self.ring_timeouts()
        if n != 0:
            raise RuntimeError("train_24p.py: %d bounded wait(s) of the loader / consumer ring kernels gave up (ep24_conv_ring_timeouts) at "
                               "step %d: the run stops and no checkpoint is written over the last good one" % (n, self.current_step))

    @staticmethod
    def resumed_step(start_epoch, max_iter):
        """Global step a run that starts at ``start_epoch`` continues from (reference: epoch * max_iter + iter)."""
        return int(start_epoch) * int(max_iter)

    def batches(self):
        """One epoch of (images, labels) on the device.  Default (round 4): the reference's ``DataPrefetcher``
        (data/data_prefetcher.py:16-51) in its ep24 form - the next batch is uploaded (and, for ``--raw-u8`` batches, letterboxed by
        ``TrainTransform.batch``) on a side stream while the current step runs.  ``--no-prefetch``: the loader's
        tensors uploaded on the compute stream, what the reference's 24p loop does (train_24p.py:86-88) - measured at BASELINE
        config 2 it leaves the GPU idle for the 98 MB upload of every step (profiles/r04_trainer.json)."""
        if self.args.no_prefetch:
            for images, labels, _info, _ids in self.train_loader:
                yield images.to(self.device, non_blocking=True), labels.to(self.device, non_blocking=True)
            return
        from ep24.input import DataPrefetcher, TrainTransform
        pf = DataPrefetcher(self.train_loader, tuple(self.input_size), TrainTransform(max_labels=50))
        self.prefetcher = pf                              # its t_loader / t_upload split the throughput record's host time
        if self.args.resident_batch:                      # diagnostic: what the input pipeline costs the GPU (tools/trainer_timing.sh)
            images, labels = pf.next()
            for _ in range(len(self.train_loader)):
                yield images, labels
            return
        while True:
            images, labels = pf.next()
            if images is None:
                return
            yield images, labels

    def TB_data(self, res, ips):
        """One D2H copy of the packed result vector: [0] loss, [1..24] weighted IOU losses, [25] conf, [26] cls,
        [29..52] iou weights, [53] obj_w, [54] cls_w (same scalars as the reference's TB_data)."""
        r = res.detach().float().cpu().tolist()
        if self.rank == 0:
            print("step %d  loss %.4f  conf %.4f  cls %.4f  num_fg %.0f  %.1f img/s" % (self.current_step, r[0], r[25], r[26], r[55], ips))
        if self.tblogger is None:
            return
        s = self.current_step
        for i in range(24):
            self.tblogger.add_scalar("Loss/IOU_Loss_{}".format(i), r[1 + i], s)
            self.tblogger.add_scalar("Weights/iou_w_{}".format(i), r[29 + i], s)
        self.tblogger.add_scalar("Loss/Total_Loss", r[0], s)
        self.tblogger.add_scalar("Loss/Confi_Loss", r[25], s)
        self.tblogger.add_scalar("Loss/Class_Loss", r[26], s)
        self.tblogger.add_scalar("Weights/obj_w", r[53], s)
        self.tblogger.add_scalar("Weights/cls_w", r[54], s)

    def save_ckpt(self, ckpt_name, update_best_ckpt=False):
        # the reference's three keys (train_24p.py:144-148) plus the global step: "start_epoch" is the first epoch that is NOT complete
        # (a run cut short by --steps inside an epoch resumes that epoch at the step it stopped, not at the next epoch's first step)
        self.check_ring_guard()                           # raises before anything is written
        state = {"start_epoch": self.epoch + (1 if self.epoch_complete else 0), "model": self.model.state_dict(),
                 "optimizer": self.optimizer.state_dict(), "global_step": self.current_step}
        if self.ema_model is not None:
            state["ema_model"], state["ema_updates"] = self.ema_model.ema.state_dict(), self.ema_model.updates
        save_checkpoint(state, update_best_ckpt, self.file_name, ckpt_name)


def make_parser():
    p = argparse.ArgumentParser("YOLOX train parser")
    p.add_argument("-b", "--batch_size", type=int, default=4, help="batch size (per GPU)")
    p.add_argument("-l", "--learn_rate", type=float, default=0.001, help="learn rate")
    p.add_argument("-s", "--start_device", default=0, type=int, help="device for start count")
    p.add_argument("-d", "--devices", default=1, type=int, help="device for training")
    p.add_argument("-f", "--exp_file", default=None, type=str, help="plz input your experiment description file")
    p.add_argument("--resume", default=False, action="store_true", help="resume training")
    p.add_argument("-c", "--ckpt", default=None, type=str, help="checkpoint file")
    p.add_argument("-e", "--start_epoch", default=None, type=int, help="resume training start epoch")
    p.add_argument("--num_machines", default=1, type=int, help="num of node for training")
    # additions of this build
    p.add_argument("--device", default="cuda", type=str, help="cuda or cuda:N (the ep24 path has no CPU fallback: cpu is refused)")
    p.add_argument("--synthetic", action="store_true", help="synthetic images / labels with the reference's layout (the only source here)")
    p.add_argument("--output-dir", default=None, type=str, help="overrides exp.output_dir")
    p.add_argument("--synthetic-len", default=0, type=int, help="images per synthetic epoch (overrides exp.synthetic_len; a checkpoint is written per epoch)")
    p.add_argument("--steps", default=0, type=int, help="stop after this many steps (0 = run all epochs)")
    p.add_argument("--log-interval", default=10, type=int)
    p.add_argument("--no-graph", action="store_true", help="reference-style eager loop instead of the captured step")
    p.add_argument("--sched", action="store_true", help="follow exp.get_lr_scheduler (yoloxwarmcos) instead of a constant rate")
    p.add_argument("--ema", action="store_true", help="keep a ModelEMA copy of the model (saved as ema_model)")
    p.add_argument("--l1", action="store_true", help="switch use_l1 on from epoch exp.L1_epoch")
    p.add_argument("--prefetch", action="store_true", help="(default since round 4; kept for old command lines) upload the next batch on a side stream: ep24.input.DataPrefetcher")
    p.add_argument("--loader-workers", default=None, type=int, help="processes of the synthetic loader (default: the Exp's loader_workers)")
    p.add_argument("--loader-pin", default=None, type=int, choices=(0, 1), help="page-locked batches from the loader (default: with loader processes and fp32 batches)")
    p.add_argument("--no-prefetch", action="store_true", help="the reference's loop: upload every batch on the compute stream (train_24p.py:86-88)")
    p.add_argument("--raw-u8", action="store_true", help="(default since round 5; kept for old command lines) the synthetic source hands over uint8 HWC "
                   "images + normalised label rows; letterbox and label scaling run on the GPU behind the prefetcher (SURVEY 8f N1)")
    p.add_argument("--fp32-batches", action="store_true", help="the reference's form of the source: ready-made fp32 canvases [B,3,S,S] + label tables from "
                   "the loader (datasets/data_augment.py TrainTransform on the host side)")
    p.add_argument("--resident-batch", action="store_true", help="diagnostic: the epoch's first batch stays on the device and is fed again and again "
                   "(no upload, no letterbox after the first step): the entry point's rate without its input pipeline")
    p.add_argument("--dp-wire", default="fp32", choices=["fp32", "bf16"], help="wire format of the gradient all-reduce under torch.distributed.run")
    p.add_argument("--throughput-json", default=None, type=str, help="with --steps N: write images/s over the run's last --throughput-window steps "
                   "(synchronised at both ends) to this file")
    p.add_argument("--throughput-window", default=200, type=int)
    return p


def main(exp, args):
    if args.output_dir:
        exp.output_dir = args.output_dir
    if args.synthetic_len:
        exp.synthetic_len = args.synthetic_len
    trainer = Trainer(exp, args)
    trainer.train()
    return trainer


if __name__ == "__main__":
    args = make_parser().parse_args()
    exp = get_exp(args.exp_file)
    main(exp, args)
