"""``Exp``: model / data / optimizer factory of the 24p detector (reference exp/yolox_base.py:11-137) on top of the
MI355X-native path: ``get_model`` builds the ep24 parameter tree (HIP plan behind ``YOLOX.forward``),
``get_optimizer`` the fused nesterov SGD, ``get_data_loader`` a synthetic source with the reference's label layout
(the reference's COCO24P loader reads hard-coded /home/gaoyu/... paths, coco24p.py:19-20, and is out of scope)."""
import _path  # noqa: F401
import torch.nn as nn

from .base_exp import BaseExp


class Exp(BaseExp):
    def __init__(self):
        super().__init__()
        # model
        self.num_classes = 80
        self.depth = 1.00
        self.width = 1.00
        self.act = "silu"
        self.backbone_type = "darknet"   # 'darknet' | 'resnet' | 'densenet' | 'vgg': the switch of yolox/models/yolo_pafpn.py:31-38
        # data
        self.data_num_workers = 8
        self.input_size = (640, 640)
        self.multiscale_range = 5
        self.train_ann = "instances_train2017.json"
        self.val_ann = "instances_val2017.json"
        self.synthetic_len = 64          # images per synthetic epoch
        self.loader_workers = 4          # processes of the synthetic fp32 loader (train_24p.py --loader-workers)
        self.synthetic_gts = 10
        # training
        self.warmup_epochs = 5
        self.max_epoch = 300
        self.L1_epoch = 100
        self.warmup_lr = 0
        self.basic_lr_per_img = 0.01 / 64.0
        self.scheduler = "yoloxwarmcos"
        self.no_aug_epochs = 100
        self.min_lr_ratio = 0.05
        self.ema = True
        self.weight_decay = 5e-4
        self.momentum = 0.9
        self.print_interval = 10
        self.eval_interval = 10
        self.exp_name = "yolox_24p"
        # testing
        self.test_size = (640, 640)
        self.test_conf = 0.01
        self.nmsthre = 0.65

    def get_model(self):
        from models import YOLOX, YOLOPAFPN, YOLOXHead
        if getattr(self, "model", None) is None:
            in_channels = [256, 512, 1024]
            backbone = YOLOPAFPN(self.depth, self.width, in_channels=in_channels, act=self.act, backbone_type=self.backbone_type)
            head = YOLOXHead(self.num_classes, self.width, in_channels=in_channels, act=self.act)
            self.model = YOLOX(backbone, head)
        for m in self.model.modules():                    # init_yolo, yolox_base.py:58-62
            if isinstance(m, nn.BatchNorm2d):
                m.eps = 1e-3
                m.momentum = 0.03
        self.model.head.initialize_biases(1e-2)
        return self.model

    def get_data_loader(self, batch_size, raw_u8=False, workers=None, pin=None):
        """``workers``: loader processes (None: ``loader_workers`` of the Exp for ready-made fp32 batches - collating 98 MB per step in
        the training process itself left the trainer at 0.21 of bench.py, four processes + page-locked batches bring 0.91 - and 0 for
        the raw uint8 source, whose batches are lists of cached images: 0.96, profiles/r04_trainer.json); ``pin``: page-locked batches
        (None: with loader processes and fp32 batches only)."""
        from datasets import ResumableSampler, SyntheticDataset, raw_collate
        import torch
        import os
        self.dataset = SyntheticDataset(self.synthetic_len, tuple(self.input_size), self.synthetic_gts, self.num_classes, raw=raw_u8)
        world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
        sampler = None
        if world > 1:                                     # every rank walks its own shard of the epoch
            sampler = torch.utils.data.distributed.DistributedSampler(self.dataset, num_replicas=world, rank=rank, shuffle=False)
        else:
            sampler = torch.utils.data.SequentialSampler(self.dataset)
        sampler = ResumableSampler(sampler)               # .start: indices a resumed run has already trained on (train_24p.py)
        if workers is None:
            workers = 0 if raw_u8 else getattr(self, "loader_workers", 0)
        workers = int(workers)
        if pin is None:
            pin = workers > 0 and not raw_u8
        kw = dict(batch_size=batch_size, num_workers=workers, drop_last=True, sampler=sampler, pin_memory=bool(pin))
        if workers > 0:
            kw.update(persistent_workers=True, prefetch_factor=2)
        if raw_u8:                                        # lists of uint8 images + label rows: letterboxed on the GPU by the prefetcher
            kw["collate_fn"] = raw_collate
        return torch.utils.data.DataLoader(self.dataset, **kw)

    def random_resize(self, data_loader=None, epoch=None):
        """A multiscale input size, multiple of 32, aspect ratio of input_size (exp/yolox_base.py:93-107); plans are cached per size."""
        import random
        size_factor = self.input_size[1] * 1.0 / self.input_size[0]
        if not hasattr(self, "random_size"):
            self.random_size = (int(self.input_size[0] / 32) - self.multiscale_range, int(self.input_size[0] / 32) + self.multiscale_range)
        size = random.randint(*self.random_size)
        return (int(32 * size), 32 * int(size * size_factor))

    def preprocess(self, inputs, targets, tsize):
        scale_y = tsize[0] / self.input_size[0]
        scale_x = tsize[1] / self.input_size[1]
        if scale_x != 1 or scale_y != 1:
            inputs = nn.functional.interpolate(inputs, size=tsize, mode="bilinear", align_corners=False)
            targets[..., 1::2] = targets[..., 1::2] * scale_x
            targets[..., 2::2] = targets[..., 2::2] * scale_y
        return inputs, targets

    def get_optimizer(self, lr):
        """Takes the learning rate (not the batch size), like the 24p trainer (yolox_base.py:120-124)."""
        from ep24.train import SGD
        self.optimizer = SGD(self.model.parameters(), lr=lr, momentum=self.momentum, nesterov=True, model=self.model)
        return self.optimizer

    def get_lr_scheduler(self, lr, iters_per_epoch, **kwargs):
        from utils import LRScheduler
        return LRScheduler(self.scheduler, lr, iters_per_epoch, self.max_epoch, warmup_epochs=self.warmup_epochs,
                           warmup_lr_start=self.warmup_lr, no_aug_epochs=self.no_aug_epochs, min_lr_ratio=self.min_lr_ratio)
