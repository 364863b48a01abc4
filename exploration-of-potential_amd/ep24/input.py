"""Input pipeline on the GPU: ``preproc`` / ``TrainTransform`` (yolox_24p/datasets/data_augment.py:109-174)
for whole batches, and the side-stream ``DataPrefetcher`` (yolox_24p/data/data_prefetcher.py:8-51).

The host hands over what the decoder produced - raw uint8 HWC images of any size and the label rows of the txt files
(class + 50 normalised coordinates) - and gets the network input ``[n,3,S,S]`` fp32 and the label table ``[n,50,51]``
fp32 on the device: one upload of the raw bytes (5x fewer than the fp32 canvas the reference ships) and two launches.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def letterbox_geometry(h, w, input_size):
    """r and the resized size exactly as ``preproc`` computes them (data_augment.py:117-121)."""
    r = min(input_size[0] / h, input_size[1] / w)
    return r, int(h * r), int(w * r)


def preproc_batch(images, input_size, device="cuda:0", out=None):
    """images: list of uint8 [h,w,3] arrays / tensors (host or device).  Returns (tensor [n,3,S_h,S_w] fp32 on the
    device, list of r)."""
    _lib.require_gpu()
    dev = torch.device(device)
    n = len(images)
    S_h, S_w = int(input_size[0]), int(input_size[1])
    if out is None:
        out = torch.empty(n, 3, S_h, S_w, dtype=torch.float32, device=dev)
    if n == 0:
        return out, []
    desc, scales, flat, rs, off = [], [], [], [], 0
    for im in images:
        im = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im))
        if im.dim() != 3 or im.shape[2] != 3 or im.dtype != torch.uint8:
            raise ValueError("preproc_batch takes uint8 [h,w,3] images")
        h, w = int(im.shape[0]), int(im.shape[1])
        r, rh, rw = letterbox_geometry(h, w, (S_h, S_w))
        desc.append([off, h, w, 3 * w, rh, rw])
        scales.append([1.0 / (rw / w) if rw else 1.0, 1.0 / (rh / h) if rh else 1.0])      # cv::resize: scale = 1/(dsize/ssize)
        flat.append(im.reshape(-1))
        rs.append(r)
        off += h * w * 3
    # (gathering the images into ONE pinned host buffer first was tried in round 4 and is slower: CPU writes into hipHostMalloc'ed
    # memory are uncached - 24 MB took ~30 ms per batch; the runtime's own staging of pageable copies is the fast path here)
    buf = torch.cat([f.to(dev, non_blocking=True) for f in flat])
    desc_t = torch.tensor(desc, dtype=torch.int64, device=dev)
    sc_t = torch.tensor(scales, dtype=torch.float64, device=dev)
    for lo in range(0, n, 65535):
        hi = min(n, lo + 65535)
        call("preproc_u8", ptr(buf), ptr(desc_t, lo * 6), ptr(sc_t, lo * 2), hi - lo, ptr(out, lo * 3 * S_h * S_w), S_h, S_w,
             stream_ptr())
    return out, rs


def labels_batch(targets, sizes, rs, max_labels=50, device="cuda:0", out=None):
    """targets: list of [k_i,51] float arrays (normalised txt rows; ``[k,0]``-shaped or empty for an image without
    labels); sizes: list of (h, w); rs: the r of every image.  Returns [n,max_labels,51] fp32 on the device."""
    _lib.require_gpu()
    dev = torch.device(device)
    n = len(targets)
    if out is None:
        out = torch.empty(n, max_labels, 51, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    rows, off = [], [0]
    for t in targets:
        t = np.asarray(t, dtype=np.float64)
        t = t.reshape(-1, 51) if t.size else np.zeros((0, 51))
        rows.append(t)
        off.append(off[-1] + t.shape[0])
    allrows = np.concatenate(rows, 0) if off[-1] else np.zeros((1, 51))
    rows_t = torch.from_numpy(np.ascontiguousarray(allrows)).to(dev)
    off_t = torch.tensor(off, dtype=torch.int64, device=dev)
    whr = torch.tensor([[float(w), float(h), float(r)] for (h, w), r in zip(sizes, rs)], dtype=torch.float64, device=dev)
    call("preproc_labels", ptr(rows_t), ptr(off_t), ptr(whr), n, ptr(out), max_labels, stream_ptr())
    return out


def preproc(img, input_size, swap=(2, 0, 1)):
    """Drop-in for the reference ``preproc`` (one image): returns (CHW fp32 device tensor, r, None) - the third value
    (the uint8 HWC canvas) is not materialised."""
    if swap != (2, 0, 1):
        raise NotImplementedError("ep24.input.preproc produces the network layout (2, 0, 1) only")
    out, rs = preproc_batch([img], input_size)
    return out[0], rs[0], None


class TrainTransform:
    """``TrainTransform(max_labels, flip_prob)`` of the reference (data_augment.py:131-174; the flip probability is
    stored and unused there as well).  ``__call__(image, targets, input_dim)`` handles one image, ``batch`` many."""

    def __init__(self, max_labels=50, flip_prob=0.5, hsv_prob=1.0, fisheye=None, seed=0):
        """``fisheye=(theta_lo, theta_hi)``: every image first goes through the sector warp (``Image_Distortion.sector_distort``,
        yolox/demo_featuremap.py:244-328) with an angle drawn from that range, on the GPU, and the WARPED image is letterboxed -
        the on-GPU fisheye augmentation of BASELINE config 5.  The reference has no label transform for the warp (it returns
        the warped mask's bounding box only), so the label rows pass through unchanged: a stress configuration, as in the
        reference's demo, not a training recipe."""
        self.max_labels, self.flip_prob = max_labels, flip_prob
        self.fisheye = fisheye
        self._rng = np.random.RandomState(seed)
        self._dist = None

    def warp(self, images, device="cuda:0"):
        from .sector import Image_Distortion
        if self._dist is None:
            self._dist = Image_Distortion(device)
        lo, hi = self.fisheye
        thetas = [int(self._rng.randint(lo, hi + 1)) for _ in images]
        dev_imgs = [torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im).to(device) for im in images]
        return self._dist.distort_batch(dev_imgs, None, thetas)[0]

    def batch(self, images, targets, input_dim, out_images=None, out_labels=None):
        if self.fisheye is not None:
            images = self.warp(images)
        imgs, rs = preproc_batch(images, input_dim, out=out_images)
        labs = labels_batch(targets, [im.shape[:2] for im in images], rs, self.max_labels, out=out_labels)
        return imgs, labs

    def __call__(self, image, targets, input_dim):
        imgs, labs = self.batch([image], [targets], input_dim)
        return imgs[0], labs[0]


class DataPrefetcher:
    """The reference's prefetcher (data/data_prefetcher.py): the next batch is prepared on a side stream while the
    current step runs; ``next()`` makes the compute stream wait for it.  ``loader`` yields either ready tensors
    ``(images, labels, info, ids)`` (as the reference's loader does) or raw batches ``(list of uint8 HWC images, list of
    label rows, info, ids)`` - then ``transform.batch`` (upload + the two preproc launches) runs on the side stream too."""

    def __init__(self, loader, input_size=(640, 640), transform=None):
        _lib.require_gpu()
        self.loader = iter(loader)
        self.stream = torch.cuda.Stream()
        self.input_size = input_size
        self.transform = transform or TrainTransform()
        # raw batches land in three rotating (images, labels) buffers: batch k + 1 is written into the buffers of batch k - 2, behind
        # the event recorded on the compute stream when batch k - 1 was handed out (everything enqueued before it - the step that
        # consumed batch k - 2 - has then run).  No allocation and no record_stream per batch.
        self._bufs, self._k, self._events = [None, None, None], 0, []
        # host seconds of this thread inside the loader (next(): dataset + collate) and inside the upload / letterbox enqueue; the
        # second one BLOCKS in the pageable copies until the device buffer of batch k - 2 is free, i.e. it is where a GPU-bound
        # loop waits for the GPU (profiles/r05_trainer_8sets.json), not host work
        self.t_loader = self.t_upload = 0.0
        self.preload()

    def preload(self):
        import time
        t0 = time.perf_counter()
        try:
            images, targets, _, _ = next(self.loader)
        except StopIteration:
            self.next_input = self.next_target = None
            return
        t1 = time.perf_counter()
        self.t_loader += t1 - t0
        try:
            self._upload(images, targets)
        finally:
            self.t_upload += time.perf_counter() - t1

    def _upload(self, images, targets):
        self._k += 1
        with torch.cuda.stream(self.stream):
            if isinstance(images, torch.Tensor):
                self.next_input = images.cuda(non_blocking=True)
                self.next_target = targets.cuda(non_blocking=True)
                self._owned = False
            else:
                slot = self._k % 3
                n, (S_h, S_w) = len(images), (int(self.input_size[0]), int(self.input_size[1]))
                b = self._bufs[slot]
                if b is None or b[0].shape != (n, 3, S_h, S_w):
                    b = (torch.empty(n, 3, S_h, S_w, dtype=torch.float32, device="cuda"),
                         torch.empty(n, self.transform.max_labels, 51, dtype=torch.float32, device="cuda"))
                    self._bufs[slot] = b
                if len(self._events) >= 2:
                    self.stream.wait_event(self._events[-2])
                self.next_input, self.next_target = self.transform.batch(images, targets, self.input_size, out_images=b[0], out_labels=b[1])
                self._owned = True

    def next(self):
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        inp, tgt = self.next_input, self.next_target
        if inp is not None and not self._owned:
            inp.record_stream(cur)
            tgt.record_stream(cur)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._events = (self._events + [ev])[-3:]
        self.preload()
        return inp, tgt
