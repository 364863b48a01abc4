"""Fisheye sector warp on the GPU: mirror of ``Image_Distortion.sector_distort`` (yolox/demo_featuremap.py:238-328).

Host side (this file) computes what is 1-D and cheap exactly as the reference does with numpy - the angle / radius
tables, the target row count T from the arc length, the crop box - and caches, per (Theta, T), the device-resident
winner map built by ``ep24_sector_map``.  Per image only ``ep24_resize_linear_u8`` + ``ep24_sector_gather`` (+
``ep24_mask_bbox``) run.  Inputs may be numpy HWC uint8 arrays (drop-in: results come back as numpy) or CUDA tensors.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

CANVAS = 1000
N_ANG = 165 * 80
MAX_ROWS = CANVAS - 100


def _tables(theta_deg, h, w, custom_rows):
    assert (theta_deg >= 15) and (theta_deg <= 180), "Theta is not in range 15°-180°!"
    canvas_w = int(CANVAS * np.sin(theta_deg / 2 * np.pi / 180) * 2)
    start = (180 - theta_deg) / 2
    ang = np.linspace(start, start + theta_deg, N_ANG, True) * np.pi / 180
    c, s = np.cos(ang), np.sin(ang)
    if custom_rows is None:
        ends = (c * CANVAS).astype(np.int16) + (s * CANVAS).astype(np.int16) * 1j
        T = int(np.clip(int(np.unique(ends).shape[0] * (h / w)), 0, MAX_ROWS))
    else:
        assert custom_rows <= MAX_ROWS, "Custom row should be limited in 900!"
        T = int(custom_rows)
    rho = np.linspace(CANVAS - T, CANVAS, T)
    return canvas_w, c, s, rho, T


def _crop_box(canvas_w, c, s, rho):
    """min / max of the destination coordinates over all (angle, radius) pairs from the 1-D tables: rho > 0, so
    rho*cos and rho*sin are monotone in each factor and truncation / clipping preserve order."""
    def dest_x(v):
        return int(np.clip(np.int16(v) + canvas_w / 2 - 1, 0, canvas_w).astype(np.int16))

    def dest_y(v):
        return int(np.clip((CANVAS - np.int16(v)) - 1, 0, CANVAS))

    cmax, cmin, smax, smin = c.max(), c.min(), s.max(), s.min()
    rmax, rmin = rho.max(), rho.min()
    vx_max = cmax * (rmax if cmax >= 0 else rmin)
    vx_min = cmin * (rmax if cmin <= 0 else rmin)
    vy_max = smax * (rmax if smax >= 0 else rmin)
    vy_min = smin * (rmax if smin <= 0 else rmin)
    return dest_y(vy_max), dest_y(vy_min), dest_x(vx_min), dest_x(vx_max)       # y0, y1, x0, x1


class Image_Distortion:
    def __init__(self, device="cuda:0"):
        self.draw_temp_size = CANVAS
        self.sector_length = MAX_ROWS
        self.draw_resolution = 80
        self.device = torch.device(device)
        self.two_pass = False
        self._maps = {}

    def _map(self, theta_deg, h, w, custom_rows):
        canvas_w, c, s, rho, T = _tables(theta_deg, h, w, custom_rows)
        key = (float(theta_deg), T)
        if key not in self._maps:
            dev = self.device
            winner = torch.full((CANVAS * canvas_w,), -1, dtype=torch.int32, device=dev)
            ct, st, rt = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (c, s, rho))
            call("sector_map", ptr(ct), ptr(st), N_ANG, ptr(rt), T, canvas_w, CANVAS, ptr(winner), stream_ptr())
            torch.cuda.current_stream().synchronize()        # ct/st/rt may be freed after this point (one-time build)
            self._maps[key] = (winner, canvas_w, _crop_box(canvas_w, c, s, rho), T)
        return self._maps[key]

    def source_index(self, theta_deg, h, w, custom_rows=None):
        """[out_h, out_w] int32 flat index into the resized image (-1 = fill): the scatter's winners."""
        winner, cw, (y0, y1, x0, x1), T = self._map(theta_deg, h, w, custom_rows)
        out = torch.empty((y1 - y0, x1 - x0), dtype=torch.int32, device=self.device)
        call("sector_gather", None, ptr(winner), cw, y0, x0, y1 - y0, x1 - x0, T, N_ANG, None, 0, ptr(out), stream_ptr())
        return out

    def _warp(self, img, winner, cw, box, T, fill):
        y0, y1, x0, x1 = box
        h, w = img.shape[0], img.shape[1]
        out = torch.empty((y1 - y0, x1 - x0, 3), dtype=torch.uint8, device=self.device)
        if self.two_pass:                      # cv2.resize(image, (13200, T)) materialised, then the gather
            resized = torch.empty((T, N_ANG, 3), dtype=torch.uint8, device=self.device)
            call("resize_linear_u8", ptr(img), h, w, ptr(resized), T, N_ANG, stream_ptr())
            call("sector_gather", ptr(resized), ptr(winner), cw, y0, x0, y1 - y0, x1 - x0, T, N_ANG, ptr(out), fill, None, stream_ptr())
        else:                                  # the same texels sampled on the fly: no 35 MB intermediate
            call("sector_warp_u8", ptr(img), h, w, ptr(winner), cw, y0, x0, y1 - y0, x1 - x0, T, N_ANG, ptr(out), fill, stream_ptr())
        return out

    def distort_batch(self, images, masks, thetas, custom_rows=None):
        """``sector_distort`` for a batch, without a host synchronisation: images / masks are lists of HWC uint8 device tensors
        (masks may be None), thetas one angle per image.  Returns (warped images, warped masks or None, boxes [n,4] int32 on
        the device as (xmin, ymin, xmax, ymax) of the warped mask's non-zero pixels, xmax = -1 when the mask came out empty).
        The winner map of a (Theta, rows) pair is built once and cached, so a training loop only runs the warp launches."""
        _lib.require_gpu()
        n = len(images)
        outs, mouts = [], [] if masks is not None else None
        boxes = torch.tensor([[2 ** 31 - 1, 2 ** 31 - 1, -1, -1]] * max(n, 1), dtype=torch.int32, device=self.device)
        for i, (img, th) in enumerate(zip(images, thetas)):
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_cuda:
                raise IndexError("expected HWC uint8 images on the GPU")
            winner, cw, box, T = self._map(th, img.shape[0], img.shape[1], custom_rows)
            outs.append(self._warp(img.contiguous(), winner, cw, box, T, 114))
            if masks is not None:
                m = self._warp(masks[i].contiguous(), winner, cw, box, T, 0)
                mouts.append(m)
                call("mask_bbox", ptr(m), m.shape[0], m.shape[1], ptr(boxes, 4 * i), stream_ptr())
        return outs, mouts, boxes[:n]

    def sector_distort(self, image, mask, Theta=60, custom_rows=None):
        _lib.require_gpu()
        as_numpy = isinstance(image, np.ndarray)
        img = torch.as_tensor(image).to(self.device).contiguous()
        msk = torch.as_tensor(mask).to(self.device).contiguous()
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise IndexError("expected HWC uint8 images")
        (new_image,), (new_mask,), boxes = self.distort_batch([img], [msk], [Theta], custom_rows)
        xmin, ymin, xmax, ymax = boxes[0].tolist()
        new_bbox = [xmin, ymin, xmax - xmin, ymax - ymin] if xmax >= 0 else []
        return (new_image.cpu().numpy() if as_numpy else new_image), new_bbox
