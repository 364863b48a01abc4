"""24-point label generation on the GPU: host mirror of ``Polygon_24`` (yolox_24p/datasets/2+24_labels_create.py).

``rotation_for_24p(center_x, center_y, mask)`` keeps the reference's signature and return values (24 contour points
``[24,2]`` int, 24 distances ``[24]`` float64) for one instance mask; ``rays_batch`` does any number of objects in one
launch (one workgroup per object and ray), ``hull_areas`` the convex-hull area of the acceptance filter
(``cv2.contourArea(cv2.convexHull(pts))``, :175-176), ``label_rows`` the two txt rows of an accepted annotation
(:181-193) and ``save_rows`` / ``load_rows`` the on-disk format (``%d`` + 50 or 26 times ``%0.4f``, :214-236; read back
by ``np.loadtxt`` in datasets/coco24p.py:46).  Decoding COCO polygons / RLE into masks (pycocotools ``annToMask``) stays
on the host and is not part of this module.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

_ROT = None


def _rot_table(device):
    """cos / sin of k*15 degrees with numpy, exactly as the reference forms them (:84-86)."""
    global _ROT
    if _ROT is None or _ROT.device != device:
        th = np.array([k * 15 * np.pi / 180 for k in range(24)])
        _ROT = torch.from_numpy(np.stack([np.cos(th), np.sin(th)], 1).copy()).to(device)
    return _ROT


def rays_batch(masks, centres, device="cuda:0"):
    """masks: list of uint8 arrays / tensors [H_i, W_i] (non-zero = object); centres: [n,2] (x, y) floats.
    Returns (points int32 [n,24,2], distances float64 [n,24]) as device tensors."""
    _lib.require_gpu()
    n = len(masks)
    dev = torch.device(device)
    pts = torch.empty(n, 24, 2, dtype=torch.int32, device=dev)
    rad = torch.empty(n, 24, dtype=torch.float64, device=dev)
    if n == 0:
        return pts, rad
    desc, flat, off = [], [], 0
    for m in masks:
        m = m if isinstance(m, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(m))
        if m.dim() != 2 or m.dtype != torch.uint8:
            raise ValueError("instance masks are uint8 [H,W] arrays")
        H, W = int(m.shape[0]), int(m.shape[1])
        L = int(np.sqrt(np.power(H, 2) + np.power(W, 2)))                     # :68
        if H + W + L >= 32768:
            raise ValueError("image too large for the reference's int16 ray coordinates")
        ns = int(np.ceil((L - 0) / 0.2))                                      # len(np.arange(0, L, 0.2)), :74
        desc.append([off, H, W, L, ns, W])
        flat.append(m.reshape(-1))
        off += H * W
    buf = torch.cat([f.to(dev, non_blocking=True) for f in flat])
    desc_t = torch.tensor(desc, dtype=torch.int64, device=dev)
    cen = torch.as_tensor(np.asarray(centres, dtype=np.float64).reshape(n, 2)).to(dev)
    rot = _rot_table(dev)
    for lo in range(0, n, 65535):
        hi = min(n, lo + 65535)
        call("ray24", ptr(buf), ptr(desc_t, lo * 6), ptr(cen, lo * 2), ptr(rot), hi - lo, ptr(pts, lo * 48), ptr(rad, lo * 24),
             stream_ptr())
    return pts, rad


def hull_areas(points):
    """points int32 [n,24,2] on the GPU -> float64 [n]."""
    _lib.require_gpu()
    points = points.contiguous()
    out = torch.empty(points.shape[0], dtype=torch.float64, device=points.device)
    call("hull_area24", ptr(points), points.shape[0], ptr(out), stream_ptr())
    return out


def rotation_for_24p(center_x, center_y, mask):
    """Drop-in for ``Polygon_24.rotation_for_24p`` (one object; numpy in, numpy out)."""
    pts, rad = rays_batch([np.ascontiguousarray(mask, dtype=np.uint8)], [[center_x, center_y]])
    pts, rad = pts[0].cpu().numpy().astype(np.int64), rad[0].cpu().numpy()
    if (pts < 0).any():
        raise ValueError("attempt to get argmin of an empty sequence")       # what np.argmin raises in the reference
    return pts, rad


def label_rows(class_idx, centres, shapes, points, radii, hull, label_areas, area_t_low=0.5, area_t_high=1.5):
    """Host assembly of the txt rows for n annotations (:177-193).  Returns (keep mask [n], cord rows [m,51] float64,
    radius rows [m,27] float64) for the annotations that pass the hull-area filter."""
    points, radii, hull = [np.asarray(v.cpu()) if isinstance(v, torch.Tensor) else np.asarray(v) for v in (points, radii, hull)]
    centres, shapes, label_areas = np.asarray(centres, np.float64), np.asarray(shapes), np.asarray(label_areas, np.float64)
    keep = ~((hull <= label_areas * area_t_low) | (hull >= label_areas * area_t_high))
    cord_rows, rad_rows = [], []
    for i in np.nonzero(keep)[0]:
        H, W = int(shapes[i][0]), int(shapes[i][1])
        diag = np.sqrt(np.power(H, 2) + np.power(W, 2))
        cord = points[i].reshape(1, -1).squeeze(0).astype(np.float32)
        cord[0::2] = cord[0::2] / W
        cord[1::2] = cord[1::2] / H
        head = np.array([class_idx[i]]), np.array([centres[i][0] / W, centres[i][1] / H])
        cord_rows.append(np.concatenate((head[0], head[1], cord), axis=0))
        rad_rows.append(np.concatenate((head[0], head[1], radii[i] / diag), axis=0))
    return keep, np.array(cord_rows).reshape(-1, 51), np.array(rad_rows).reshape(-1, 27)


def save_rows(path, rows):
    """One image's label file (:214-236): ``%d`` then ``%0.4f`` per remaining column; an empty file for no labels."""
    rows = np.asarray(rows)
    if rows.shape[0]:
        np.savetxt(str(path), rows, fmt=["%d"] + ["%0.4f"] * (rows.shape[1] - 1))
    else:
        np.savetxt(str(path), rows)


def load_rows(path):
    """What the dataset does with a label file (datasets/coco24p.py:46, :85-86): float rows, one row kept 2-D."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        info = np.loadtxt(str(path), dtype=float)
    return info[np.newaxis, :] if info.ndim == 1 and info.size else info
