"""Oracle: 24-point label generation.  Test infrastructure only.

Restates ``Polygon_24.rotation_for_24p`` (yolox_24p/datasets/2+24_labels_create.py:61-116): 24 rays at 15 degree steps
from the object's box centre, sampled every 0.2 px and truncated to integer pixels; on every ray the unmasked pixel
nearest to the centre (inside the image grown by a one-pixel ring) is the contour point.  The reference paints every ray
into an image padded by the diagonal and reads it back; here the same pixel set is handled as a coordinate list - the
result (including the row-major tie-break of ``np.where`` + ``np.argmin`` and the one-pixel offset of the ring that the
distances carry, as written) is pinned by tests/golden/g13_labels24.npz, produced by the reference function itself
with ``np.pad`` standing in for the absent ``cv2.copyMakeBorder(BORDER_CONSTANT, 0)``.

``hull_area`` restates the acceptance filter's ``cv2.contourArea(cv2.convexHull(points))`` (:175-180) as monotone-chain
hull + shoelace; cv2 is not installed here, so that function is "parity unpinned" against cv2 and is checked against
scipy.spatial.ConvexHull instead (the area of a convex hull does not depend on the hull algorithm).
"""
import numpy as np


def ray_tables():
    """cos / sin of k*15 degrees exactly as the reference forms them (:84-86)."""
    th = np.array([k * 15 * np.pi / 180 for k in range(24)])
    return np.cos(th), np.sin(th)


def rotation_for_24p(center_x, center_y, mask):
    H, W = mask.shape[0], mask.shape[1]
    L = int(np.sqrt(np.power(H, 2) + np.power(W, 2)))                        # :68
    xs = np.arange(0, L, 0.2)                                                # :74
    cos_t, sin_t = ray_tables()
    pts, rad = [], []
    for k in range(24):
        ex = (cos_t[k] * xs).astype(np.int16)                                # :88, the y row of the line is zero
        ey = (sin_t[k] * xs).astype(np.int16)
        px = (ex + center_x + L).astype(np.int16).astype(np.int64)            # :94-95, stored back into the int16 array
        py = (ey + center_y + L).astype(np.int16).astype(np.int64)
        iy, ix = py - L, px - L                                              # image coordinates
        inside = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
        covered = np.zeros(len(xs), dtype=bool)
        covered[inside] = mask[iy[inside], ix[inside]] != 0                  # :98
        my, mx = iy + 1, ix + 1                                              # :100, window grown by one pixel
        keep = (~covered) & (my >= 0) & (my < H + 2) & (mx >= 0) & (mx < W + 2)
        my, mx = my[keep], mx[keep]
        order = np.lexsort((mx, my))                                         # np.where order: rows, then columns
        my, mx = my[order], mx[order]
        dist = np.sqrt(np.power(mx - center_x, 2) + np.power(my - center_y, 2))   # :103
        j = int(np.argmin(dist))
        pts.append(np.array([np.clip(mx[j], 0, W), np.clip(my[j], 0, H)]))   # :106-107
        rad.append(dist[j])
    return np.array(pts), np.array(rad)


def hull_area(points):
    """Area of the convex hull of integer points [n,2] (monotone chain + shoelace)."""
    p = sorted(set((int(x), int(y)) for x, y in np.asarray(points)))
    if len(p) < 3:
        return 0.0

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for q in p:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], q) <= 0:
            lower.pop()
        lower.append(q)
    for q in reversed(p):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], q) <= 0:
            upper.pop()
        upper.append(q)
    h = lower[:-1] + upper[:-1]
    s = 0
    for i in range(len(h)):
        x0, y0 = h[i]
        x1, y1 = h[(i + 1) % len(h)]
        s += x0 * y1 - x1 * y0
    return abs(s) / 2.0


def label_rows(class_idx, center_x, center_y, mask, label_area, area_t_low=0.5, area_t_high=1.5):
    """One annotation -> (cord row [51], radius row [27]) or None when the hull-area filter rejects it (:160-193)."""
    H, W = mask.shape
    pts, rad = rotation_for_24p(center_x, center_y, mask)
    area = hull_area(pts)
    if area <= label_area * area_t_low or area >= label_area * area_t_high:
        return None
    diag = np.sqrt(np.power(H, 2) + np.power(W, 2))
    cord = pts.reshape(1, -1).squeeze(0).astype(np.float32)
    cord[0::2] = cord[0::2] / W
    cord[1::2] = cord[1::2] / H
    head = np.array([class_idx]), np.array([center_x / W, center_y / H])
    return np.concatenate((head[0], head[1], cord), axis=0), np.concatenate((head[0], head[1], rad / diag), axis=0)
