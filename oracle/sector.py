"""Oracle: fisheye sector warp.  Test infrastructure only.

Numpy restatement of Image_Distortion.sector_distort (yolox/demo_featuremap.py:244-328): geometry of the
1000-px-radius annular sector, the integer destination of every (angle, radius) sample, "last writer wins" of
the reference's fancy-index scatter (C iteration order: angle-major, radius-minor), crop box and mask bbox.
The bilinear resize to [T, 13200] is restated from OpenCV's published INTER_LINEAR fixed-point arithmetic
(cv2 is not installed here: that one step is unpinned; the index map is pinned by tests/golden/g8_sector.npz).
"""
import numpy as np

CANVAS = 1000
N_ANG = 165 * 80
MAX_ROWS = CANVAS - 100


def geometry(theta_deg, h, w, custom_rows=None):
    """-> dict(T, canvas_w, cos, sin, rho) following demo_featuremap.py:245-281."""
    assert 15 <= theta_deg <= 180, "Theta is not in range 15-180!"
    canvas_w = int(CANVAS * np.sin(theta_deg / 2 * np.pi / 180) * 2)
    start = (180 - theta_deg) / 2
    ang = np.linspace(start, start + theta_deg, N_ANG, True) * np.pi / 180
    c, s = np.cos(ang), np.sin(ang)
    if custom_rows is None:
        ex = (c * CANVAS).astype(np.int16)
        ey = (s * CANVAS).astype(np.int16)
        arc_len = np.unique(ex + ey * 1j).shape[0]
        T = int(np.clip(int(arc_len * (h / w)), 0, MAX_ROWS))
    else:
        assert custom_rows <= MAX_ROWS
        T = custom_rows
    rho = np.linspace(CANVAS - T, CANVAS, T)
    return dict(T=T, canvas_w=canvas_w, cos=c, sin=s, rho=rho)


def destinations(g):
    """Canvas (y, x) of every sample [N_ANG, T] (int), demo_featuremap.py:282-294."""
    px = (g["cos"][:, None] * g["rho"][None, :]).astype(np.int16)
    py = (g["sin"][:, None] * g["rho"][None, :]).astype(np.int16)
    X = np.clip(px + g["canvas_w"] / 2 - 1, 0, g["canvas_w"]).astype(np.int16)
    Y = np.clip((CANVAS - py) - 1, 0, CANVAS).astype(np.int16)
    return Y.astype(np.int64), X.astype(np.int64)


def winner_map(theta_deg, h, w, custom_rows=None):
    """-> (src [out_h,out_w] int32: flat index row*N_ANG+col into the resized image or -1, crop box, T)."""
    g = geometry(theta_deg, h, w, custom_rows)
    T, cw = g["T"], g["canvas_w"]
    Y, X = destinations(g)
    dest = (Y * cw + X).ravel()                               # sample order = scatter order (a-major, r-minor)
    n = dest.shape[0]
    uniq, first_rev = np.unique(dest[::-1], return_index=True)
    last = n - 1 - first_rev                                  # last writer of every touched canvas pixel
    a, r = last // T, last % T
    src = (T - 1 - r) * N_ANG + (N_ANG - 1 - a)               # img_resize[ptx[:, ::-1], pty[::-1, :]]
    canvas = np.full(CANVAS * cw, -1, dtype=np.int64)
    canvas[uniq] = src
    canvas = canvas.reshape(CANVAS, cw)
    y0, y1, x0, x1 = int(Y.min()), int(Y.max()), int(X.min()), int(X.max())
    return canvas[y0:y1, x0:x1].astype(np.int32), (y0, y1, x0, x1), T


def mask_bbox(mask_out):
    ys, xs = np.nonzero(mask_out[..., 0])
    if len(xs) == 0:
        return []
    return [int(xs.min()), int(ys.min()), int(xs.max() - xs.min()), int(ys.max() - ys.min())]


def apply(src_map, resized, fill):
    out = np.full(src_map.shape + (3,), fill, dtype=np.uint8)
    ok = src_map >= 0
    out[ok] = resized.reshape(-1, 3)[src_map[ok]]
    return out


def resize_linear_u8(img, dw, dh):
    """OpenCV INTER_LINEAR for uint8 in its fixed-point form (see csrc/sector.hip)."""
    sh, sw = img.shape[:2]

    def coef(n, scale, ssize):
        f = ((np.arange(n) + 0.5) * scale - 0.5).astype(np.float32)             # scale stays a double (cv::hal::resize)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= ssize - 1
        f[hi], s[hi] = 0, ssize - 1
        s1 = np.minimum(s + 1, ssize - 1)
        return s, s1, np.rint((1 - f) * np.float32(2048)).astype(np.int64), np.rint(f * np.float32(2048)).astype(np.int64)

    x0, x1, ax0, ax1 = coef(dw, 1.0 / (dw / sw), sw)                             # inv_scale = dsize/ssize, scale = 1./inv_scale
    y0, y1, by0, by1 = coef(dh, 1.0 / (dh / sh), sh)
    im = img.astype(np.int64)
    h = im[:, x0] * ax0[None, :, None] + im[:, x1] * ax1[None, :, None]
    v = (((by0[:, None, None] * (h[y0] >> 4)) >> 16) + ((by1[:, None, None] * (h[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def sector_distort(image, mask, theta_deg=60, custom_rows=None):
    """Full path: -> (new_image, new_bbox) as the reference returns them."""
    h, w = image.shape[:2]
    src, box, T = winner_map(theta_deg, h, w, custom_rows)
    img_r = resize_linear_u8(image, N_ANG, T)
    msk_r = resize_linear_u8(mask, N_ANG, T)
    return apply(src, img_r, 114), mask_bbox(apply(src, msk_r, 0))
