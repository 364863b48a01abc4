"""GPU parity of the label-generation kernels with the reference's rotation_for_24p (G13, bit for bit),
with the oracle on random instance masks, and of the hull-area filter / txt rows."""
import numpy as np
import pytest
import torch

from test_oracle_labels24 import cases

pytestmark = pytest.mark.gpu


def random_masks(n, seed):
    g = np.random.RandomState(seed)
    out = []
    for i in range(n):
        H, W = int(g.randint(40, 500)), int(g.randint(40, 660))
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.zeros((H, W), np.uint8)
        for _ in range(int(g.randint(1, 6))):
            cy, cx = g.uniform(0, H), g.uniform(0, W)
            ry, rx = g.uniform(3, H / 3), g.uniform(3, W / 3)
            m |= ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0).astype(np.uint8)
        if i % 5 == 0:
            m[:, : W // 7] = 0                                # cuts and holes
            m[H // 2: H // 2 + 3, :] = 0
        if not m.any():
            m[H // 2, W // 2] = 1
        ys, xs = np.nonzero(m)
        x0, y0 = float(xs.min()), float(ys.min())
        w, h = float(xs.max() - xs.min()) + g.uniform(0, 1), float(ys.max() - ys.min()) + g.uniform(0, 1)
        out.append((m, x0 + w / 2, y0 + h / 2))
    return out


def test_rays_vs_reference_golden(golden):
    from ep24 import labels24
    cs = list(cases(golden))
    pts, rad = labels24.rays_batch([c[1] for c in cs], [[c[2], c[3]] for c in cs])
    pts, rad = pts.cpu().numpy(), rad.cpu().numpy()
    for i, (tag, mask, cx, cy, want_p, want_r) in enumerate(cs):
        assert np.array_equal(pts[i], want_p), tag
        assert np.array_equal(rad[i], want_r), tag            # float64, bit for bit
    # the single-object drop-in has the reference's signature and return types
    p1, r1 = labels24.rotation_for_24p(cs[1][2], cs[1][3], cs[1][1])
    assert p1.shape == (24, 2) and r1.shape == (24,) and r1.dtype == np.float64
    assert np.array_equal(p1, cs[1][4]) and np.array_equal(r1, cs[1][5])


def test_rays_hull_and_rows_vs_oracle_on_random_masks():
    from ep24 import labels24
    from oracle import labels24 as olab
    objs = random_masks(40, seed=17)
    pts, rad = labels24.rays_batch([o[0] for o in objs], [[o[1], o[2]] for o in objs])
    hull = labels24.hull_areas(pts)
    pts_h, rad_h, hull_h = pts.cpu().numpy(), rad.cpu().numpy(), hull.cpu().numpy()
    areas, want_rows = [], []
    for i, (m, cx, cy) in enumerate(objs):
        wp, wr = olab.rotation_for_24p(cx, cy, m)
        assert np.array_equal(pts_h[i], wp), i
        assert np.array_equal(rad_h[i], wr), i
        assert hull_h[i] == olab.hull_area(wp), i
        areas.append(float(m.sum()))
        want_rows.append(olab.label_rows(i % 80, cx, cy, m, areas[-1]))
    keep, cord, radius = labels24.label_rows([i % 80 for i in range(40)], [[o[1], o[2]] for o in objs],
                                             [o[0].shape for o in objs], pts, rad, hull, areas)
    assert list(keep) == [w is not None for w in want_rows] and 0 < keep.sum() < 40
    kept = [w for w in want_rows if w is not None]
    assert cord.shape == (len(kept), 51) and radius.shape == (len(kept), 27)
    for j, (wc, wr) in enumerate(kept):
        assert np.array_equal(cord[j], wc) and np.array_equal(radius[j], wr)


def test_hull_area_degenerate_inputs():
    from ep24 import labels24
    p = torch.zeros(3, 24, 2, dtype=torch.int32, device="cuda:0")
    p[1, :, 0] = torch.arange(24, dtype=torch.int32)                       # collinear
    p[2, :4] = torch.tensor([[0, 0], [10, 0], [10, 5], [0, 5]], dtype=torch.int32)   # rectangle + 20 copies of (0,0)
    assert labels24.hull_areas(p).cpu().tolist() == [0.0, 0.0, 50.0]
