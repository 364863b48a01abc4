"""Oracle label generation (oracle/labels24.py) against the reference's rotation_for_24p (G13), the hull
area against scipy, and the host-side pieces of the product (txt label format)."""
import numpy as np
import pytest

from oracle import labels24 as olab


def cases(golden):
    z = golden("g13_labels24")
    for tag in z["tags"]:
        tag = str(tag)
        H, W = [int(v) for v in z[tag + "_shape"]]
        mask = np.unpackbits(z[tag + "_mask"], axis=1)[:, :W].astype(np.uint8)
        assert mask.shape == (H, W)
        cx, cy = [float(v) for v in z[tag + "_centre"]]
        yield tag, mask, cx, cy, z[tag + "_pts"], z[tag + "_rad"]


def test_rays_vs_reference(golden):
    n = 0
    for tag, mask, cx, cy, pts, rad in cases(golden):
        got_p, got_r = olab.rotation_for_24p(cx, cy, mask)
        assert np.array_equal(got_p, pts), tag
        assert np.array_equal(got_r, rad), tag                    # doubles, bit for bit
        n += 1
    assert n == 6


def test_hull_area_vs_scipy():
    from scipy.spatial import ConvexHull
    g = np.random.RandomState(7)
    for _ in range(50):
        pts = g.randint(0, 400, size=(24, 2))
        assert abs(olab.hull_area(pts) - ConvexHull(pts).volume) < 1e-6
    assert olab.hull_area(np.array([[0, 0], [4, 0], [8, 0]])) == 0.0          # collinear
    assert olab.hull_area(np.array([[0, 0], [4, 0], [4, 3], [0, 3], [2, 1], [4, 0]])) == 12.0


def test_label_rows_and_filter(golden):
    for tag, mask, cx, cy, pts, rad in cases(golden):
        area = float(mask.sum())
        rows = olab.label_rows(3, cx, cy, mask, area)
        hull = olab.hull_area(pts)
        if hull <= 0.5 * area or hull >= 1.5 * area:
            assert rows is None, tag
            continue
        cord, radius = rows
        H, W = mask.shape
        assert cord.shape == (51,) and radius.shape == (27,) and cord[0] == 3
        assert np.allclose(cord[3::2] * W, pts[:, 0], atol=1e-3) and np.allclose(cord[4::2] * H, pts[:, 1], atol=1e-3)
        assert np.array_equal(radius[3:], rad / np.sqrt(H * H + W * W))


def test_txt_round_trip(tmp_path):
    from ep24 import labels24
    g = np.random.RandomState(3)
    rows = np.concatenate([g.randint(0, 80, (5, 1)).astype(np.float64), g.rand(5, 50)], 1)
    labels24.save_rows(tmp_path / "a.txt", rows)
    first = open(tmp_path / "a.txt").readline().split()
    assert len(first) == 51 and "." not in first[0] and all(len(v.split(".")[1]) == 4 for v in first[1:])
    back = labels24.load_rows(tmp_path / "a.txt")
    assert back.shape == (5, 51) and np.allclose(back, rows, atol=5e-5) and np.array_equal(back[:, 0], rows[:, 0])
    labels24.save_rows(tmp_path / "b.txt", rows[:1])
    assert labels24.load_rows(tmp_path / "b.txt").shape == (1, 51)
    labels24.save_rows(tmp_path / "c.txt", np.zeros((0, 51)))
    assert labels24.load_rows(tmp_path / "c.txt").size == 0


def test_dataset_script_surface():
    """The offline script keeps the reference's class surface; COCO ids map to the reference's contiguous indices
    (2+24_labels_create.py:36-51: '1'->0, '13'->11, '27'->24, '67'->60, '84'->73, '90'->79)."""
    import importlib.util
    import os
    import sys
    from conftest import PKG
    d = os.path.join(PKG, "yolox_24p")
    sys.path.insert(0, d)
    try:
        spec = importlib.util.spec_from_file_location("labels_create_24p", os.path.join(d, "datasets", "labels_create_24p.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(d)
    idx = {str(c): i for i, c in enumerate(mod.COCO_IDS)}
    assert len(idx) == 80 and [idx[k] for k in ("1", "13", "27", "67", "84", "90")] == [0, 11, 24, 60, 73, 79]
    for name in ("rotation_for_24p", "json_anno_process", "save_24r_to_txt", "load_label_json"):
        assert hasattr(mod.Polygon_24, name)
    try:
        import pycocotools  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="pycocotools"):
            mod.Polygon_24()
