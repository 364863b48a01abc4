"""Oracle sector warp vs the golden index maps generated from the reference (G8)."""
import zlib

import numpy as np
import pytest

from oracle import sector

CASES = [(t, h, w) for t in (30, 60, 90, 180) for (h, w) in ((427, 640), (640, 640), (1280, 1280))]


@pytest.mark.parametrize("theta,h,w", CASES)
def test_g8_winner_map(golden, theta, h, w):
    z = golden("g8_sector")
    key = "t%d_%dx%d_" % (theta, h, w)
    src, box, T = sector.winner_map(theta, h, w)
    assert T == int(z[key + "T"])
    assert list(src.shape) == list(z[key + "shape"][:2])
    assert np.array_equal(src[::4, ::4], z[key + "src_sub"])
    assert zlib.crc32(np.ascontiguousarray(src).tobytes()) == int(z[key + "src_crc"])
    assert int((src < 0).sum()) == int(z[key + "n_fill"])
    # mask rectangle -> bbox, through the same map
    r0, r1, c0, c1 = [int(v) for v in z[key + "mask_rect"]]
    mask = np.zeros((T, sector.N_ANG, 3), np.uint8)
    mask[r0:r1 + 1, c0:c1 + 1] = 255
    bbox = sector.mask_bbox(sector.apply(src, mask, 0))
    assert bbox == [int(v) for v in z[key + "bbox"]]


def test_theta_range_guard():
    with pytest.raises(AssertionError):
        sector.geometry(10, 640, 640)


def test_resize_identity_and_constant():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (7, 11, 3)).astype(np.uint8)
    assert np.array_equal(sector.resize_linear_u8(img, 11, 7), img)
    flat = np.full((5, 9, 3), 77, np.uint8)
    assert np.all(sector.resize_linear_u8(flat, 40, 13) == 77)
