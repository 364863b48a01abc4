"""Oracle input pipeline (oracle/input.py) against the reference's preproc / TrainTransform (G14)."""
import zlib

import numpy as np

from oracle import input as oin

INPUT_CASES = [("wide", 97, 131, (160, 160), 3), ("tall", 211, 120, (160, 192), 0), ("up", 48, 64, (160, 160), 60),
               ("exact", 160, 160, (160, 160), 1), ("vga", 480, 640, (640, 640), 7)]


def input_case(tag, h, w, k, seed):
    g = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3 + yy) % 256, (xx + yy * 5) % 256, g.randint(0, 256, (h, w))], -1).astype(np.uint8)
    targets = np.concatenate([g.randint(0, 80, (k, 1)).astype(np.float64), np.round(g.rand(k, 50), 4)], 1) if k else np.zeros((0, 0))
    return img, targets


def test_preproc_and_transform_vs_reference(golden):
    z = golden("g14_input")
    for i, (tag, h, w, size, k) in enumerate(INPUT_CASES):
        img, targets = input_case(tag, h, w, k, 140 + i)
        out, r, padded = oin.preproc(img, size)
        assert r == float(z[tag + "_r"]) and out.dtype == np.float32 and out.shape == (3,) + size
        assert zlib.crc32(out.tobytes()) == int(z[tag + "_crc"]), tag
        assert np.array_equal(out[:, ::7, ::5], z[tag + "_sub"])
        image_t, labels = oin.train_transform(img, targets, size)
        assert np.array_equal(image_t, out) and np.array_equal(labels, z[tag + "_labels"]) and labels.dtype == np.float32
        rh, rw = int(h * r), int(w * r)
        assert (out[:, rh:, :] == 114).all() and (out[:, :, rw:] == 114).all()


def test_resize_is_identity_at_equal_size_and_exact_on_constants():
    from oracle.sector import resize_linear_u8
    g = np.random.RandomState(1)
    img = g.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(resize_linear_u8(img, 53, 37), img)
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert (resize_linear_u8(flat, 91, 64) == 77).all()
