"""GPU parity of the fisheye sector warp against the reference-generated index maps (G8) and the oracle."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [(t, h, w) for t in (30, 60, 90, 180) for (h, w) in ((427, 640), (640, 640), (1280, 1280))]


@pytest.fixture(scope="module")
def dist():
    from ep24.sector import Image_Distortion
    return Image_Distortion("cuda:0")


@pytest.mark.parametrize("theta,h,w", CASES)
def test_winner_map_bit_exact(golden, dist, theta, h, w):
    z = golden("g8_sector")
    key = "t%d_%dx%d_" % (theta, h, w)
    src = dist.source_index(theta, h, w).cpu().numpy()
    assert list(src.shape) == list(z[key + "shape"][:2])
    assert np.array_equal(src[::4, ::4], z[key + "src_sub"])
    assert zlib.crc32(np.ascontiguousarray(src).tobytes()) == int(z[key + "src_crc"])


@pytest.mark.parametrize("two_pass", [False, True])
@pytest.mark.parametrize("theta,h,w", [(60, 640, 640), (90, 427, 640), (30, 1280, 1280)])
def test_full_warp_vs_oracle(dist, theta, h, w, two_pass):
    """Both device forms - resize + gather as two kernels, and the fused one-pass warp - reproduce the oracle."""
    from oracle import sector as osec
    dist.two_pass = two_pass
    rng = np.random.RandomState(theta)
    image = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    mask = np.zeros((h, w, 3), np.uint8)
    mask[h // 4: h // 2, w // 3: w // 2] = 255
    got_img, got_box = dist.sector_distort(image, mask, Theta=theta)
    want_img, want_box = osec.sector_distort(image, mask, theta)
    assert got_img.shape == want_img.shape and got_img.dtype == np.uint8
    dist.two_pass = False
    assert np.array_equal(got_img, want_img)
    assert got_box == want_box


def test_guards(dist):
    with pytest.raises(AssertionError):
        dist.sector_distort(np.zeros((64, 64, 3), np.uint8), np.zeros((64, 64, 3), np.uint8), Theta=10)
    img, box = dist.sector_distort(np.zeros((64, 64, 3), np.uint8), np.zeros((64, 64, 3), np.uint8), Theta=60)
    assert box == [] and img.dtype == np.uint8


def test_batch_warp_and_fisheye_transform(dist):
    """distort_batch == sector_distort image by image (no host sync inside), and TrainTransform(fisheye=...) letterboxes the
    warped images: the on-GPU augmentation of BASELINE config 5 through the public API."""
    import torch
    from ep24 import input as ein
    from oracle import sector as osec
    rng = np.random.RandomState(7)
    images = [rng.randint(0, 256, (h, w, 3)).astype(np.uint8) for (h, w) in ((320, 320), (240, 320), (320, 320))]
    masks = [np.zeros_like(im) for im in images]
    for m in masks[:2]:
        m[m.shape[0] // 4: m.shape[0] // 2, m.shape[1] // 3: m.shape[1] // 2] = 255        # the third mask stays empty
    thetas = [40, 75, 60]
    dev = lambda a: torch.from_numpy(a).to("cuda:0")
    outs, mouts, boxes = dist.distort_batch([dev(i) for i in images], [dev(m) for m in masks], thetas)
    boxes = boxes.cpu().tolist()
    for im, mk, th, o, b in zip(images, masks, thetas, outs, boxes):
        want_img, want_box = osec.sector_distort(im, mk, th)
        assert np.array_equal(o.cpu().numpy(), want_img)
        assert ([b[0], b[1], b[2] - b[0], b[3] - b[1]] if b[2] >= 0 else []) == want_box
    assert boxes[2][2] == -1
    # the transform: same angles drawn from the same seed, then the ordinary letterbox of the warped images
    tt = ein.TrainTransform(50, fisheye=(30, 90), seed=3)
    ref_rng = np.random.RandomState(3)
    th2 = [int(ref_rng.randint(30, 91)) for _ in images]
    targets = [np.zeros((0, 51)) for _ in images]
    got, _labs = tt.batch(images, targets, (256, 256))
    warped = dist.distort_batch([dev(i) for i in images], None, th2)[0]
    want, _ = ein.preproc_batch(warped, (256, 256))
    assert got.shape == (3, 3, 256, 256) and torch.equal(got, want)
