"""Oracle: the input pipeline.  Test infrastructure only.

Restates ``preproc`` and ``TrainTransform.__call__`` (yolox_24p/datasets/data_augment.py:109-174): letterbox to the
top-left corner of a 114-filled canvas, HWC -> CHW fp32, labels from normalised to network-input coordinates, zero
padded to 50 rows.  ``cv2.resize(..., INTER_LINEAR)`` is ``oracle.sector.resize_linear_u8`` - OpenCV's published
fixed-point arithmetic; cv2 is not installed here, so that one step is "parity unpinned" (everything around it - r,
the resized size, padding, layout, the label arithmetic - is pinned by tests/golden/g14_input.npz, produced by the
reference functions themselves with this resize standing in for cv2's).
"""
import numpy as np

from .sector import resize_linear_u8


def preproc(img, input_size, swap=(2, 0, 1)):
    if len(img.shape) == 3:                                                           # :112-115
        padded = np.ones((input_size[0], input_size[1], 3), dtype=np.uint8) * 114
    else:
        padded = np.ones(input_size, dtype=np.uint8) * 114
    r = min(input_size[0] / img.shape[0], input_size[1] / img.shape[1])               # :117
    rw, rh = int(img.shape[1] * r), int(img.shape[0] * r)
    padded[:rh, :rw] = resize_linear_u8(img, rw, rh)                                  # :118-124
    out = np.ascontiguousarray(padded.transpose(swap), dtype=np.float32)
    return out, r, padded


def train_transform(image, targets, input_dim, max_labels=50):
    if targets.shape[1] == 0:                                                         # :141-144
        return preproc(image, input_dim)[0], np.zeros((max_labels, 51), dtype=np.float32)
    t = targets.copy()
    h, w, _ = image.shape
    boxes, cls = t[:, 1:], t[:, 0]
    boxes[:, 0::2] = boxes[:, 0::2] * w                                               # :155-156
    boxes[:, 1::2] = boxes[:, 1::2] * h
    image_t, r, _ = preproc(image, input_dim)
    boxes *= r                                                                        # :161
    rows = np.hstack((np.expand_dims(cls, 1), boxes))
    padded = np.zeros((max_labels, 51))
    padded[range(len(rows))[:max_labels]] = rows[:max_labels]
    return image_t, np.ascontiguousarray(padded, dtype=np.float32)
