"""Deterministic synthetic inputs for the YOLOX-24p training path.

Host-side only (torch CPU generators); the tensors are moved to the GPU by the caller.
Layouts follow the reference's data contract:

* images  ``[B,3,S,S]`` fp32, 0..255, no mean/std  (yolox_24p/datasets/data_augment.py:109-128)
* labels  ``[B,50,51]`` fp32: col 0 class id, 1-2 centre (px), 3..50 = 24 x (x,y) vertices (px),
  15 degrees apart starting on the +x axis; valid rows are a prefix, padding rows are all zero
  (yolox_24p/datasets/data_augment.py:138-174, consumed at yolox_24p/models/losses.py:190,219-220)
* head outputs ``[B,A,107]`` fp32: 0-1 centre, 2-25 radii (decoded, px), 26 obj logit, 27.. class logits
  (yolox_24p/models/yolo_head_24p.py:212-237)
"""
import math

import torch

MAX_LABELS = 50
LABEL_COLS = 51
NUM_RAYS = 24
STRIDES = (8, 16, 32)


def _hw(size):
    return (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))


def make_images(batch, size=640, seed=1):
    """``size``: the side of a square input or (height, width)."""
    g = torch.Generator().manual_seed(seed)
    h, w = _hw(size)
    return torch.rand(batch, 3, h, w, generator=g) * 255.0


def make_labels(batch, num_gt=10, size=640, seed=2, star=False, num_classes=80):
    """Padded GT table.  ``num_gt`` may be an int or a per-image list (0 allowed)."""
    g = torch.Generator().manual_seed(seed)
    counts = [num_gt] * batch if isinstance(num_gt, int) else list(num_gt)
    assert len(counts) == batch and max(counts) <= MAX_LABELS
    labels = torch.zeros(batch, MAX_LABELS, LABEL_COLS)
    ang = torch.arange(NUM_RAYS, dtype=torch.float64) * (15.0 * math.pi / 180.0)
    cosa, sina = torch.cos(ang), torch.sin(ang)
    h, w = _hw(size)
    scale = min(h, w) / 640.0
    for b, n in enumerate(counts):
        for i in range(n):
            cls = int(torch.randint(0, num_classes, (1,), generator=g))
            cx, cy = (torch.rand(2, generator=g, dtype=torch.float64) * torch.tensor([w - 200.0, h - 200.0], dtype=torch.float64) + 100.0).tolist()
            r = (torch.rand(NUM_RAYS, generator=g, dtype=torch.float64) * 80.0 + 20.0) * scale
            if star:
                # non-convex: every other ray pulled in to 35 % of its length
                r = torch.where(torch.arange(NUM_RAYS) % 2 == 0, r, r * 0.35)
            row = labels[b, i]
            row[0] = float(cls)
            row[1] = cx
            row[2] = cy
            row[3::2] = (cx + r * cosa).float()
            row[4::2] = (cy + r * sina).float()
    return labels


def anchor_grid(size=640, strides=STRIDES):
    """x_shift, y_shift, stride per anchor: level-major, row-major (y then x) inside a level
    (yolox_24p/models/yolo_head_24p.py:222-230)."""
    xs, ys, ss = [], [], []
    h, w = _hw(size)
    for s in strides:
        yv, xv = torch.meshgrid(torch.arange(h // s), torch.arange(w // s), indexing="ij")
        xs.append(xv.reshape(-1).float())
        ys.append(yv.reshape(-1).float())
        ss.append(torch.full(((h // s) * (w // s),), float(s)))
    return torch.cat(xs), torch.cat(ys), torch.cat(ss)


def make_raw_head(batch, size=640, seed=3, num_classes=80):
    """Raw (undecoded) head outputs ``[B,A,27+C]``: t_xy ~ N(0,.5), t_r ~ N(1.4,.6), logits ~ N(-3,1.5)."""
    g = torch.Generator().manual_seed(seed)
    xs, _, _ = anchor_grid(size)
    a = xs.numel()
    raw = torch.empty(batch, a, 27 + num_classes)
    raw[..., 0:2] = torch.randn(batch, a, 2, generator=g) * 0.5
    raw[..., 2:26] = torch.randn(batch, a, 24, generator=g) * 0.6 + 1.4
    raw[..., 26:] = torch.randn(batch, a, 1 + num_classes, generator=g) * 1.5 - 3.0
    return raw


def decode_head(raw, size=640):
    """xy=(t+grid)*s, r=exp(t)*s, logits untouched (yolox_24p/models/yolo_head_24p.py:232-235)."""
    xs, ys, ss = anchor_grid(size)
    out = raw.clone()
    out[..., 0] = (raw[..., 0] + xs) * ss
    out[..., 1] = (raw[..., 1] + ys) * ss
    out[..., 2:26] = torch.exp(raw[..., 2:26]) * ss[:, None]
    return out


def outputs_train_tuple(outputs, size=640):
    """The 5-tuple ``YOLOXHead.forward(train=True)`` returns (yolo_head_24p.py:205-206)."""
    xs, ys, ss = anchor_grid(size)
    x_shifts, y_shifts, strides = [], [], []
    o = 0
    h, w = _hw(size)
    for s in STRIDES:
        n = (h // s) * (w // s)
        x_shifts.append(xs[o:o + n][None].to(outputs.device))
        y_shifts.append(ys[o:o + n][None].to(outputs.device))
        strides.append(ss[o:o + n][None].to(outputs.device))
        o += n
    return x_shifts, y_shifts, strides, outputs, []


def fill_state(model, seed=0):
    """Deterministic, name-keyed weights for models too large to ship in a fixture (the config-4 ResNet network): every
    state-dict entry is drawn from a generator seeded by its own name, so the reference module, the oracle and the ep24
    mirror - which share key names - receive identical values whatever their construction order."""
    import zlib
    sd = model.state_dict()
    new = {}
    for name in sorted(sd):
        t = sd[name]
        if not t.dtype.is_floating_point:
            new[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if t.dim() == 4:
            v = torch.randn(t.shape, generator=g) * math.sqrt(2.0 / (t.shape[1] * t.shape[2] * t.shape[3]))
        elif t.dim() == 2:
            v = torch.randn(t.shape, generator=g) * 0.01
        elif name.endswith("running_var") or (name.endswith("weight") and t.dim() == 1):
            v = torch.rand(t.shape, generator=g) + 0.5
        else:
            v = torch.randn(t.shape, generator=g) * 0.1
        new[name] = v.to(t.dtype)
    model.load_state_dict(new, strict=True)
    return model
