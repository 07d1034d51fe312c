"""Seeded inputs of the validate / target-rendering cases -- shared by tests/golden/make_golden_infer.py (which feeds them
to the REFERENCE's functions) and the tests (which feed them to the oracle and to the HIP path)."""
import hashlib

import numpy as np

MPII_PAIRS = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]
COCO_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


def gaussian(h, w, cx, cy, sigma=2.0):
    y, x = np.mgrid[0:h, 0:w]
    return np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sigma ** 2)).astype(np.float32)


def heatmaps(seed, b, j, h, w):
    rng = np.random.RandomState(seed)
    out = np.zeros((b, j, h, w), np.float32)
    for n in range(b):
        for c in range(j):
            mode = rng.randint(0, 8)
            cx, cy = rng.randint(0, w), rng.randint(0, h)
            if mode == 0:                   # border peaks: no quarter-pixel shift (inference.py:62)
                cx = rng.choice([0, 1, w - 2, w - 1])
            if mode == 1:
                cy = rng.choice([0, 1, h - 2, h - 1])
            out[n, c] = gaussian(h, w, cx + 0.3 * rng.standard_normal(), cy + 0.3 * rng.standard_normal()) \
                + 0.01 * rng.standard_normal((h, w)).astype(np.float32)
            if mode == 2:                   # non-positive map: coordinates zeroed
                out[n, c] = -np.abs(out[n, c])
            if mode == 3:                   # exact tie + equal neighbours (sign(0) = 0)
                out[n, c] = 0
                out[n, c, h // 2, w // 2] = 1.0
                out[n, c, h // 2 + 3, w // 2 + 1] = 1.0
    return out


def centers_scales(seed, b, dtype):
    rng = np.random.RandomState(seed)
    c = rng.uniform(100, 900, (b, 2)).astype(dtype)
    s = rng.uniform(0.6, 3.5, (b, 1)).astype(dtype) * np.array([[1.0, 1.25]], dtype)
    return c, s



# name: (seed, batch, joints, h, w, flip pairs, dtype of center/scale as the dataset delivers them)
POST_CASES = {
    'sq64': (0, 3, 16, 64, 64, MPII_PAIRS, np.float64), 'coco64x48': (1, 2, 17, 64, 48, COCO_PAIRS, np.float32),
    'tall96x72': (2, 1, 17, 96, 72, COCO_PAIRS, np.float32), 'small': (3, 2, 5, 16, 12, [[0, 1], [2, 4]], np.float64),
}
# name: (seed, joints, image w, image h, heat-map w, heat-map h, sigma)
TARGET_CASES = {'mpii': (0, 16, 256, 256, 64, 64, 2), 'coco': (1, 17, 192, 256, 48, 64, 2), 'coco384': (2, 17, 288, 384, 72, 96, 3)}


def target_inputs(seed, j, iw, ih, hw, sigma, b=6):
    """joints [b,j,3] float64 in network-input pixels (some outside the crop), visibility [b,j] float64."""
    rng = np.random.RandomState(100 + seed)
    joints = np.zeros((b, j, 3))
    joints[..., 0] = rng.uniform(-40, iw + 40, (b, j))
    joints[..., 1] = rng.uniform(-40, ih + 40, (b, j))
    joints[0, :4, 0] = [-(3 * sigma) * iw / hw - 3, iw - 1, 0.0, iw + (3 * sigma) * iw / hw + 6]   # border-line cases
    vis = (rng.uniform(0, 1, (b, j)) < 0.8).astype(np.float64)
    return joints, vis


def digest(a):
    """sha256 of the array's bytes as a uint8 vector (bit-exact comparison without storing the array)."""
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()
