#!/usr/bin/env python
"""Golden vectors of the per-iteration PCK metric, produced by the REFERENCE's own lib/core/evaluate.py + inference.py
(imported from /root/reference with cv2 stubbed: inference.py pulls utils.transforms -> cv2, unused by these functions).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pck.py        (build container only)

Cases: square 64x64 J=16 and COCO-shaped 64x48 (h x w) J=17 maps; noisy predictions around rendered Gaussian targets,
exact ties (first maximum must win), all-non-positive maps (coordinates zeroed), targets in the top-left corner
(skipped), joints with no valid sample, and border-line distances on the non-square map where [h,w] vs [w,h] differ."""
import os
import sys
import types

import numpy as np

REF = '/root/reference/lib'
sys.dont_write_bytecode = True
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
sys.path.insert(0, REF)
from core.evaluate import accuracy  # noqa: E402
from core.inference import get_max_preds  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def gaussian(h, w, cx, cy, sigma=2.0):
    y, x = np.mgrid[0:h, 0:w]
    return np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sigma ** 2)).astype(np.float32)


def make_case(seed, b, j, h, w):
    rng = np.random.RandomState(seed)
    tg = np.zeros((b, j, h, w), np.float32)
    out = np.zeros((b, j, h, w), np.float32)
    for n in range(b):
        for c in range(j):
            cx, cy = rng.randint(0, w), rng.randint(0, h)
            mode = rng.randint(0, 10)
            if mode == 0:                       # invisible joint: zero target map (coordinates zeroed -> skipped)
                pass
            elif mode == 1:                     # corner target: x <= 1 or y <= 1 -> skipped
                tg[n, c] = gaussian(h, w, rng.randint(0, 2), cy)
            else:
                tg[n, c] = gaussian(h, w, cx, cy)
            # prediction: target peak displaced by an offset that straddles the 0.5 threshold in both axes
            dx, dy = rng.randint(-4, 5), rng.randint(-4, 5)
            out[n, c] = gaussian(h, w, np.clip(cx + dx, 0, w - 1), np.clip(cy + dy, 0, h - 1)) + \
                0.02 * rng.standard_normal((h, w)).astype(np.float32)
            if mode == 2:                       # exact tie: two equal maxima, the first (row-major) wins
                out[n, c] = 0
                out[n, c, min(cy + 1, h - 1), min(cx + 2, w - 1)] = 1.0
                out[n, c, cy, cx] = 1.0
            if mode == 3:                       # all-non-positive prediction
                out[n, c] = -np.abs(out[n, c])
    out[:, j - 1] = 0.5 * out[:, j - 1]
    tg[:, 0] = 0                                # a joint with no valid sample at all: acc = -1, not counted
    # stored as float16 (small fixture); every value is exactly representable, the test widens back to float32
    return out.astype(np.float16).astype(np.float32), tg.astype(np.float16).astype(np.float32)


def main():
    res = {}
    for name, (seed, b, j, h, w) in {'sq64': (0, 3, 16, 64, 64), 'coco64x48': (1, 3, 17, 64, 48),
                                     'tall96x72': (2, 1, 17, 96, 72), 'small': (3, 4, 5, 16, 12)}.items():
        out, tg = make_case(seed, b, j, h, w)
        acc, avg, cnt, pred = accuracy(out, tg)
        res[name + '/output'], res[name + '/target'] = out.astype(np.float16), tg.astype(np.float16)
        res[name + '/acc'], res[name + '/avg_acc'], res[name + '/cnt'] = acc, np.float64(avg), np.int64(cnt)
        res[name + '/pred'] = pred
        res[name + '/gt'] = get_max_preds(tg)[0]
    np.savez_compressed(os.path.join(HERE, 'pck_ref.npz'), **res)
    for k in sorted(res):
        if k.endswith('avg_acc') or k.endswith('cnt'):
            print(k, res[k])


if __name__ == '__main__':
    main()
