"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference's per-iteration training metric.

Follows /root/reference/lib/core/inference.py:18-46 (get_max_preds) and lib/core/evaluate.py:16-71 (calc_dists, dist_acc,
accuracy) line by line, quirks included: the normaliser is `[h, w] / 10` applied to `(x, y)` (evaluate.py:55), distances
are float64, targets whose arg-max lies at x <= 1 or y <= 1 are skipped (evaluate.py:22).  Pinned to the reference's own
functions by tests/golden/pck_ref.npz (tests/golden/make_golden_pck.py imports the reference with a stubbed cv2).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np


def get_max_preds(batch_heatmaps):
    """inference.py:18-46."""
    b, j = batch_heatmaps.shape[:2]
    width = batch_heatmaps.shape[3]
    flat = batch_heatmaps.reshape((b, j, -1))
    idx = np.argmax(flat, 2).reshape((b, j, 1))
    maxvals = np.amax(flat, 2).reshape((b, j, 1))
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % width
    preds[:, :, 1] = np.floor(preds[:, :, 1] / width)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def calc_dists(preds, target, normalize):
    """evaluate.py:16-29."""
    preds, target = preds.astype(np.float32), target.astype(np.float32)
    dists = np.zeros((preds.shape[1], preds.shape[0]))
    for n in range(preds.shape[0]):
        for c in range(preds.shape[1]):
            if target[n, c, 0] > 1 and target[n, c, 1] > 1:
                dists[c, n] = np.linalg.norm(preds[n, c, :] / normalize[n] - target[n, c, :] / normalize[n])
            else:
                dists[c, n] = -1
    return dists


def dist_acc(dists, thr=0.5):
    """evaluate.py:32-39."""
    cal = np.not_equal(dists, -1)
    n = cal.sum()
    return np.less(dists[cal], thr).sum() * 1.0 / n if n > 0 else -1


def accuracy(output, target, thr=0.5):
    """evaluate.py:42-71 (hm_type 'gaussian').  Returns (acc[J+1], avg_acc, cnt, pred)."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    dists = calc_dists(pred, tgt, norm)
    j = output.shape[1]
    acc = np.zeros(j + 1)
    avg, cnt = 0, 0
    for i in range(j):
        acc[i + 1] = dist_acc(dists[i], thr)
        if acc[i + 1] >= 0:
            avg += acc[i + 1]
            cnt += 1
    avg = avg / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg
    return acc, avg, cnt, pred
