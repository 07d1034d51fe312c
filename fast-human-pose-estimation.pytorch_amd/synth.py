"""Synthetic MPII/COCO-shaped batches for benchmarking and smoke tests (there is no dataset on the box).

Same recipe as SURVEY.md section 8(d): N(0,1) "ImageNet-normalised" crops, J joints uniform in the image,
Bernoulli(0.85) visibility, un-normalised sigma=2 Gaussian heat-map targets rendered the way
/root/reference/lib/dataset/JointsDataset.py:233-289 (generate_target) does: centre int(x/stride+0.5),
(6*sigma+1)^2 patch clipped at the border, weight 0 when the patch is fully outside."""
import numpy as np
import torch


def gaussian_targets(xy, vis, image_size, heatmap_size, sigma):
    """xy [B,J,2] pixels, vis [B,J] -> target [B,J,h,w] f32, target_weight [B,J,1] f32."""
    B, J = vis.shape
    wh, hh = heatmap_size
    rad = 3 * sigma
    size = 2 * rad + 1
    ax = np.arange(size, dtype=np.float32) - rad
    patch = np.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / (2.0 * sigma ** 2)).astype(np.float32)
    target = np.zeros((B, J, hh, wh), np.float32)
    weight = vis.astype(np.float32).copy()
    mu_x = (xy[..., 0] / (image_size[0] / wh) + 0.5).astype(np.int64)
    mu_y = (xy[..., 1] / (image_size[1] / hh) + 0.5).astype(np.int64)
    for b in range(B):
        for j in range(J):
            x0, y0 = mu_x[b, j] - rad, mu_y[b, j] - rad
            x1, y1 = x0 + size, y0 + size
            if x0 >= wh or y0 >= hh or x1 < 0 or y1 < 0:
                weight[b, j] = 0.0
                continue
            if weight[b, j] <= 0.5:
                continue
            cx0, cy0, cx1, cy1 = max(x0, 0), max(y0, 0), min(x1, wh), min(y1, hh)
            target[b, j, cy0:cy1, cx0:cx1] = patch[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0]
    return target, weight[..., None]


def make_batch(seed, batch, num_joints, image_size=(256, 256), heatmap_size=(64, 64), sigma=2, p_vis=0.85):
    """(input [B,3,H,W], target [B,J,h,w], target_weight [B,J,1]) as CPU float tensors."""
    rng = np.random.RandomState(seed)
    w, h = image_size
    inp = rng.standard_normal((batch, 3, h, w)).astype(np.float32)
    xy = np.stack([rng.uniform(0, w, (batch, num_joints)), rng.uniform(0, h, (batch, num_joints))], -1)
    vis = (rng.uniform(0, 1, (batch, num_joints)) < p_vis).astype(np.float32)
    tg, tw = gaussian_targets(xy, vis, image_size, heatmap_size, sigma)
    return torch.from_numpy(inp), torch.from_numpy(tg), torch.from_numpy(tw)
