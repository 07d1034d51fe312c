"""Seeded synthetic stand-in for the reference's MPII / COCO datasets (lib/dataset/mpii.py, coco.py need the image
archives, which are not available offline).  Samples have the structure of `JointsDataset.__getitem__`
(/root/reference/lib/dataset/JointsDataset.py:113-198): (input, target, target_weight, meta) with meta carrying
image / joints / joints_vis / center / scale / rotation / score, plus `flip_pairs` and an `evaluate()` with the signature
`core.function.validate` calls (function.py:298-301)."""
import numpy as np
import torch
import torch.utils.data

from ... import synth

MPII_FLIP_PAIRS = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]                                   # mpii.py:27
COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]                  # coco.py:73-74


class SyntheticPose(torch.utils.data.Dataset):
    POOL = 256          # distinct samples generated up front (one vectorised call); indices wrap around the pool

    def __init__(self, cfg, n, seed):
        self.n, self.seed = n, seed
        self.num_joints = cfg.MODEL.NUM_JOINTS
        self.image_size, self.heatmap_size = tuple(cfg.MODEL.IMAGE_SIZE), tuple(cfg.MODEL.HEATMAP_SIZE)
        self.sigma = cfg.MODEL.SIGMA
        self.flip_pairs = MPII_FLIP_PAIRS if self.num_joints == 16 else (COCO_FLIP_PAIRS if self.num_joints == 17 else [])
        self.pool = synth.make_batch(seed * 1000003, min(n, self.POOL), self.num_joints, self.image_size,
                                     self.heatmap_size, self.sigma)
        k = self.pool[0].shape[0]
        rng = np.random.RandomState(seed * 7919 + 1)
        # every crop is its own "image": the person box is the crop itself (scale*200 px = crop width), centred
        w, h = self.image_size
        self.center = np.tile(np.array([[w * 0.5, h * 0.5]]), (k, 1))
        self.scale = np.tile(np.array([[w / 200.0, h / 200.0]]), (k, 1))
        self.score = rng.uniform(0.5, 1.0, k)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i

    def collate(self, idx):
        """One gather per tensor instead of default_collate's stack of B samples (48 ms per batch of 32 on one core)."""
        x, t, w = self.pool
        idx = torch.as_tensor(idx)
        k = idx % x.shape[0]
        kn = k.numpy()
        meta = {'index': idx, 'image': ['synthetic/%d' % int(v) for v in idx], 'center': torch.from_numpy(self.center[kn]),
                'scale': torch.from_numpy(self.scale[kn]), 'score': torch.from_numpy(self.score[kn]),
                'rotation': torch.zeros(len(kn))}
        return x[k], t[k], w[k], meta

    def evaluate(self, cfg, preds, output_dir, all_boxes, img_path, *args, **kwargs):
        """(name_value, perf_indicator) like the reference datasets' evaluate (mpii.py:96-176): here PCK@0.5 of the
        predicted image coordinates against the arg-max of the synthetic targets mapped to image coordinates, threshold
        in units of a tenth of the crop size (no head boxes in synthetic data)."""
        _, t, w = self.pool
        n = preds.shape[0]
        k = np.arange(n) % t.shape[0]
        hm = t.numpy()[k]
        hw, hh = self.heatmap_size
        flat = hm.reshape(n, self.num_joints, -1)
        idx = flat.argmax(2)
        gt = np.stack([idx % hw, idx // hw], -1).astype(np.float64)
        sx, sy = self.image_size[0] / hw, self.image_size[1] / hh
        gt_img = (gt - np.array([hw * 0.5, hh * 0.5])) * np.array([sx, sx]) + self.center[k][:, None, :]
        vis = (w.numpy()[k][..., 0] > 0.5) & (flat.max(2) > 0)
        d = np.linalg.norm(preds[:, :, 0:2] - gt_img, axis=2) / (0.1 * self.image_size[1])
        pck = float(((d < 0.5) & vis).sum() / max(vis.sum(), 1))
        return {'PCK@0.5': pck}, pck
