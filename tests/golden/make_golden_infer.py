#!/usr/bin/env python
"""Golden vectors of the validate / flip-test post-processing and of the target rendering, produced by the REFERENCE's
own functions (imported from /root/reference; build container only -- the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_infer.py

  utils.transforms.flip_back / get_affine_transform / affine_transform / transform_preds   (lib/utils/transforms.py:15-102)
  core.inference.get_final_preds                                                          (lib/core/inference.py:49-79)
  dataset.JointsDataset.JointsDataset.generate_target                                     (lib/dataset/JointsDataset.py:233-289)
  the loop body of core.function.validate (lib/core/function.py:206-262) around the reference's HourglassNet in eval mode
  (restated here because validate() itself calls .cuda()), with FLIP_TEST / SHIFT_HEATMAP / POST_PROCESS on.

cv2 is not installed: the only cv2 function on these paths is getAffineTransform (3 point pairs -> 2x3 matrix), stubbed
with the 6x6 linear system OpenCV solves (LU, float64).  torchvision / json_tricks are stubbed as empty modules (imported
by files on the path, never called)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = '/root/reference/lib'
HERE = os.path.dirname(os.path.abspath(__file__))


def _get_affine_transform(src, dst):
    """cv::getAffineTransform: solve [x y 1 0 0 0; 0 0 0 x y 1] m = [X; Y] for the three point pairs (float64 LU)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    a, b = np.zeros((6, 6)), np.zeros(6)
    for i in range(3):
        a[i, 0:2], a[i, 2] = src[i], 1
        a[i + 3, 3:5], a[i + 3, 5] = src[i], 1
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


cv2 = types.ModuleType('cv2')
cv2.getAffineTransform = _get_affine_transform
sys.modules['cv2'] = cv2
for name in ('torchvision', 'torchvision.transforms', 'json_tricks'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, REF)
from core.evaluate import accuracy  # noqa: E402
from core.inference import get_final_preds  # noqa: E402
from utils.transforms import affine_transform, flip_back, get_affine_transform  # noqa: E402

from oracle import fpd_ref  # noqa: E402,F401
from tests import _cases  # noqa: E402

from tests._cases_infer import (COCO_PAIRS, MPII_PAIRS, POST_CASES, TARGET_CASES, centers_scales, digest, heatmaps,  # noqa: E402
                                target_inputs)


class AD(dict):
    __getattr__ = dict.__getitem__


def post_cases(res):
    for name, (seed, b, j, h, w, pairs, cdt) in POST_CASES.items():
        a, bf = heatmaps(seed, b, j, h, w), heatmaps(seed + 50, b, j, h, w)      # regenerated from the seeds by the tests
        for shift in (0, 1):
            f = flip_back(bf.copy(), pairs)
            f = torch.from_numpy(f.copy())
            if shift:
                f[:, :, :, 1:] = f.clone()[:, :, :, 0:-1]          # function.py:233-236
            merged = ((torch.from_numpy(a) + f) * 0.5).numpy()      # function.py:238
            res['%s/merged%d_sha' % (name, shift)] = digest(merged)      # bit-exactness pinned by digest;
            if name == 'small':                                           # the small case is stored in full
                res['%s/merged%d' % (name, shift)] = merged
        c, s = centers_scales(seed + 7, b, cdt)
        for pp in (0, 1):
            cfg = AD(TEST=AD(POST_PROCESS=bool(pp)))
            preds, maxvals = get_final_preds(cfg, merged.copy(), c, s)
            res['%s/preds%d' % (name, pp)], res['%s/maxvals%d' % (name, pp)] = preds, maxvals
        res[name + '/trans'] = np.stack([get_affine_transform(c[i], s[i], 0, [w, h], inv=1) for i in range(b)])


def target_cases(res):
    # by file path: the package __init__ pulls coco.py -> pycocotools (absent); JointsDataset.py itself needs cv2 (stub), torch
    spec = importlib.util.spec_from_file_location('ref_joints_dataset', os.path.join(REF, 'dataset', 'JointsDataset.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    JointsDataset = mod.JointsDataset
    for name, (seed, j, iw, ih, hw, hh, sigma) in TARGET_CASES.items():
        joints, vis = target_inputs(seed, j, iw, ih, hw, sigma)
        b = joints.shape[0]
        jv = np.zeros((b, j, 3))
        jv[..., 0] = jv[..., 1] = vis
        fake = types.SimpleNamespace(num_joints=j, target_type='gaussian', heatmap_size=np.array([hw, hh]),
                                     image_size=np.array([iw, ih]), sigma=sigma, use_different_joints_weight=False)
        tg, tw = zip(*[JointsDataset.generate_target(fake, joints[i], jv[i]) for i in range(b)])
        res['tg_%s/target_sha' % name] = digest(np.stack(tg))             # all samples, bit for bit
        res['tg_%s/weight' % name] = np.stack(tw)
        res['tg_%s/target0' % name] = np.stack(tg)[0]                     # one sample in full (sparse: compresses well)
    # affine_transform of joints (JointsDataset.py:170-172) under a rotated / scaled crop
    rng = np.random.RandomState(9)
    c, s, r = np.array([333.3, 251.7]), np.array([1.7, 1.7]), 23.0
    t = get_affine_transform(c, s, r, np.array([256, 256]))
    pts = rng.uniform(0, 600, (16, 2))
    res['aff/center'], res['aff/scale'], res['aff/rot'], res['aff/trans'], res['aff/pts'] = c, s, np.float64(r), t, pts
    res['aff/out'] = np.stack([affine_transform(p, t) for p in pts])


def loop_case(res):
    """validate()'s loop body (function.py:206-262) on the reference's HourglassNet ('tiny' teacher: hg3x64, eval mode)."""
    spec = importlib.util.spec_from_file_location('ref_hourglass', os.path.join(REF, 'models', 'hourglass.py'))
    ref_hg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_hg)
    spec = importlib.util.spec_from_file_location('ref_loss', os.path.join(REF, 'core', 'loss.py'))
    ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_loss)
    c = _cases.CONFIGS['tiny']
    _, t_sd = _cases.state_dicts('tiny')
    cfgm = AD(MODEL=AD(NUM_JOINTS=c['joints'], EXTRA=AD(NUM_FEATURES=c['t'][0], NUM_STACKS=c['t'][1], NUM_BLOCKS=1)))
    model = ref_hg.get_pose_net(cfgm, is_train=False)
    model.load_state_dict(t_sd, strict=True)
    model.eval()
    criterion = ref_loss.JointsMSELoss(use_target_weight=True)
    x, tg, tw = _cases.batch('tiny', 0)
    center, scale = centers_scales(3, x.shape[0], np.float64)
    cfg = AD(TEST=AD(POST_PROCESS=True))
    with torch.no_grad():
        output = model(x)[-1]
        input_flipped = torch.from_numpy(np.flip(x.numpy(), 3).copy())
        output_flipped = model(input_flipped)[-1]
        output_flipped = torch.from_numpy(flip_back(output_flipped.numpy(), MPII_PAIRS).copy())
        output_flipped[:, :, :, 1:] = output_flipped.clone()[:, :, :, 0:-1]
        output = (output + output_flipped) * 0.5
        loss = criterion(output, tg, tw)
        _, avg_acc, cnt, pred = accuracy(output.numpy(), tg.numpy())
        preds, maxvals = get_final_preds(cfg, output.clone().numpy(), center, scale)
    res['loop/output'], res['loop/loss'] = output.numpy(), np.float64(loss.item())
    res['loop/avg_acc'], res['loop/cnt'], res['loop/pred'] = np.float64(avg_acc), np.int64(cnt), pred
    res['loop/preds'], res['loop/maxvals'] = preds, maxvals
    res['loop/center'], res['loop/scale'] = center, scale


def main():
    res = {}
    post_cases(res)
    target_cases(res)
    loop_case(res)
    np.savez_compressed(os.path.join(HERE, 'infer_ref.npz'), **res)
    print('wrote infer_ref.npz: %d arrays, %.1f KB' % (len(res), os.path.getsize(os.path.join(HERE, 'infer_ref.npz')) / 1e3))
    print('loop: loss %.6f acc %.4f cnt %d' % (res['loop/loss'], res['loop/avg_acc'], res['loop/cnt']))


if __name__ == '__main__':
    main()
