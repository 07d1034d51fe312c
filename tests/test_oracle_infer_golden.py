"""The inference / data-pipeline oracle (oracle/infer_ref.py) against the REFERENCE's own functions
(tests/golden/infer_ref.npz, written by tests/golden/make_golden_infer.py importing /root/reference)."""
import os

import numpy as np

from oracle import infer_ref
from tests import _cases_infer as CI

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'infer_ref.npz'))


def test_flip_merge_is_bit_identical_to_the_reference():
    for name, (seed, b, j, h, w, pairs, cdt) in CI.POST_CASES.items():
        a, bf = CI.heatmaps(seed, b, j, h, w), CI.heatmaps(seed + 50, b, j, h, w)
        for shift in (0, 1):
            m = infer_ref.flip_merge(a, bf, pairs, shift)
            assert (CI.digest(m) == GOLD['%s/merged%d_sha' % (name, shift)]).all(), (name, shift)
            if name == 'small':
                assert np.array_equal(m, GOLD['small/merged%d' % shift])


def test_final_preds_match_the_reference():
    for name, (seed, b, j, h, w, pairs, cdt) in CI.POST_CASES.items():
        a, bf = CI.heatmaps(seed, b, j, h, w), CI.heatmaps(seed + 50, b, j, h, w)
        merged = infer_ref.flip_merge(a, bf, pairs, 1)
        c, s = CI.centers_scales(seed + 7, b, cdt)
        for pp in (0, 1):
            preds, maxvals, _ = infer_ref.get_final_preds(pp, merged.copy(), c, s)
            assert np.array_equal(maxvals, GOLD['%s/maxvals%d' % (name, pp)])
            # image coordinates: a float64 3-point solve cast to float32 (the reference solves OpenCV's 6x6 system)
            np.testing.assert_allclose(preds, GOLD['%s/preds%d' % (name, pp)], rtol=0, atol=2e-4)
        t = np.stack([infer_ref.get_affine_transform(c[i], s[i], 0, [w, h], inv=1) for i in range(b)])
        np.testing.assert_allclose(t, GOLD[name + '/trans'], rtol=1e-12, atol=1e-9)


def test_generate_target_is_bit_identical_to_the_reference():
    for name, (seed, j, iw, ih, hw, hh, sigma) in CI.TARGET_CASES.items():
        joints, vis = CI.target_inputs(seed, j, iw, ih, hw, sigma)
        jv = np.zeros(joints.shape)
        jv[..., 0] = jv[..., 1] = vis
        tg, tw = zip(*[infer_ref.generate_target(joints[i], jv[i], (iw, ih), (hw, hh), sigma) for i in range(joints.shape[0])])
        assert (CI.digest(np.stack(tg)) == GOLD['tg_%s/target_sha' % name]).all(), name
        assert np.array_equal(np.stack(tg)[0], GOLD['tg_%s/target0' % name])
        assert np.array_equal(np.stack(tw), GOLD['tg_%s/weight' % name])


def test_affine_transform_of_joints():
    t = infer_ref.get_affine_transform(GOLD['aff/center'], GOLD['aff/scale'], float(GOLD['aff/rot']), np.array([256, 256]))
    np.testing.assert_allclose(t, GOLD['aff/trans'], rtol=1e-12, atol=1e-9)
    out = np.stack([infer_ref.affine_transform(p, GOLD['aff/trans']) for p in GOLD['aff/pts']])
    np.testing.assert_allclose(out, GOLD['aff/out'], rtol=1e-13, atol=1e-10)


def test_warp_affine_restatement_properties():
    """cv2 is absent (parity unpinned, see oracle/infer_ref.py): the fixed-point warp must reproduce integer shifts exactly,
    stay within one grey level of a float64 bilinear interpolation, and give zeros outside the source."""
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    m = np.array([[1.0, 0.0, 5.0], [0.0, 1.0, -3.0]])                 # dst = src shifted by (+5, -3)
    out = infer_ref.warp_affine_u8(img, infer_ref.invert_affine(m), 53, 37)
    exp = np.zeros_like(img)
    exp[0:34, 5:53] = img[3:37, 0:48]
    assert np.array_equal(out, exp)
    m = infer_ref.get_affine_transform(np.array([26.0, 18.0]), np.array([0.2, 0.2]), 17.0, [64, 48])
    minv = infer_ref.invert_affine(m)
    out = infer_ref.warp_affine_u8(img, minv, 64, 48).astype(np.float64)
    ys, xs = np.mgrid[0:48, 0:64].astype(np.float64)
    sx = minv[0, 0] * xs + minv[0, 1] * ys + minv[0, 2]
    sy = minv[1, 0] * xs + minv[1, 1] * ys + minv[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < 37) & (xx >= 0) & (xx < 53)
        return img[np.clip(yy, 0, 36), np.clip(xx, 0, 52)].astype(np.float64) * ok[..., None]
    ref = tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) + tap(y0 + 1, x0) * (1 - fx) * fy + \
        tap(y0 + 1, x0 + 1) * fx * fy
    # coordinates are quantised to 1/32 pixel: up to |gradient| * 1/32 * sqrt(2) grey levels on white noise
    assert np.abs(out - ref).max() <= 255 * 2 / 32 + 1
    assert np.abs(out - ref).mean() < 2.5
