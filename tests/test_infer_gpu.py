"""validate / flip-test post-processing and the device data pipeline on the MI355X, through the C ABI, against the
REFERENCE's own functions (tests/golden/infer_ref.npz) and the numpy oracle (oracle/infer_ref.py).
Index / byte / exact-fp32 work: bit-identical (digests).  Image coordinates (float64 affine cast to float32): 2e-4 px."""
import os

import numpy as np
import pytest
import torch

from oracle import infer_ref
from tests import _cases
from tests import _cases_infer as CI

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'infer_ref.npz'))


class AD(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.parametrize('name', sorted(CI.POST_CASES))
def test_flip_merge_bit_identical(name):
    from fpd_amd.lib.utils import transforms as T
    seed, b, j, h, w, pairs, cdt = CI.POST_CASES[name]
    a, bf = CI.heatmaps(seed, b, j, h, w), CI.heatmaps(seed + 50, b, j, h, w)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(bf).cuda()
    for shift in (0, 1):
        m = T.flip_merge(ta, tb, pairs, shift).cpu().numpy()
        assert (CI.digest(m) == GOLD['%s/merged%d_sha' % (name, shift)]).all(), (name, shift)
        assert np.array_equal(m, infer_ref.flip_merge(a, bf, pairs, shift))
    fb = T.flip_back(tb, pairs).cpu().numpy()
    assert np.array_equal(fb, infer_ref.flip_back(bf.copy(), pairs))
    x = torch.randn(2, 3, 8, w, device='cuda')
    assert torch.equal(T.flip_input(x), torch.flip(x, dims=[3]))


@pytest.mark.parametrize('name', sorted(CI.POST_CASES))
def test_final_preds_match_reference(name):
    from fpd_amd.lib.core import inference as I
    seed, b, j, h, w, pairs, cdt = CI.POST_CASES[name]
    a, bf = CI.heatmaps(seed, b, j, h, w), CI.heatmaps(seed + 50, b, j, h, w)
    merged = infer_ref.flip_merge(a, bf, pairs, 1)
    c, s = CI.centers_scales(seed + 7, b, cdt)
    for pp in (0, 1):
        preds, maxvals = I.get_final_preds(AD(TEST=AD(POST_PROCESS=bool(pp))), torch.from_numpy(merged).cuda(), c, s)
        assert preds.dtype == np.float32 and preds.shape == (b, j, 2) and maxvals.shape == (b, j, 1)
        assert np.array_equal(maxvals, GOLD['%s/maxvals%d' % (name, pp)])
        np.testing.assert_allclose(preds, GOLD['%s/preds%d' % (name, pp)], rtol=0, atol=2e-4)
        # heat-map coordinates (arg-max + quarter-pixel shift): exact against the oracle
        coords, _, mv = I.final_preds_device(torch.from_numpy(merged).cuda(), None, pp)
        _, _, ocoords = infer_ref.get_final_preds(pp, merged.copy(), c, s)
        assert np.array_equal(coords.cpu().numpy(), ocoords)
    # numpy input is accepted like the reference's
    p2, _ = I.get_final_preds(AD(TEST=AD(POST_PROCESS=True)), merged, c, s)
    np.testing.assert_allclose(p2, GOLD['%s/preds1' % name], rtol=0, atol=2e-4)


@pytest.mark.parametrize('name', sorted(CI.TARGET_CASES))
def test_render_targets_bit_identical(name):
    from fpd_amd.lib.dataset import DevicePipeline
    seed, j, iw, ih, hw, hh, sigma = CI.TARGET_CASES[name]
    joints, vis = CI.target_inputs(seed, j, iw, ih, hw, sigma)
    pipe = DevicePipeline((iw, ih), (hw, hh), sigma, 'cuda')
    assert np.array_equal(pipe.g.cpu().numpy(), infer_ref.gaussian_patch(sigma).astype(np.float32))
    jv = np.zeros(joints.shape)
    jv[..., 0] = jv[..., 1] = vis
    tg, tw = pipe.generate_target(joints, jv)
    tg, tw = tg.cpu().numpy(), tw.cpu().numpy()
    assert (CI.digest(tg) == GOLD['tg_%s/target_sha' % name]).all()
    assert np.array_equal(tg[0], GOLD['tg_%s/target0' % name])
    assert np.array_equal(tw, GOLD['tg_%s/weight' % name])


def test_transform_joints_matches_reference():
    from fpd_amd.lib.dataset import DevicePipeline
    from fpd_amd.lib.utils import transforms as T
    t = T.get_affine_transform(GOLD['aff/center'], GOLD['aff/scale'], float(GOLD['aff/rot']), np.array([256, 256]))
    np.testing.assert_allclose(t, GOLD['aff/trans'], rtol=1e-12, atol=1e-9)
    pipe = DevicePipeline((256, 256), (64, 64), 2, 'cuda')
    pts = GOLD['aff/pts']
    joints = np.zeros((1, pts.shape[0], 3))
    joints[0, :, 0:2] = pts
    vis = np.ones((1, pts.shape[0]))
    vis[0, 3] = 0
    out = pipe.transform_joints(joints, vis, GOLD['aff/trans'][None]).numpy()
    exp = GOLD['aff/out'].copy()
    exp[3] = pts[3]                                   # invisible joints are left alone (JointsDataset.py:170-172)
    np.testing.assert_allclose(out[0, :, 0:2], exp, rtol=1e-13, atol=1e-10)


def test_warp_affine_equals_the_fixed_point_restatement():
    """cv2 is absent from the build container (parity unpinned, oracle/infer_ref.py): the kernel must agree bit for bit
    with the numpy restatement of OpenCV's fixed-point scheme, including ToTensor + Normalize in fp32."""
    from fpd_amd.lib.dataset import DevicePipeline
    from fpd_amd.lib.utils import transforms as T
    rng = np.random.RandomState(5)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    pipe = DevicePipeline((192, 256), (48, 64), 2, 'cuda', mean, std)
    imgs, trans = [], []
    for k, (h, w, c, s, r) in enumerate([(300, 420, (200.0, 150.0), 1.1, 0.0), (97, 61, (30.0, 50.0), 0.4, 33.0),
                                         (640, 480, (600.0, 20.0), 2.5, -71.0), (256, 192, (96.0, 128.0), 1.28, 0.0)]):
        imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        trans.append(T.get_affine_transform(np.array(c), np.array([s * 0.75, s]), r, [192, 256]))
    out = pipe.crop([torch.from_numpy(i).cuda() for i in imgs], np.stack(trans)).cpu().numpy()
    for k in range(len(imgs)):
        u8 = infer_ref.warp_affine_u8(imgs[k], infer_ref.invert_affine(trans[k]), 192, 256)
        exp = infer_ref.to_tensor_normalize(u8, mean, std)
        assert np.array_equal(out[k], exp), 'sample %d: max |diff| %.3e' % (k, np.abs(out[k] - exp).max())


class _ValSet:
    flip_pairs = CI.MPII_PAIRS

    def __init__(self, n):
        self.n = n
        self.got = None

    def __len__(self):
        return self.n

    def evaluate(self, cfg, preds, output_dir, all_boxes, img_path, *a, **kw):
        self.got = (preds.copy(), all_boxes.copy(), list(img_path))
        return {'Mean': 0.25, 'Mean@0.1': 0.5}, 0.25


def test_validate_loop_matches_reference():
    """core.function.validate (flip test + shift + post-processing on) around the HIP-backed HourglassNet, against the
    reference's loop body run on the reference's HourglassNet (tests/golden/make_golden_infer.py::loop_case)."""
    from fpd_amd.lib.core import function as F
    from fpd_amd.lib.core.loss import JointsMSELoss
    from fpd_amd.lib.models import hourglass
    from tests.test_model_gpu import make_cfg
    c = _cases.CONFIGS['tiny']
    _, t_sd = _cases.state_dicts('tiny')
    model = hourglass.get_pose_net(make_cfg(c['t'][0], c['t'][1], c['joints']), is_train=False)
    model.load_state_dict(t_sd, strict=True)
    model = model.cuda()
    x, tg, tw = _cases.batch('tiny', 0)
    center, scale = CI.centers_scales(3, x.shape[0], np.float64)
    meta = {'center': torch.from_numpy(center), 'scale': torch.from_numpy(scale), 'score': torch.tensor([0.9, 0.8]),
            'image': ['a.jpg', 'b.jpg']}
    cfg = AD(MODEL=AD(NUM_JOINTS=c['joints'], NAME='hourglass'), PRINT_FREQ=1,
             TEST=AD(FLIP_TEST=True, SHIFT_HEATMAP=True, POST_PROCESS=True))
    ds = _ValSet(x.shape[0])
    perf = F.validate(cfg, [(x, tg, tw, meta)], ds, model, JointsMSELoss(True).cuda(), '/tmp', '/tmp', None)
    assert perf == 0.25
    last = F.validate.last
    assert abs(last['loss'] - float(GOLD['loop/loss'])) <= 1e-5 * max(1.0, float(GOLD['loop/loss']))
    preds, boxes, paths = ds.got
    assert paths == ['a.jpg', 'b.jpg'] and preds.shape == (2, c['joints'], 3)
    assert np.array_equal(boxes[:, 0:2], center) and np.array_equal(boxes[:, 2:4], scale)
    np.testing.assert_allclose(boxes[:, 4], np.prod(scale * 200, 1))
    np.testing.assert_allclose(boxes[:, 5], [0.9, 0.8], atol=1e-7)
    # max values follow the heat-maps (fp32 network noise ~1e-4, tests/_cases.assert_parity); the arg-max may move where
    # two pixels are within that noise, so the coordinates are compared per joint with a small allowance
    np.testing.assert_allclose(preds[:, :, 2:3], GOLD['loop/maxvals'], rtol=0, atol=3e-4)
    d = np.abs(preds[:, :, 0:2] - GOLD['loop/preds']).max(axis=2)
    assert (d > 1e-3).sum() <= 2, 'image coordinates differ for %d of %d joints' % ((d > 1e-3).sum(), d.size)
    assert abs(last['acc'] - float(GOLD['loop/avg_acc'])) <= 1.0 / 8
    # without the flip test the output is the plain eval forward
    cfg2 = AD(MODEL=cfg.MODEL, PRINT_FREQ=1, TEST=AD(FLIP_TEST=False, SHIFT_HEATMAP=False, POST_PROCESS=False))
    F.validate(cfg2, [(x, tg, tw, meta)], ds, model, JointsMSELoss(True).cuda(), '/tmp', '/tmp', None)
    out = model(x.cuda())[-1].cpu().numpy()
    p2, mv2, _ = infer_ref.get_final_preds(False, out, center, scale)
    np.testing.assert_allclose(ds.got[0][:, :, 0:2], p2, rtol=0, atol=2e-4)
    assert np.array_equal(ds.got[0][:, :, 2:3], mv2)


def test_synthetic_dataset_validate_smoke():
    """tools/fpd_train.py's validation set: SyntheticPose.collate meta + evaluate() through validate()."""
    from fpd_amd.lib.config import _defaults
    from fpd_amd.lib.core import function as F
    from fpd_amd.lib.core.loss import JointsMSELoss
    from fpd_amd.lib.dataset import SyntheticPose
    from fpd_amd.lib.models import hourglass
    cfg = _defaults()
    cfg.MODEL.EXTRA.NUM_FEATURES, cfg.MODEL.EXTRA.NUM_STACKS = 32, 2
    cfg.MODEL.IMAGE_SIZE, cfg.MODEL.HEATMAP_SIZE = [128, 128], [32, 32]
    cfg.TEST.FLIP_TEST = cfg.TEST.SHIFT_HEATMAP = cfg.TEST.POST_PROCESS = True
    ds = SyntheticPose(cfg, 8, seed=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.collate)
    model = hourglass.get_pose_net(cfg, is_train=False).cuda()
    perf = F.validate(cfg, loader, ds, model, JointsMSELoss(True).cuda(), '/tmp', '/tmp', None)
    assert 0.0 <= perf <= 1.0 and np.isfinite(F.validate.last['loss'])
    assert np.isfinite(F.validate.last['all_preds']).all()
