"""The oracle (CPU restatement) against the outputs of the reference's own modules (tests/golden)."""
import numpy as np
import pytest
import torch

from oracle import fpd_ref, hourglass_ref
from tests import _cases


def test_loss_small_matches_reference():
    g = np.load(_cases.GOLDEN + '/loss_small.npz')
    rng = np.random.RandomState(7)
    p = torch.from_numpy(rng.standard_normal((3, 17, 12, 9)).astype(np.float32)).requires_grad_(True)
    t = torch.from_numpy(rng.standard_normal((3, 17, 12, 9)).astype(np.float32))
    w = torch.from_numpy((rng.uniform(0, 1.5, (3, 17, 1)) * (rng.uniform(0, 1, (3, 17, 1)) < 0.8)).astype(np.float32))
    l = fpd_ref.joints_mse_loss(p, t, w)
    l.backward()
    assert abs(l.item() - float(g['loss'])) < 1e-6
    np.testing.assert_allclose(p.grad.numpy(), g['grad'], atol=1e-8)
    cf = fpd_ref.joints_mse_closed_form(p.detach(), t, w)
    assert abs(cf.item() - float(g['loss'])) < 1e-6


@pytest.mark.parametrize('name', ['tiny', 'cfg1'])
def test_calibration_reproduces(name):
    c = _cases.CONFIGS[name]
    g = _cases.load_golden(name)
    calib = [fpd_ref.synth_batch(200 + i, c['batch'], c['joints'], c['image'], c['heat'])[0] for i in range(2)]
    sd = fpd_ref.synth_state_dict(hourglass_ref.hourglass_keys(c['s'][0], c['s'][1], c['joints']), 1)
    fpd_ref.calibrate_bn(sd, c['s'][1], calib)
    for k in sd:
        if 'running' in k:
            np.testing.assert_allclose(sd[k].numpy(), g['s_calib/' + k], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('name', ['tiny', 'cfg1'])
def test_fpd_step_matches_reference(name):
    c = _cases.CONFIGS[name]
    g = _cases.load_golden(name)
    s_sd, t_sd = _cases.state_dicts(name, g)
    x, tg, tw = _cases.batch(name, 0)
    adam = {}
    r = fpd_ref.fpd_step(s_sd, t_sd, c['s'][1], c['t'][1], x, tg, tw, 0.5, adam_state=adam)
    np.testing.assert_allclose(r['toutput'].numpy(), g['toutput'], atol=2e-5)
    for i, o in enumerate(r['outputs']):
        np.testing.assert_allclose(o.numpy(), g['output%d' % i], atol=2e-5)
    assert abs(r['pose'].item() - float(g['pose'])) < 1e-6
    assert abs(r['kd'].item() - float(g['kd'])) < 1e-6
    assert abs(r['loss'].item() - float(g['loss'])) < 1e-6
    names = fpd_ref.param_names(s_sd)
    flat = torch.cat([r['grads'][k].reshape(-1) for k in names])
    stride = int(g['grad_stride'])
    np.testing.assert_allclose(flat[::stride].numpy(), g['grad_flat'], atol=1e-6, rtol=1e-3)
    for k in s_sd:
        if 'running' in k:
            np.testing.assert_allclose(s_sd[k].numpy(), g['s_after/' + k], rtol=1e-4, atol=1e-5)
    # 3-step Adam trajectory (loss values; raw post-step weights of zero-gradient biases are noise-driven)
    traj = [[r['pose'].item(), r['kd'].item(), r['loss'].item()]]
    for step in (1, 2):
        x, tg, tw = _cases.batch(name, step)
        r = fpd_ref.fpd_step(s_sd, t_sd, c['s'][1], c['t'][1], x, tg, tw, 0.5, adam_state=adam)
        traj.append([r['pose'].item(), r['kd'].item(), r['loss'].item()])
    np.testing.assert_allclose(np.array(traj), g['traj'], rtol=5e-3, atol=1e-4)


def test_generate_target_properties():
    # centre value 1, 13x13 support for sigma=2, clipped at the border, weight 0 when fully outside
    xy = np.array([[128.0, 128.0], [1.0, 1.0], [-100.0, 50.0], [255.9, 255.9]])
    t, w = fpd_ref.generate_target(xy, np.ones(4), (256, 256), (64, 64), 2)
    assert t[0, 32, 32] == 1.0 and (t[0] > 0).sum() == 13 * 13
    assert t[1, 0, 0] == 1.0 and (t[1] > 0).sum() == 7 * 7
    assert w[2, 0] == 0 and t[2].sum() == 0
    assert w[3, 0] == 1 and t[3].max() < 1.0  # mu = 64 is outside the 64-wide map; tail only
