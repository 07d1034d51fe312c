"""oracle/hrnet_ref.py (functional restatement of the reference HRNet) against golden vectors produced by the reference's
own pose_hrnet.py + loss.py (tests/golden/make_golden_hrnet.py).  Test infrastructure for the HRNet rows of SURVEY.md
section 8(a); the product has no HRNet path yet."""
import os

import numpy as np
import torch

from oracle import fpd_ref, hrnet_ref
from tests._cases_hrnet import CONFIG, extra_cfg

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'hrnet_tiny.npz')


def _models():
    c = CONFIG
    s_cfg, t_cfg = extra_cfg(c['s']), extra_cfg(c['t'])
    s_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(s_cfg, c['joints']), 1)
    t_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(t_cfg, c['joints']), 2)
    return s_cfg, t_cfg, s_sd, t_sd


def test_hrnet_oracle_matches_reference_outputs():
    c = CONFIG
    gold = np.load(GOLD)
    s_cfg, t_cfg, s_sd, t_sd = _models()
    inp, tg, tw = fpd_ref.synth_batch(100, c['batch'], c['joints'], c['image'], c['heat'])
    names = [k for k in s_sd if s_sd[k].is_floating_point() and 'running' not in k]
    for k in names:
        s_sd[k].requires_grad_(True)
    with torch.no_grad():
        tout = hrnet_ref.hrnet_forward(t_sd, t_cfg, inp, train=False)
    out = hrnet_ref.hrnet_forward(s_sd, s_cfg, inp, train=True)
    pose, kd, loss = fpd_ref.fpd_losses([out], tout, tg, tw, c['alpha'])
    loss.backward()
    assert np.abs(tout.numpy() - gold['toutput']).max() < 1e-5
    assert np.abs(out.detach().numpy() - gold['output']).max() < 1e-5
    assert abs(pose.item() - float(gold['pose'])) < 1e-6 and abs(kd.item() - float(gold['kd'])) < 1e-6
    assert abs(loss.item() - float(gold['loss'])) < 1e-6
    grads = torch.cat([s_sd[k].grad.reshape(-1) for k in names])
    stride = int(gold['grad_stride'])
    ref = gold['grad_flat']
    assert grads[::stride].numel() == ref.size
    assert np.abs(grads[::stride].numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    assert abs(float(grads.double().norm()) - float(gold['grad_norm'])) < 1e-4 * float(gold['grad_norm'])
    for k in gold.files:                                   # BN running statistics after the train-mode forward
        if k.startswith('s_after/'):
            assert np.abs(s_sd[k[len('s_after/'):]].detach().numpy() - gold[k]).max() < 1e-6, k


def test_hrnet_key_lists_of_the_published_configs():
    """W32 / W48 (experiments/fpd_coco/hrnet/w{32,48}_*.yaml): parameter counts quoted in SURVEY.md section 8(a)."""
    for widths, nparam in (([32, 64, 128, 256], 28536113), ([48, 96, 192, 384], 63595745)):
        keys = hrnet_ref.hrnet_keys(extra_cfg(dict(widths=widths, blocks=4, modules=(1, 4, 3))), 17)
        n = sum(int(np.prod(s)) for k, s in keys if 'running' not in k and 'num_batches' not in k)
        assert n == nparam and len(keys) == 1754
