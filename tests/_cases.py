"""Shared seeded test cases -- MUST match tests/golden/make_golden.py."""
import os

import numpy as np
import torch

from oracle import fpd_ref, hourglass_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

CONFIGS = {
    'tiny': dict(s=(32, 2), t=(64, 3), joints=16, batch=2, image=(128, 128), heat=(32, 32)),
    'cfg1': dict(s=(64, 2), t=(64, 2), joints=16, batch=2, image=(256, 256), heat=(64, 64)),
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, 'fpd_%s.npz' % name))


def batch(name, step=0):
    c = CONFIGS[name]
    return fpd_ref.synth_batch(100 + step, c['batch'], c['joints'], c['image'], c['heat'])


def state_dicts(name, golden=None):
    """Synthetic student/teacher checkpoints with the calibrated BN stats stored in the golden file."""
    c = CONFIGS[name]
    golden = golden if golden is not None else load_golden(name)
    out = []
    for tag, (feats, stacks), seed in (('s', c['s'], 1), ('t', c['t'], 2)):
        sd = fpd_ref.synth_state_dict(hourglass_ref.hourglass_keys(feats, stacks, c['joints']), seed)
        for k in list(sd):
            if 'running' in k:
                sd[k] = torch.from_numpy(golden['%s_calib/%s' % (tag, k)].copy())
        out.append(sd)
    return out
