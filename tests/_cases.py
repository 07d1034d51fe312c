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


# ------------------------------------------------------------------------------------------------
# parity criterion
# ------------------------------------------------------------------------------------------------
_TRUTH = {}


def truth64(name):
    """fp64 evaluation of the oracle for config `name`: dict(toutput, outputs[list], pose, kd, grads{key}).
    Measured in this repo: the reference's own fp32 CPU result differs from this by 0.7e-4..2.2e-4 on the
    heat-maps (O(1) values), i.e. the north-star tolerance 1e-4 sits AT the fp32 noise floor of the network."""
    if name in _TRUTH:
        return _TRUTH[name]
    c = CONFIGS[name]
    s_sd, t_sd = state_dicts(name)
    s64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in s_sd.items()}
    t64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in t_sd.items()}
    x, tg, tw = batch(name, 0)
    r = fpd_ref.fpd_step(s64, t64, c['s'][1], c['t'][1], x.double(), tg.double(), tw.double(), 0.5)
    _TRUTH[name] = r
    return r


_TRAJ = {}


def traj64(name, steps=3):
    """fp64 oracle loss trajectory [[pose, kd, loss]] over `steps` Adam steps.  Adam's first update is
    lr*sign(g), so sign flips of noise-level gradients make fp32 trajectories diverge: measured here, the
    reference's own fp32 trajectory is 1.3e-3..1.5e-3 away from this one at steps 2-3 of 'tiny'."""
    if name in _TRAJ and len(_TRAJ[name]) >= steps:
        return _TRAJ[name][:steps]
    import numpy as np
    c = CONFIGS[name]
    s_sd, t_sd = state_dicts(name)
    s64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in s_sd.items()}
    t64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in t_sd.items()}
    adam, traj = {}, []
    for it in range(steps):
        x, tg, tw = batch(name, it)
        r = fpd_ref.fpd_step(s64, t64, c['s'][1], c['t'][1], x.double(), tg.double(), tw.double(), 0.5, adam_state=adam)
        traj.append([r['pose'].item(), r['kd'].item(), r['loss'].item()])
    _TRAJ[name] = np.array(traj)
    return _TRAJ[name]


def assert_traj(ours, gold32, name, slack=2.5, floor=1e-3):
    import numpy as np
    t64 = traj64(name, len(ours))
    ref_dev = np.abs(np.asarray(gold32) - t64).max()
    our_dev = np.abs(np.asarray(ours) - t64).max()
    assert our_dev <= max(floor, slack * ref_dev), 'trajectory: |ours-fp64| %.3e vs reference fp32 deviation %.3e' % (our_dev, ref_dev)
    return our_dev, ref_dev


def assert_parity(ours, gold32, truth, label, floor=5e-5, slack=1.5, atol=1e-4):
    """ours / gold32 / truth: numpy arrays of one quantity.
      (1) |ours - truth64| <= max(floor, slack * |ref32 - truth64|)   -- no less accurate than the reference
      (2) |ours - ref32|   <= atol + |ref32 - truth64|                -- north-star 1e-4 above the noise floor
    """
    import numpy as np
    ours = np.asarray(ours, np.float64)
    ref_err = float(np.abs(np.asarray(gold32, np.float64) - truth).max())
    our_err = float(np.abs(ours - truth).max())
    d = float(np.abs(ours - np.asarray(gold32, np.float64)).max())
    assert our_err <= max(floor, slack * ref_err), '%s: |ours-fp64| %.3e vs reference fp32 error %.3e' % (label, our_err, ref_err)
    assert d <= atol + ref_err, '%s: |ours-ref32| %.3e > %.1e + floor %.3e' % (label, d, atol, ref_err)
    return our_err, ref_err, d
