#!/usr/bin/env python
"""Generate tests/golden/hrnet_tiny.npz by running the REFERENCE's own HRNet module on CPU.

Run in the build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hrnet.py

Imports /root/reference/lib/models/pose_hrnet.py and lib/core/loss.py by file path, loads the deterministic synthetic
checkpoints of oracle.fpd_ref.synth_state_dict into two small `PoseHighResolutionNet`s (student / teacher) and records
what one FPD iteration (lib/core/function.py:114-147, single-tensor branch :127-134) produces: teacher map, student map,
pose / kd / total loss, a strided sample of the student gradients and the BN running statistics after the step.
Inputs and weights are regenerated from seeds by the tests, so only outputs are stored.  Groundwork for the HRNet rows of
SURVEY.md section 8(a): it pins oracle/hrnet_ref.py; there is no HRNet product path yet.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = '/root/reference'

from oracle import fpd_ref, hrnet_ref  # noqa: E402
from tests._cases_hrnet import CONFIG, extra_cfg  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class AD(dict):
    __getattr__ = dict.__getitem__


def to_ad(d):
    return AD({k: to_ad(v) if isinstance(v, dict) else v for k, v in d.items()})


def build(ref, extra, joints, seed):
    cfg = AD(MODEL=AD(NUM_JOINTS=joints, EXTRA=to_ad(dict(extra, PRETRAINED_LAYERS=['*'])), INIT_WEIGHTS=False, PRETRAINED=''))
    net = ref.PoseHighResolutionNet(cfg)
    keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert keys == hrnet_ref.hrnet_keys(extra, joints), 'oracle key list != reference'
    sd = fpd_ref.synth_state_dict(keys, seed)
    net.load_state_dict(sd, strict=True)
    return net, sd


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_by_path('ref_pose_hrnet', os.path.join(REF, 'lib/models/pose_hrnet.py'))
    ref_loss = load_by_path('ref_loss', os.path.join(REF, 'lib/core/loss.py'))
    c = CONFIG
    inp, tg, tw = fpd_ref.synth_batch(100, c['batch'], c['joints'], c['image'], c['heat'])
    student, _ = build(ref, extra_cfg(c['s']), c['joints'], 1)
    teacher, _ = build(ref, extra_cfg(c['t']), c['joints'], 2)
    crit = ref_loss.JointsMSELoss(use_target_weight=True)
    student.train()
    teacher.eval()
    output = student(inp)                      # function.py:119-120
    toutput = teacher(inp)
    assert not isinstance(output, list)
    pose = crit(output, tg, tw)                # :127-134, single-tensor branch
    kd = crit(output, toutput, tw)
    loss = (1 - c['alpha']) * pose + c['alpha'] * kd
    student.zero_grad()
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in student.parameters()])
    out = {'output': output.detach().numpy(), 'toutput': toutput.detach().numpy(), 'pose': np.float64(pose.item()),
           'kd': np.float64(kd.item()), 'loss': np.float64(loss.item()), 'grad_stride': np.int64(c['grad_stride']),
           'grad_flat': grads[::c['grad_stride']].numpy(), 'grad_norm': np.float64(grads.double().norm().item()),
           'torch_version': np.array(torch.__version__)}
    for k, v in student.state_dict().items():
        if 'running' in k and ('stage4.0.branches.0' in k or k.startswith('bn1.') or 'transition3' in k):
            out['s_after/' + k] = v.numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'hrnet_tiny.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes; loss', loss.item(), 'grad elems', grads.numel())


if __name__ == '__main__':
    main()
