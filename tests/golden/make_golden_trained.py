#!/usr/bin/env python
"""Generate tests/golden/trained_tiny.npz: the 'tiny' student / teacher pair TRAINED with the REFERENCE's own modules.

Run in the build container only (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trained.py

Why: the accuracy statement of the bf16 build needs networks whose activations are not the chaotic amplifier a random
initialisation is.  Pre-training on the device made the trained pair differ from run to run; this
script produces ONE committed pair instead: `/root/reference/lib/models/hourglass.py` HourglassNet (student hg2x32, teacher
hg3x64, the 'tiny' shapes of tests/_cases.py) + `/root/reference/lib/core/loss.py` JointsMSELoss + torch.optim.Adam
(lib/utils/utils.py:69-73), plain supervised training (the loop body of lib/core/function.py:28-96: per-stack loss sum,
zero_grad / backward / step) for STEPS iterations on the seeded, LEARNABLE synthetic batches of oracle.fpd_ref.blob_batch
(colour-coded joint blobs: the random crops of synth_batch carry no information about their joints), fp32 on the CPU.
Stored: both state_dicts (fp32, BN running statistics included) and the loss curves.
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

from oracle import fpd_ref, hourglass_ref  # noqa: E402
from tests._cases import CONFIGS  # noqa: E402

STEPS = 500
LR = 1e-3


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class AD(dict):
    __getattr__ = dict.__getitem__


def make_cfg(feats, stacks, joints):
    return AD(MODEL=AD(NUM_JOINTS=joints, EXTRA=AD(NUM_FEATURES=feats, NUM_STACKS=stacks, NUM_BLOCKS=1)))


def train(ref_hg, ref_loss, c, feats, stacks, seed, seed0):
    net = ref_hg.get_pose_net(make_cfg(feats, stacks, c['joints']), is_train=True)
    keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert keys == hourglass_ref.hourglass_keys(feats, stacks, c['joints'])
    net.load_state_dict(fpd_ref.synth_state_dict(keys, seed), strict=True)
    crit = ref_loss.JointsMSELoss(use_target_weight=True)
    opt = torch.optim.Adam(net.parameters(), lr=LR)
    net.train()
    curve = []
    for it in range(STEPS):
        x, g, w = fpd_ref.blob_batch(seed0 + it, c['batch'], c['joints'], c['image'], c['heat'])
        outputs = net(x)                              # lib/core/function.py:47-60: sum of the per-stack losses
        loss = crit(outputs[0], g, w)
        for o in outputs[1:]:
            loss = loss + crit(o, g, w)
        opt.zero_grad()
        loss.backward()
        opt.step()
        curve.append(loss.item())
    return net, curve


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_hg = load_by_path('ref_hourglass', os.path.join(REF, 'lib/models/hourglass.py'))
    ref_loss = load_by_path('ref_loss', os.path.join(REF, 'lib/core/loss.py'))
    c = CONFIGS['tiny']
    out = {}
    for tag, (feats, stacks), seed, seed0 in (('t', c['t'], 2, 5000), ('s', c['s'], 1, 7000)):
        net, curve = train(ref_hg, ref_loss, c, feats, stacks, seed, seed0)
        for k, v in net.state_dict().items():
            out['%s/%s' % (tag, k)] = v.detach().numpy().copy()
        out['%s_curve' % tag] = np.array(curve, np.float64)
        print(tag, 'loss %.5f -> %.5f after %d steps' % (curve[0], curve[-1], STEPS))
    out['steps'] = np.int64(STEPS)
    out['torch_version'] = np.array(torch.__version__)
    path = os.path.join(ROOT, 'tests', 'golden', 'trained_tiny.npz')
    np.savez_compressed(path, **out)
    print('->', path, '%.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
