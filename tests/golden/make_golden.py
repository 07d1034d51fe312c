#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports /root/reference/lib/models/hourglass.py and lib/core/loss.py by file path (both are
pure torch), loads the deterministic synthetic checkpoints from oracle.fpd_ref.synth_state_dict
into reference `HourglassNet`s, and records what one FPD iteration
(lib/core/function.py:114-147) produces: teacher map, per-stack student maps, pose / kd / total
loss, student gradients, BN running stats after the step, and a 3-step Adam loss trajectory.
Inputs and weights are regenerated from seeds by the tests, so only outputs are stored.
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


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class AD(dict):
    __getattr__ = dict.__getitem__


def make_cfg(feats, stacks, joints):
    return AD(MODEL=AD(NUM_JOINTS=joints, EXTRA=AD(NUM_FEATURES=feats, NUM_STACKS=stacks, NUM_BLOCKS=1)))


CONFIGS = {
    # name: (student F,S), (teacher F,S), J, batch, image (W,H), heatmap (W,H)
    'tiny': dict(s=(32, 2), t=(64, 3), joints=16, batch=2, image=(128, 128), heat=(32, 32)),
    'cfg1': dict(s=(64, 2), t=(64, 2), joints=16, batch=2, image=(256, 256), heat=(64, 64)),
}


def build(ref_hg, feats, stacks, joints, seed, calib_batches):
    net = ref_hg.get_pose_net(make_cfg(feats, stacks, joints), is_train=True)
    keys = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert keys == hourglass_ref.hourglass_keys(feats, stacks, joints), 'oracle key list != reference'
    sd = fpd_ref.synth_state_dict(keys, seed)
    fpd_ref.calibrate_bn(sd, stacks, calib_batches)
    net.load_state_dict(sd, strict=True)
    return net, sd


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_hg = load_by_path('ref_hourglass', os.path.join(REF, 'lib/models/hourglass.py'))
    ref_loss = load_by_path('ref_loss', os.path.join(REF, 'lib/core/loss.py'))
    alpha = 0.5
    for name, c in CONFIGS.items():
        inp, tg, tw = fpd_ref.synth_batch(100, c['batch'], c['joints'], c['image'], c['heat'])
        calib = [fpd_ref.synth_batch(200 + i, c['batch'], c['joints'], c['image'], c['heat'])[0]
                 for i in range(2)]
        student, s_sd = build(ref_hg, c['s'][0], c['s'][1], c['joints'], 1, calib)
        teacher, t_sd = build(ref_hg, c['t'][0], c['t'][1], c['joints'], 2, calib)
        out = {}
        for k, v in s_sd.items():
            if 'running' in k:
                out['s_calib/' + k] = v.numpy()
        for k, v in t_sd.items():
            if 'running' in k:
                out['t_calib/' + k] = v.numpy()
        crit = ref_loss.JointsMSELoss(use_target_weight=True)
        opt = torch.optim.Adam(student.parameters(), lr=2.5e-4)      # utils.py:69-73
        student.train()
        teacher.eval()
        traj = []
        for step in range(3):
            x, g, w = fpd_ref.synth_batch(100 + step, c['batch'], c['joints'], c['image'], c['heat'])
            # ---- lib/core/function.py:119-147, verbatim semantics (teacher NOT under no_grad) ----
            outputs = student(x)
            toutput = teacher(x)
            toutput = toutput[-1]
            pose = crit(outputs[0], g, w)
            kd = crit(outputs[0], toutput, w)
            for o in outputs[1:]:
                pose = pose + crit(o, g, w)
                kd = kd + crit(o, toutput, w)
            loss = (1 - alpha) * pose + alpha * kd
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out['toutput'] = toutput.detach().numpy()
                for i, o in enumerate(outputs):
                    out['output%d' % i] = o.detach().numpy()
                out['pose'] = np.float64(pose.item())
                out['kd'] = np.float64(kd.item())
                out['loss'] = np.float64(loss.item())
                flat = torch.cat([p.grad.reshape(-1) for p in student.parameters()])
                stride = 1 if name == 'tiny' else 5
                out['grad_stride'] = np.int64(stride)
                out['grad_flat'] = flat[::stride].numpy()
                out['grad_norms'] = np.array([p.grad.norm().item() for p in student.parameters()])
            opt.step()
            if step == 0:
                for k, v in student.state_dict().items():
                    if 'running' in k:
                        out['s_after/' + k] = v.numpy().copy()
                pflat = torch.cat([p.detach().reshape(-1) for p in student.parameters()])
                out['param_after_step1'] = pflat[::(1 if name == 'tiny' else 5)].numpy()
            traj.append([pose.item(), kd.item(), loss.item()])
        out['traj'] = np.array(traj, np.float64)
        out['torch_version'] = np.array(torch.__version__)
        path = os.path.join(ROOT, 'tests', 'golden', 'fpd_%s.npz' % name)
        np.savez_compressed(path, **out)
        print(name, 'pose %.6f kd %.6f loss %.6f' % tuple(traj[0]), 'traj', [t[2] for t in traj],
              '->', path, '%.1f KB' % (os.path.getsize(path) / 1024))

    # loss-only golden: JointsMSELoss on random maps incl. zero weights and w not in {0,1}
    rng = np.random.RandomState(7)
    p = torch.from_numpy(rng.standard_normal((3, 17, 12, 9)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((3, 17, 12, 9)).astype(np.float32))
    w = torch.from_numpy((rng.uniform(0, 1.5, (3, 17, 1)) * (rng.uniform(0, 1, (3, 17, 1)) < 0.8)).astype(np.float32))
    p.requires_grad_(True)
    crit = ref_loss.JointsMSELoss(use_target_weight=True)
    l = crit(p, g, w)
    l.backward()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'loss_small.npz'),
                        loss=np.float64(l.item()), grad=p.grad.numpy())
    print('loss_small', l.item())


if __name__ == '__main__':
    main()
