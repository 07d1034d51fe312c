"""The entry points BASELINE.json's north_star names, driven end to end on the MI355X: `core.function.fpd_train` with the
reference's signature over a synthetic loader (fp32 parity build against the golden Adam trajectory; pipelined loop ==
un-pipelined steps; per-iteration meters; LR schedule reaching the device; checkpoint -> AUTO_RESUME round trip) and
`tools/fpd_train.py` as a subprocess.  Reference: tools/fpd_train.py:96-294, lib/core/function.py:99-187."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import _cases
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


class AD(dict):
    __getattr__ = dict.__getitem__


def _models(name='tiny', dtype='fp32'):
    from tests.test_model_gpu import build_models
    return build_models(name, dtype)


class _Loader:
    """Yields (input, target, target_weight, meta) like JointsDataset through a DataLoader."""

    def __init__(self, name, n, same_batch=True):
        self.b = [_cases.batch(name, 0 if same_batch else i) for i in range(n)]

    def __iter__(self):
        for x, t, w in self.b:
            yield x, t, w, {}

    def __len__(self):
        return len(self.b)


def _cfgnode(alpha=0.5, print_freq=2):
    return AD(KD=AD(ALPHA=alpha), PRINT_FREQ=print_freq, DEBUG=AD(DEBUG=False))


def test_fpd_train_follows_the_golden_adam_trajectory_and_feeds_meters_every_iteration():
    """3 iterations on the golden batch == the reference's 3-step Adam trajectory (tests/golden: losses of steps 0..2
    written by the reference loop body); every iteration reaches the loss/accuracy meters although only every second
    one prints; the device PCK entries equal the oracle metric on the maps of that iteration."""
    from fpd_amd.lib.core import function as F
    from fpd_amd.lib.core.loss import JointsMSELoss
    from fpd_amd.lib.utils.utils import FusedAdam
    c, gold, student, teacher = _models('tiny')
    opt = FusedAdam(student, lr=2.5e-4)                       # make_golden.py: Adam(lr=2.5e-4), batches 100, 101, 102
    crit = JointsMSELoss(True).cuda()
    fed = []
    orig = F.AverageMeter.update

    def spy(self, val, n=1):
        fed.append(val)
        return orig(self, val, n)
    F.AverageMeter.update = spy
    try:
        loss_avg = F.fpd_train(_cfgnode(), _Loader('tiny', 3, same_batch=False), student, teacher, crit, crit, opt, 0,
                               '/tmp', '/tmp', None)
    finally:
        F.AverageMeter.update = orig
    step = F.fused_step_for(student, teacher, opt, _cases.batch('tiny', 0)[0].shape, 0.5, 1, (True, True))
    assert int(opt.step_dev) == 3
    got = step.metric.log.view(-1, 4)[:3].cpu().numpy()      # {avg_acc, cnt, pose, kd} of iterations 0..2
    rows = np.stack([got[:, 2], got[:, 3], 0.5 * got[:, 2] + 0.5 * got[:, 3]], 1)
    assert abs(rows[0, 0] - float(gold['pose'])) < 2e-5 and abs(rows[0, 1] - float(gold['kd'])) < 2e-5
    _cases.assert_traj(rows, gold['traj'], 'tiny')           # the criterion of the fused-step trajectory test
    assert abs(loss_avg - float(rows[:, 2].mean())) < 1e-9   # the epoch average covers EVERY iteration (function.py:150-152)
    for v in rows[:, 2]:                                      # ... each of which reached a meter
        assert any(isinstance(f, float) and abs(f - v) < 1e-12 for f in fed), (v, fed)
    # PCK entries: the reference metric of the last student map of iteration 2 against its target (oracle restatement)
    from oracle import pck_ref
    o = step.student.output_view(c['s'][1] - 1).permute(0, 3, 1, 2).float().cpu().numpy()
    _, avg, cnt, _ = pck_ref.accuracy(o, _cases.batch('tiny', 2)[1].numpy())
    assert got[2, 1] == cnt and got[2, 0] == avg, (got[2], avg, cnt)


def test_pipelined_loop_equals_unpipelined_steps_and_lr_schedule_reaches_the_device():
    """fpd_train's loop (teacher one batch ahead on its own stream, two-slot staged teacher map, deferred readback) against
    plain un-pipelined FusedFPDStep.step() calls on 6 DISTINCT batches with lr = 0 (no parameter drift, so the two runs
    see identical weights; only BN running statistics move): the per-iteration pose / KD losses agree to 1e-6 relative
    -- a stale or wrong-slot teacher map would move the KD term by percents.  Then with lr > 0: parameters stay close
    (Adam turns noise-level gradients -- fp32 atomics order -- into +-lr updates), and a changed param_groups lr reaches
    the Adam kernel."""
    from fpd_amd import executor as E
    from fpd_amd.lib.core import function as F
    from fpd_amd.lib.core.loss import JointsMSELoss
    from fpd_amd.lib.utils.utils import FusedAdam
    n = 6
    loader = _Loader('tiny', n, same_batch=False)
    crit = JointsMSELoss(True).cuda()

    def both(lr):
        _, _, s1, t1 = _models('tiny')
        opt = FusedAdam(s1, lr=lr)
        F.fpd_train(_cfgnode(print_freq=100), loader, s1, t1, crit, crit, opt, 0, '/tmp', '/tmp', None)
        st1 = F.fused_step_for(s1, t1, opt, _cases.batch('tiny', 0)[0].shape, 0.5, 1, (True, True))
        pipelined = st1.metric.log.view(-1, 4)[:n, 2:4].cpu().numpy().copy()
        c, _, s2, t2 = _models('tiny')
        step = E.FusedFPDStep(s2.device_state(), s2.cfg_hg, t2.device_state(), t2.cfg_hg, c['batch'], c['image'][1],
                              c['image'][0], alpha=0.5, lr=lr)
        plain = []
        for x, tg, tw, _ in loader:
            step.set_batch(x, tg, tw)
            step.step()
            plain.append(step.losses()[:2])
        torch.cuda.synchronize()
        return pipelined, np.array(plain), s1, t1, opt, s2
    pipelined, plain, *_ = both(0.0)
    dev = np.abs(pipelined - plain) / np.abs(plain)
    assert dev.max() < 1e-6, (dev, pipelined, plain)
    assert np.abs(plain[1:, 1] - plain[:-1, 1]).min() > 1e-3 * plain[:, 1].mean()      # the batches really differ in their KD term
    pipelined, plain, s1, t1, opt, s2 = both(2.5e-4)
    assert (np.abs(pipelined - plain) / np.abs(plain)).max() < 5e-2
    p1, p2 = s1._flat['param'], s2._flat['param']
    assert float((p1 - p2).norm() / p2.norm()) < 2e-2
    # lr: zero it through param_groups (what the LR schedule / tools/fpd_train.py write) and run one more epoch
    torch.cuda.synchronize()
    before = p1.clone()
    opt.param_groups[0]['lr'] = 0.0
    F.fpd_train(_cfgnode(print_freq=100), _Loader('tiny', 1), s1, t1, crit, crit, opt, 1, '/tmp', '/tmp', None)
    torch.cuda.synchronize()
    assert float(opt.lr_dev) == 0.0 and torch.equal(before, s1._flat['param'])      # lr 0 on the device: no update


def test_unsupported_optimizer_or_criterion_is_an_error_not_ignored():
    from fpd_amd import runtime as R
    from fpd_amd.lib.core import function as F
    from fpd_amd.lib.core.loss import JointsMSELoss
    from fpd_amd.lib.utils.utils import FusedAdam
    c, gold, student, teacher = _models('tiny')
    crit = JointsMSELoss(True).cuda()
    sgd = torch.optim.SGD(student.parameters(), lr=0.1)
    with pytest.raises(R.FpdError):
        F.fpd_train(_cfgnode(), _Loader('tiny', 1), student, teacher, crit, crit, sgd, 0, '/tmp', '/tmp', None)
    with pytest.raises(R.FpdError):
        F.fpd_train(_cfgnode(), _Loader('tiny', 1), student, teacher, torch.nn.MSELoss(), crit, FusedAdam(student), 0, '/tmp', '/tmp', None)


def test_use_target_weight_flags_reach_the_fused_loss():
    """LOSS.USE_TARGET_WEIGHT false (loss.py:30-37: plain MSE) for one / both criteria against the oracle's per-joint loop."""
    from fpd_amd import executor as E
    from oracle import fpd_ref, hourglass_ref
    c, gold, student, teacher = _models('tiny')
    x, tg, tw = _cases.batch('tiny', 0)
    for flags in ((False, False), (True, False), (False, True)):
        step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, c['batch'],
                              c['image'][1], c['image'][0], alpha=0.5, use_target_weight=flags)
        step.set_batch(x, tg, tw)
        step.teacher_async(x)
        s = step.student
        torch.cuda.current_stream().wait_event(step.ev_t[0])
        s.run('prep'); s.run('fwd'); s.run('mid')
        pose, kd, _ = step.losses()
        outs = [s.output_view(i).permute(0, 3, 1, 2).float().cpu() for i in range(c['s'][1])]
        tmap = step.tmap[0].view(outs[0].shape[0], outs[0].shape[2], outs[0].shape[3], -1).permute(0, 3, 1, 2).float().cpu()
        rp = sum(float(fpd_ref.joints_mse_loss(o, tg, tw, flags[0])) for o in outs)
        rk = sum(float(fpd_ref.joints_mse_loss(o, tmap, tw, flags[1])) for o in outs)
        assert abs(pose - rp) < 1e-5 * max(1, abs(rp)) and abs(kd - rk) < 1e-5 * max(1, abs(rk)), (flags, pose, rp, kd, rk)


def test_teacher_chunks_match_single_chunk():
    """teacher_chunks: the frozen teacher cut into per-chunk graphs on their own streams (bf16 fused Bottlenecks read
    the folded BN tables the FIRST chunk's prep pass wrote) == the one-chunk teacher map, bit for bit."""
    from fpd_amd import executor as E
    c, gold, student, teacher = _models('tiny', 'bf16')
    x, tg, tw = _cases.batch('tiny', 0)
    maps = []
    for chunks in (1, 2):
        step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg, c['batch'],
                              c['image'][1], c['image'][0], alpha=0.5, teacher_chunks=chunks)
        assert len(step.teachers) == chunks
        step.set_batch(x, tg, tw)
        step.teacher_async(x)
        torch.cuda.synchronize()
        maps.append(step.tmap[0].float().cpu().clone())
    assert maps[0].abs().max() > 0 and torch.equal(maps[0], maps[1])


def test_plain_train_matches_the_oracle_without_a_teacher():
    """core.function.train (function.py:28-96): one iteration without distillation == the oracle loop with alpha 0."""
    from fpd_amd.lib.core import function as F
    from fpd_amd.lib.core.loss import JointsMSELoss
    from fpd_amd.lib.utils.utils import FusedAdam
    from oracle import fpd_ref
    c, gold, student, teacher = _models('tiny')
    s_sd, t_sd = _cases.state_dicts('tiny', gold)
    x, tg, tw = _cases.batch('tiny', 0)
    ref = fpd_ref.fpd_step({k: v.clone() for k, v in s_sd.items()}, t_sd, c['s'][1], c['t'][1], x, tg, tw, 0.0)
    opt = FusedAdam(student, lr=0.0)
    loss = F.train(_cfgnode(alpha=0.0), _Loader('tiny', 1), student, JointsMSELoss(True).cuda(), opt, 0, '/tmp', '/tmp', None)
    assert abs(loss - float(ref['pose'])) < 2e-5, (loss, float(ref['pose']))


def test_tools_fpd_train_cli_smoke_and_auto_resume(tmp_path):
    """`python tools/fpd_train.py --cfg ... --tcfg ... KEY VALUE` (tools/fpd_train.py:44-83) for two epochs of 3 iterations,
    then a second launch that AUTO_RESUMEs from checkpoint.pth (:224-234) and continues at epoch 2 with the schedule's lr."""
    cfgd = os.path.join(ROOT, 'experiments', 'fpd_synthetic')
    base = [sys.executable, os.path.join(ROOT, 'tools', 'fpd_train.py'), '--cfg', os.path.join(cfgd, 'hg4x128_student.yaml'),
            '--tcfg', os.path.join(cfgd, 'hg8x256_teacher.yaml'), '--max-iters', '3',
            'OUTPUT_DIR', str(tmp_path), 'MODEL.EXTRA.NUM_FEATURES', '32', 'MODEL.EXTRA.NUM_STACKS', '2', 'MODEL.IMAGE_SIZE', '128,128',
            'MODEL.HEATMAP_SIZE', '32,32', 'TRAIN.BATCH_SIZE_PER_GPU', '4', 'DATASET.NUM_SAMPLES', '32', 'PRINT_FREQ', '1',
            'TRAIN.LR_STEP', '[2,3]', 'AUTO_RESUME', 'True', 'MODEL.DTYPE', 'fp32']
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(base + ['TRAIN.END_EPOCH', '2'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    log = r.stdout + r.stderr
    assert 'Total Parameters' in log and log.count('\tPOSE_Loss') == 6 and 'epoch 1 done' in log
    import re
    last = [float(m) for m in re.findall(r'last logged loss ([0-9.eE+-]+)', log)]
    assert len(last) == 2 and all(0 < v < 10 for v in last), last                     # it trains on finite, sane losses
    ckpts = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f == 'checkpoint.pth']
    assert len(ckpts) == 1
    ck = torch.load(ckpts[0], map_location='cpu', weights_only=False)
    assert ck['epoch'] == 2 and ck['optimizer']['param_groups'][0]['initial_lr'] == 2.5e-4
    # epoch 1 ran at the scheduler's value for epoch 2 = first milestone (reference: scheduler stepped at epoch start)
    assert abs(ck['optimizer']['param_groups'][0]['lr'] - 2.5e-5) < 1e-12
    assert len(ck['optimizer']['state']) == len([k for k in ck['best_state_dict'] if 'running' not in k and 'tracked' not in k])
    r2 = subprocess.run(base + ['TRAIN.END_EPOCH', '3'], env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-3000:])
    log2 = r2.stdout + r2.stderr
    assert 'loaded checkpoint' in log2 and 'epoch 2 done' in log2 and 'epoch 0 done' not in log2
    ck2 = torch.load(ckpts[0], map_location='cpu', weights_only=False)
    assert ck2['epoch'] == 3 and abs(ck2['optimizer']['param_groups'][0]['lr'] - 2.5e-6) < 1e-13   # second milestone (3 <= 2+1)
    assert float(ck2['optimizer']['state'][0]['step']) == 9
    # teacher and student validated before the first epoch (tools/fpd_train.py:243-250), the student after every epoch
    # (:266-285), and the best model was kept
    assert log.count('Test: [0/') == 4 and 'PCK@0.5' in log
    assert log.index('Test: [0/') < log.index('Epoch: [0][0/')
    assert any(f == 'model_best.pth' for _, _, fs in os.walk(tmp_path) for f in fs)
    # tools/test.py (reference tools/test.py:38-135): validation of the trained model from <output dir>/final_state.pth, flip test on
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'test.py'), '--cfg', os.path.join(cfgd, 'hg4x128_student.yaml'),
                         'OUTPUT_DIR', str(tmp_path), 'MODEL.EXTRA.NUM_FEATURES', '32', 'MODEL.EXTRA.NUM_STACKS', '2',
                         'MODEL.IMAGE_SIZE', '128,128', 'MODEL.HEATMAP_SIZE', '32,32', 'TEST.BATCH_SIZE_PER_GPU', '8',
                         'DATASET.NUM_VALID_SAMPLES', '16', 'MODEL.DTYPE', 'fp32', 'TEST.FLIP_TEST', 'True', 'TEST.SHIFT_HEATMAP', 'True',
                         'TEST.POST_PROCESS', 'True', 'PRINT_FREQ', '1'], env=env, capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, (r3.stdout[-1500:], r3.stderr[-3000:])
    log3 = r3.stdout + r3.stderr
    assert log3.count('Test: [') == 2 and 'validation done' in log3 and 'PCK@0.5' in log3


def test_launch_log_keys_every_kernel_launch_on_its_plan_op_and_shape(tmp_path):
    """FPD_LAUNCH_LOG (csrc/common.h FPD_LAUNCH, csrc/api.hip set_op_tag): one line per kernel launch in host order --
    kernel expression, grid, block, plan op index, op kind + shape -- which tools/profile_summarize.py aligns with the
    rocprofv3 trace rows (same order) to key them on the tensor shape.  Run in a subprocess: the variable is read when the
    library is loaded."""
    import os
    import subprocess
    import sys
    log = tmp_path / 'launch.log'
    code = (
        "import torch\n"
        "from fpd_amd import executor as E\n"
        "from tests import _cases\n"
        "from tests.test_model_gpu import build_models\n"
        "c, gold, s, t = build_models('tiny', 'bf16')\n"
        "st = E.FusedFPDStep(s.device_state(), s.cfg_hg, t.device_state(), t.cfg_hg, c['batch'], c['image'][1], c['image'][0], alpha=0.5)\n"
        "st.set_batch(*_cases.batch('tiny', 0)); st.step(); torch.cuda.synchronize()\n"
        "print('LAUNCHES', st.launches_per_step()['total'])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPD_LAUNCH_LOG=str(log), PYTHONPATH=root)
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l.rstrip('\n').split('\t') for l in open(log)]
    assert len(lines) > 100 and all(len(f) == 5 for f in lines), lines[:3]
    kinds = {f[4].split(' ')[0] for f in lines}
    assert {'conv', 'ew', 'wgrad', 'loss', 'adam', 'stem_fwd', 'wprep'} <= kinds, kinds
    convs = [f for f in lines if f[4].startswith('conv ')]
    assert all(' N=2 ' in f[4] and ' C=' in f[4] and (' fwd' in f[4] or ' dgrad' in f[4]) for f in convs), convs[:3]
    # the lowering-time fusions leave their tag: every Bottleneck of the tiny student has BN-backward applies that the FOLD variants of
    # the halo-tile kernel evaluate (size-independent); '+wgrad' needs a persistent-kernel launch (>= 256 tiles), which tiny maps lack
    assert any('+fold' in f[4] for f in convs), sorted({f[4].split(' R=')[0] for f in convs})[:5]
    assert all(f[1].isdigit() and f[2].isdigit() and f[3].lstrip('-').isdigit() for f in lines)
    # plan ops and logged launches agree: memsets are not kernels of the library, adam is two launches (update + tick)
    n_ops = int(out.stdout.split('LAUNCHES')[1].split()[0])
    assert abs(len(lines) - n_ops) <= 16, (len(lines), n_ops)


def test_graft_entry_smoke_runs_in_a_fresh_process():
    """`__graft_entry__.smoke()` is what the driver runs on the box before the bench: one fused step of the 'tiny' case on cuda:0
    against the goldens and the fp64 oracle.  Run the way the driver does -- a fresh interpreter started at the repo root."""
    p = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.smoke()'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert 'smoke ok' in p.stdout, p.stdout[-2000:]
