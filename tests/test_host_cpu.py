"""Host logic that needs no GPU: config shim, module tree / state_dict interop, metric, synthetic data, data-parallel
helpers over gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import fpd_ref, hourglass_ref
from tests.conftest import ROOT


class AD(dict):
    __getattr__ = dict.__getitem__


def _cfg(feats, stacks, joints, dtype='fp32'):
    return AD(MODEL=AD(NUM_JOINTS=joints, DTYPE=dtype, EXTRA=AD(NUM_FEATURES=feats, NUM_STACKS=stacks, NUM_BLOCKS=1)))


def test_config_merge_like_yacs():
    from fpd_amd.lib.config import cfg
    c = cfg.clone()
    c.merge_from_dict({'GPUS': '(0,1)', 'MODEL': {'EXTRA': {'NUM_FEATURES': 128}}})
    c.merge_from_list(['KD.ALPHA', '0.25', 'MODEL.IMAGE_SIZE', '256,192', 'KD.TRAIN_TYPE', 'FPD'])
    assert c.GPUS == (0, 1) and c.MODEL.EXTRA.NUM_FEATURES == 128 and c['MODEL']['EXTRA']['NUM_STACKS'] == 8
    assert c.KD.ALPHA == 0.25 and tuple(c.MODEL.IMAGE_SIZE) == (256, 192) and c.KD.TRAIN_TYPE == 'FPD'
    assert cfg.KD.ALPHA == 0.5                     # the clone is independent


def test_state_dict_interop_and_flat_arena():
    from fpd_amd.lib.models import hourglass

    class AD(dict):
        __getattr__ = dict.__getitem__
    m = hourglass.get_pose_net(AD(MODEL=AD(NUM_JOINTS=16, EXTRA=AD(NUM_FEATURES=32, NUM_STACKS=2, NUM_BLOCKS=1))), True)
    keys = hourglass_ref.hourglass_keys(32, 2, 16)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == keys
    sd = fpd_ref.synth_state_dict(keys, 5)
    m.load_state_dict({'module.' + k: v for k, v in sd.items()}, strict=False)      # wrong prefix: nothing loads
    from fpd_amd.lib.utils.utils import load_checkpoint
    load_checkpoint({'state_dict': {'module.' + k: v for k, v in sd.items()}}, m, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    b = m.table['hg.1.hg.0.0.0.conv2.weight']
    flat = m._flat['param'][b.off:b.off + b.numel].view(b.shape)
    assert torch.equal(flat, sd['hg.1.hg.0.0.0.conv2.weight'].permute(0, 2, 3, 1))   # K,R,S,C in memory
    n0 = sum(p.numel() for p in m.parameters())
    assert n0 == sum(int(np.prod(s)) for k, s in keys if 'running' not in k and 'tracked' not in k)
    # default init statistics (torch defaults the reference relies on)
    m.reset_parameters()
    w = m.state_dict()['layer1.0.conv2.weight']
    assert abs(w.abs().max().item() - 1 / np.sqrt(8 * 9)) < 0.02 and m.state_dict()['bn1.running_var'].eq(1).all()


def _pck_cases():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pck_ref.npz'))
    for name in sorted({k.split('/')[0] for k in g.files}):
        yield name, {k.split('/')[1]: g[k] for k in g.files if k.startswith(name + '/')}


def test_pck_oracle_and_product_match_the_reference_goldens():
    """tests/golden/pck_ref.npz was produced by the reference's own core.evaluate.accuracy / core.inference.get_max_preds
    (make_golden_pck.py).  The oracle restatement and the product's torch `accuracy` reproduce (acc, avg_acc, cnt, pred)
    bit for bit -- square and 64x48 / 96x72 maps (where [h,w]/10 vs [w,h]/10 differ), ties, non-positive maps."""
    from fpd_amd.lib.core.evaluate import accuracy
    from oracle import pck_ref
    seen_nonsquare_effect = False
    for name, c in _pck_cases():
        out, tg = c['output'].astype(np.float32), c['target'].astype(np.float32)
        acc, avg, cnt, pred = pck_ref.accuracy(out, tg)
        np.testing.assert_array_equal(acc, c['acc'], err_msg=name)
        assert avg == float(c['avg_acc']) and cnt == int(c['cnt']), name
        np.testing.assert_array_equal(pred, c['pred'], err_msg=name)
        acc_t, avg_t, cnt_t, pred_t = accuracy(torch.from_numpy(out), torch.from_numpy(tg))
        np.testing.assert_array_equal(acc_t.numpy(), c['acc'], err_msg=name)
        assert avg_t == float(c['avg_acc']) and cnt_t == int(c['cnt']), name
        np.testing.assert_array_equal(pred_t.numpy(), c['pred'], err_msg=name)
        if out.shape[2] != out.shape[3]:              # the quirk matters: the "corrected" [w,h] order gives another answer
            h, w = out.shape[2:]
            d = pck_ref.calc_dists(c['pred'], c['gt'], np.ones((out.shape[0], 2)) * np.array([w, h]) / 10)
            alt = [pck_ref.dist_acc(d[i]) for i in range(out.shape[1])]
            seen_nonsquare_effect |= not np.array_equal(np.array(alt), c['acc'][1:])
    assert seen_nonsquare_effect


def test_product_synth_matches_oracle_target_rendering():
    from fpd_amd import synth
    rng = np.random.RandomState(1)
    xy = np.stack([rng.uniform(-20, 276, (3, 16)), rng.uniform(-20, 276, (3, 16))], -1)
    vis = (rng.uniform(0, 1, (3, 16)) < 0.8).astype(np.float32)
    tg, tw = synth.gaussian_targets(xy, vis, (256, 256), (64, 64), 2)
    for b in range(3):
        t, w = fpd_ref.generate_target(xy[b], vis[b], (256, 256), (64, 64), 2)
        np.testing.assert_array_equal(tg[b], t)
        np.testing.assert_array_equal(tw[b], w)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fpd_amd import dist as fdist
    from tests import _cases
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    c = _cases.CONFIGS['tiny']
    s_sd, t_sd = _cases.state_dicts('tiny')
    x, tg, tw = _cases.batch('tiny')                       # global batch of 2 -> 1 sample per rank
    xs, tgs, tws = fdist.shard([x, tg, tw], rank, world)
    r = fpd_ref.fpd_step(s_sd, t_sd, c['s'][1], c['t'][1], xs, tgs, tws, 0.5)
    names = fpd_ref.param_names(s_sd)
    flat = torch.cat([r['grads'][k].reshape(-1) for k in names]) / world      # the loss kernel's grad_scale = 1/world
    wait = fdist.make_allreduce(dist)(flat)                                     # async hook used by FusedFPDStep
    wait()
    # broadcast_state: rank 1 starts from different weights and must end up with rank 0's
    class M:
        pass
    m = M()
    m._flat = {'param': torch.full((8,), float(rank)), 'rstat': torch.full((4,), float(rank)),
               'nbt': torch.full((2,), rank, dtype=torch.int64)}
    fdist.broadcast_state(dist, m)
    ok = all(float(v.float().sum()) == 0.0 for v in m._flat.values())
    if rank == 0:
        q.put((flat.numpy(), ok))
    else:
        q.put((None, ok))
    dist.destroy_process_group()


def test_data_parallel_gradient_is_mean_of_shard_gradients():
    """world_size 2 over gloo: all-reduced (1/world-scaled) shard gradients == mean of the per-shard gradients, i.e. the
    gradient of the global-batch mean loss with per-replica BN statistics (nn.DataParallel semantics)."""
    from tests import _cases
    world, port = 2, 29500 + os.getpid() % 500
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=300) for _ in ps]
    [p.join(60) for p in ps]
    assert all(r[1] for r in res)
    got = next(r[0] for r in res if r[0] is not None)
    c = _cases.CONFIGS['tiny']
    x, tg, tw = _cases.batch('tiny')
    ref = 0
    for r in range(world):
        s_sd, t_sd = _cases.state_dicts('tiny')
        g = fpd_ref.fpd_step(s_sd, t_sd, c['s'][1], c['t'][1], x[r:r + 1], tg[r:r + 1], tw[r:r + 1], 0.5)['grads']
        ref = ref + torch.cat([g[k].reshape(-1) for k in fpd_ref.param_names(s_sd)]) / world
    # same arithmetic, different CPU thread counts -> summation-order noise only (gradients here reach |g| ~ 40)
    np.testing.assert_allclose(got, ref.numpy(), rtol=1e-4, atol=2e-6 * float(ref.abs().max()))


def _hrnet_cfg(widths, blocks, modules, joints, init=False, pretrained=''):
    from fpd_amd.lib.config import CfgNode, _wrap
    from tests._cases_hrnet import extra_cfg
    extra = dict(extra_cfg(dict(widths=widths, blocks=blocks, modules=modules)), PRETRAINED_LAYERS=['*'])
    return _wrap({'MODEL': {'NAME': 'pose_hrnet', 'NUM_JOINTS': joints, 'INIT_WEIGHTS': init, 'PRETRAINED': pretrained, 'EXTRA': extra}})


def test_hrnet_module_api_keys_init_and_checkpoint_interop():
    """models.pose_hrnet.get_pose_net (resolved like the reference's eval('models.' + NAME + '.get_pose_net')): state_dict
    keys / shapes / order of W32 equal the reference's (oracle key list, asserted equal to the reference module's by
    tests/golden/make_golden_hrnet.py), typed leaves carry the reference's conv geometry, init_weights follows
    pose_hrnet.py:462-492, strict checkpoint load works, and a CPU tensor is an error (no eager fallback)."""
    import torch.nn as nn
    from fpd_amd.lib import models
    from fpd_amd.runtime import FpdError
    from oracle import hrnet_ref
    from tests._cases_hrnet import extra_cfg
    cfg = _hrnet_cfg([32, 64, 128, 256], 4, (1, 4, 3), 17)
    m = eval('models.' + cfg.MODEL.NAME + '.get_pose_net')(cfg, is_train=True)
    keys = hrnet_ref.hrnet_keys(extra_cfg(dict(widths=[32, 64, 128, 256], blocks=4, modules=(1, 4, 3))), 17)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == keys
    assert sum(p.numel() for p in m.parameters()) == 28536113                       # README / SURVEY section 8(a)
    mods = dict(m.named_modules())
    c1, f = mods['conv1'], mods['stage3.0.fuse_layers.2.0.1.0']
    assert isinstance(c1, nn.Conv2d) and c1.stride == (2, 2) and c1.padding == (1, 1) and c1.bias is None
    assert f.stride == (2, 2) and f.kernel_size == (3, 3) and isinstance(mods['stage3.0.fuse_layers.2.0.1.1'], nn.BatchNorm2d)
    assert mods['final_layer'].bias is not None and mods['final_layer'].kernel_size == (1, 1)
    assert mods['stage2.0.fuse_layers.0.1.0'].kernel_size == (1, 1) and mods['transition1.1.0.0'].stride == (2, 2)
    small = _hrnet_cfg([8, 16, 32, 64], 1, (1, 2, 1), 5, init=True)
    ms = models.pose_hrnet.get_pose_net(small, is_train=True)                       # INIT_WEIGHTS: N(0, 1e-3) convs, BN (1, 0)
    w = ms.state_dict()['stage2.0.branches.0.0.conv1.weight']
    assert 5e-4 < float(w.std()) < 2e-3 and float(ms.state_dict()['bn1.weight'].min()) == 1.0
    assert float(ms.state_dict()['final_layer.bias'].abs().max()) == 0.0
    sd = fpd_ref.synth_state_dict([(k, tuple(v.shape)) for k, v in ms.state_dict().items()], 3)
    ms.load_state_dict(sd, strict=True)
    b = ms.table['stage4.0.branches.3.0.conv2.weight']
    assert torch.equal(ms._flat['param'][b.off:b.off + b.numel].view(b.shape), sd['stage4.0.branches.3.0.conv2.weight'].permute(0, 2, 3, 1))
    with pytest.raises(FpdError):
        ms(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ValueError):
        models.pose_hrnet.get_pose_net(_hrnet_cfg([8, 16, 32, 64], 1, (1, 1, 1), 5, init=True, pretrained='/nonexistent.pth'), is_train=True)


def test_fused_adam_state_dict_interoperates_with_torch_adam_and_resumes():
    """checkpoint['optimizer'] round trip (tools/fpd_train.py:224-234 AUTO_RESUME): FusedAdam's state dict has
    torch.optim.Adam's layout (so the reference's Adam can load it and vice versa), keeps 'initial_lr', and the
    closed-form MultiStepLR reproduces the reference's schedule (scheduler stepped at the START of each epoch) on a
    fresh run and on a resumed one alike."""
    from fpd_amd.lib.models import hourglass
    from fpd_amd.lib.utils.utils import FusedAdam, multistep_lr
    m = hourglass.get_pose_net(_cfg(32, 1, 4), is_train=True)
    opt = FusedAdam(m, lr=1e-3)
    params = list(m.parameters())
    ref = torch.optim.Adam(params, lr=1e-3)
    gen = torch.Generator().manual_seed(0)
    for p in params:
        p.grad = torch.randn(p.shape, generator=gen)
    ref.step(); ref.step()
    sd_ref = ref.state_dict()
    sd_ref['param_groups'][0]['initial_lr'] = 1e-3
    sd_ref['param_groups'][0]['lr'] = 1e-4                       # a decayed lr, as a checkpoint written after a milestone has
    opt.load_state_dict(sd_ref)                                  # reference (torch Adam) checkpoint -> FusedAdam
    assert int(opt.step_dev) == 2 and opt.param_groups[0]['lr'] == 1e-4 and opt.param_groups[0]['initial_lr'] == 1e-3
    for (mv, vv), p in zip(zip(opt._views(opt.m), opt._views(opt.v)), params):
        torch.testing.assert_close(mv, ref.state[p]['exp_avg'], rtol=0, atol=0)
        torch.testing.assert_close(vv, ref.state[p]['exp_avg_sq'], rtol=0, atol=0)
    sd = opt.state_dict()                                        # FusedAdam checkpoint -> torch Adam
    ref2 = torch.optim.Adam(params, lr=5.0)
    ref2.load_state_dict(sd)
    assert ref2.param_groups[0]['lr'] == 1e-4 and ref2.param_groups[0]['initial_lr'] == 1e-3
    for p in params:
        torch.testing.assert_close(ref2.state[p]['exp_avg'], ref.state[p]['exp_avg'], rtol=0, atol=0)
        assert float(ref2.state[p]['step']) == 2
    opt2 = FusedAdam(m, lr=7.0)                                  # and back into a fresh FusedAdam (AUTO_RESUME)
    opt2.load_state_dict(torch.load(_roundtrip(sd), weights_only=False))
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v) and int(opt2.step_dev) == 2
    assert opt2.param_groups[0]['initial_lr'] == 1e-3
    # a fresh optimizer's state dict has no per-parameter state, like torch's
    assert FusedAdam(m, lr=1e-3).state_dict()['state'] == {}
    # the retired round-1 flat {m, v, step} checkpoints cannot be mapped onto the re-ordered arena: refused, not mis-assigned
    with pytest.raises(ValueError, match='flat'):
        opt2.load_state_dict({'m': opt.m.clone(), 'v': opt.v.clone(), 'step': torch.tensor(2), 'param_groups': sd['param_groups']})
    # schedule: reference = MultiStepLR(milestones, gamma, last_epoch=-1) stepped at the start of every epoch
    import warnings
    o = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=2.5e-4)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sch = torch.optim.lr_scheduler.MultiStepLR(o, [3, 5], 0.1, last_epoch=-1)
        for epoch in range(8):
            sch.step()                                           # tools/fpd_train.py:253
            assert abs(o.param_groups[0]['lr'] - multistep_lr(2.5e-4, [3, 5], 0.1, epoch)) < 1e-12, epoch
    assert multistep_lr(1.0, [3, 5], 0.1, 1) == 1.0 and abs(multistep_lr(1.0, [3, 5], 0.1, 2) - 0.1) < 1e-15


def _roundtrip(obj):
    import io
    buf = io.BytesIO()
    torch.save(obj, buf)
    buf.seek(0)
    return buf


def test_bench_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no torch.distributed environment re-executes itself under torch.distributed.run
    (one rank per GPU, 127.0.0.1 rendezvous) and rank 0 prints ONE JSON line with n_gpus = 2.  Here: gloo + a stub step
    (FPD_BENCH_STUB) -- the launch path, the bucketed all-reduce hook and the timing contract, no GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['FPD_BENCH_STUB'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['data'] == 'stub'
    assert out['config']['ranks'] == 2 and out['config']['allreduce_ok'] is True
    # a launcher/flag mismatch is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and 'started 1 ranks' in (r2.stderr + r2.stdout)


def test_module_tree_hooks_and_model_summary_match_the_reference():
    """SURVEY section 8(b) module-tree conventions: real leaf modules with the reference's classes/names, forward hooks
    on every non-container module fire in the reference's execution order (HourglassNet.shape_forward), and
    utils.get_model_summary reproduces the reference's text (tests/golden/summary_hg2x64.txt, written by the reference's
    own get_model_summary over its own hourglass.py)."""
    import torch.nn as nn
    from fpd_amd import runtime as R
    from fpd_amd.lib.models import hourglass
    from fpd_amd.lib.utils.utils import get_model_summary
    m = hourglass.get_pose_net(_cfg(64, 2, 16), is_train=True)
    names = dict(m.named_modules())
    assert isinstance(names['conv1'], nn.Conv2d) and names['conv1'].kernel_size == (7, 7) and names['conv1'].stride == (2, 2)
    assert isinstance(names['hg.1.hg.0.3.0.bn2'], nn.BatchNorm2d) and names['hg.1.hg.0.3.0.bn2'].momentum == 0.1
    assert isinstance(names['hg.0.upsample'], nn.Upsample) and isinstance(names['maxpool'], nn.MaxPool2d)
    assert m.fc[0][2] is m.relu                                     # _make_fc shares the model's ReLU
    w = names['layer1.0.conv2'].weight
    assert tuple(w.shape) == (16, 16, 3, 3) and not w.is_meta and w.data_ptr() >= m._flat['param'].data_ptr()
    with pytest.raises(R.FpdError):
        names['conv1'](torch.zeros(1, 3, 8, 8))                                 # no eager path hides behind the tree
    gold = open(os.path.join(ROOT, 'tests', 'golden', 'summary_hg2x64.txt')).read()
    terse, verbose = gold.split('\n=====VERBOSE=====\n')
    x = torch.rand(1, 3, 256, 256)
    assert get_model_summary(m, x) == terse
    assert get_model_summary(m, x, verbose=True) == verbose
    assert not any(mod._forward_hooks for mod in m.modules())                   # hooks removed again
