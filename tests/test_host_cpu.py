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


def _accuracy_numpy(output, target, thr=0.5):
    """numpy restatement of /root/reference/lib/core/evaluate.py:16-71 + inference.py:18-46 (test oracle)."""
    def max_preds(hm):
        b, j, h, w = hm.shape
        flat = hm.reshape(b, j, -1)
        idx = flat.argmax(2)
        mv = flat.max(2)
        pr = np.stack([idx % w, np.floor(idx / w)], -1).astype(np.float32)
        pr *= (mv > 0)[..., None]
        return pr
    pred, gt = max_preds(output), max_preds(target)
    h, w = output.shape[2:]
    norm = np.array([w, h]) / 10.0
    b, j = pred.shape[:2]
    accs = []
    for jj in range(j):
        d = []
        for bb in range(b):
            if gt[bb, jj, 0] > 1 and gt[bb, jj, 1] > 1:
                d.append(np.linalg.norm(pred[bb, jj] / norm - gt[bb, jj] / norm))
        if d:
            accs.append(np.mean(np.array(d) < thr))
    return float(np.mean(accs)) if accs else 0.0, len(accs)


def test_accuracy_matches_reference_definition():
    from fpd_amd.lib.core.evaluate import accuracy
    rng = np.random.RandomState(0)
    _, tg, _ = fpd_ref.synth_batch(3, 4, 16)
    out = tg.numpy() + 0.3 * rng.standard_normal(tg.shape).astype(np.float32)
    _, avg, cnt, _ = accuracy(torch.from_numpy(out), tg)
    ravg, rcnt = _accuracy_numpy(out, tg.numpy())
    assert cnt == rcnt and abs(avg - ravg) < 1e-6


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


def test_hrnet_factory_fails_loudly():
    """MODEL.NAME pose_hrnet resolves (like the reference's eval('models.' + NAME + '.get_pose_net')) but the path is not
    built: a clear error, never a silent torch fallback."""
    from fpd_amd.lib import models
    from fpd_amd.runtime import FpdError
    with pytest.raises(FpdError, match='HRNet'):
        eval('models.pose_hrnet.get_pose_net')(None, is_train=True)
