"""The data-parallel device path with world_size > 1, pinned on ONE GPU (SURVEY.md section 8(a) row `nn.DataParallel`,
8(e); /root/reference/tools/fpd_train.py:143,173,202): two replicas of the fused step, each on its shard of the batch with
`world_size=2` (per-replica BN statistics, the loss kernel folding 1/world into the gradient), their flat gradient arenas
summed on the device -- what the RCCL all-reduce does between real ranks -- and Adam applied on both.

Checked against the oracle's arithmetic (the mean of the per-shard gradients, tests/test_host_cpu.py builds the same over
gloo): summed gradient under tests/_cases.assert_parity with an fp64 referee, identical parameters on both replicas after
Adam, and -- for the bucketed layout -- that every gradient bucket's slice is FINAL at the plan op its `wait` names (a copy
taken on a side stream right behind that wait equals the slice after the whole backward), that the buckets tile the arena,
and that the per-bucket reduction equals the single one bit for bit."""
import copy

import numpy as np
import pytest
import torch

from oracle import fpd_ref
from tests import _cases

pytestmark = pytest.mark.gpu


class _Replica:
    """One rank's FusedFPDStep plus the hook that stands in for dist.make_allreduce on a single device."""

    def __init__(self, step, model):
        self.step, self.model = step, model
        self.grad = model.device_state().A.tensor('grad')
        self.snap = torch.zeros_like(self.grad)            # per-bucket copies taken at the buckets' completion points
        self.side = torch.cuda.Stream()
        self.buckets = None

    def hook(self, flat_grad, buckets=None):
        assert flat_grad.data_ptr() == self.grad.data_ptr()
        self.buckets = buckets
        for lo, hi, wait in buckets or []:
            with torch.cuda.stream(self.side):
                if wait is not None:
                    wait(self.side)                        # device-side wait for the plan op that completes this slice
                else:
                    self.side.wait_stream(torch.cuda.current_stream())
                self.snap[lo:hi].copy_(flat_grad[lo:hi])
        return lambda: None                                # the "collective" is performed by the test between the steps


def _two_replicas(make_models, batch, H, W, alpha, lr, world_size=2):
    """make_models() -> (student, teacher) with identical weights on every call."""
    from fpd_amd import executor as E
    reps = []
    for r in range(2):
        student, teacher = make_models()
        step = E.FusedFPDStep(student.device_state(), student.cfg_hg, teacher.device_state(), teacher.cfg_hg,
                              batch // 2, H, W, alpha=alpha, lr=lr, world_size=world_size)
        reps.append(_Replica(step, student))
    return reps


def _check_against_single_rank_runs(make_models, reps, total, x, tg, tw, batch, H, W, alpha, lr):
    """The SHARP check of the world > 1 arithmetic, free of rounding-noise tolerances: the same two shards run as two
    independent world_size = 1 steps give gradients g_0, g_1 of their local mean losses; the all-reduced gradient of the
    world_size = 2 replicas (the loss kernel scaled every shard's gradient by 1/2 -- a power of two, exact) must be
    (g_0 + g_1) / 2 up to the re-rounding of a few fp32 sums, and every rank must report the same local loss either way."""
    solo = _two_replicas(make_models, batch, H, W, alpha, lr, world_size=1)
    _run_shards(solo, x, tg, tw)
    mean = 0.5 * (solo[0].grad.double() + solo[1].grad.double())
    rel = float((total.double() - mean).norm() / mean.norm())
    assert rel < 1e-5, 'all-reduced world=2 gradient vs the mean of the single-rank shard gradients: relative L2 %.3e' % rel
    for a, b in zip(reps, solo):
        la, lb = a.step.losses(), b.step.losses()
        assert max(abs(u - v) for u, v in zip(la, lb)) < 1e-9, (la, lb)
    return rel


def _run_shards(reps, x, tg, tw):
    from fpd_amd import dist as fdist
    for r, rep in enumerate(reps):
        xs, tgs, tws = fdist.shard([x, tg, tw], r, 2)
        rep.step.set_batch(xs, tgs, tws)
        rep.step.teacher_async(xs)
        rep.step.student_step(rep.hook)                    # Adam is deferred until flush(), like with a real collective
    torch.cuda.synchronize()


def _check_buckets_and_reduce(reps):
    n = reps[0].grad.numel()
    for rep in reps:
        cover = sorted((lo, hi) for lo, hi, _ in rep.buckets)
        assert cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), cover
        # a slice copied right behind its bucket's completion point is what the slice holds after the whole backward
        assert torch.equal(rep.snap, rep.grad), 'a gradient bucket was still being written behind its completion point'
    total = reps[0].grad + reps[1].grad                    # the all-reduce (sum) of the whole arena ...
    bucketed = torch.empty_like(total)
    for lo, hi, _ in reps[0].buckets:                      # ... and bucket by bucket from the early copies
        bucketed[lo:hi] = reps[0].snap[lo:hi] + reps[1].snap[lo:hi]
    assert torch.equal(total, bucketed)
    for rep in reps:
        rep.grad.copy_(total)
    return total


def _flat_model_grads(model):
    model._attach_grads()
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()


def test_hourglass_two_shards_on_one_gpu_match_the_mean_of_shard_gradients():
    from tests.test_model_gpu import build_models
    name = 'tiny'
    c = _cases.CONFIGS[name]
    lr = 2.5e-4

    def make():
        _, _, s, t = build_models(name)
        return s, t
    # global batch 4 = two samples per shard: with ONE sample a shard's train-mode BN normalises the 2x2 maps at the bottom of
    # this small hourglass over 4 values, and the reference's own fp32 gradient is then 1.0 (of ~40) away from fp64
    GB = 4
    reps = _two_replicas(make, GB, c['image'][1], c['image'][0], 0.5, lr)
    assert len(reps[0].step.student.state.table.buckets) >= 2      # one bucket per stack: the bucketed layout is exercised
    x, tg, tw = fpd_ref.synth_batch(100, GB, c['joints'], c['image'], c['heat'])
    _run_shards(reps, x, tg, tw)
    total = _check_buckets_and_reduce(reps)
    _check_against_single_rank_runs(make, reps, total, x, tg, tw, GB, c['image'][1], c['image'][0], 0.5, lr)
    ours = _flat_model_grads(reps[0].model)
    # oracle: per-shard gradients of the per-shard mean loss, averaged over the shards (fp32 and fp64)
    keys = reps[0].model.table.trainable_keys()
    mean32, mean64, losses32, losses64 = 0, 0, [], []
    for r in range(2):
        s_sd, t_sd = _cases.state_dicts(name)
        g32 = fpd_ref.fpd_step(s_sd, t_sd, c['s'][1], c['t'][1], x[2 * r:2 * r + 2], tg[2 * r:2 * r + 2], tw[2 * r:2 * r + 2], 0.5)
        losses32.append((float(g32['pose']), float(g32['kd'])))
        mean32 = mean32 + torch.cat([g32['grads'][k].reshape(-1) for k in keys]) / 2
        s_sd, t_sd = _cases.state_dicts(name)
        s64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in s_sd.items()}
        t64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in t_sd.items()}
        g64 = fpd_ref.fpd_step(s64, t64, c['s'][1], c['t'][1], x[2 * r:2 * r + 2].double(), tg[2 * r:2 * r + 2].double(), tw[2 * r:2 * r + 2].double(), 0.5)
        mean64 = mean64 + torch.cat([g64['grads'][k].reshape(-1) for k in keys]) / 2
        losses64.append((float(g64['pose']), float(g64['kd'])))
    for r, rep in enumerate(reps):                         # each rank reports the loss of ITS shard (local mean)
        # a scalar loss is ONE realisation of the fp32 rounding noise: an absolute band; the gradient check below is the
        # statistically meaningful one (thousands of elements, max-norm, fp64 referee)
        for ours_l, r32, r64 in zip(rep.step.losses()[:2], losses32[r], losses64[r]):
            assert abs(ours_l - r64) <= 1e-4 and abs(ours_l - r32) <= 1e-4, (r, ours_l, r32, r64)
    # the whole gradient vector against the fp64 oracle, relative L2.  This small hourglass normalises 2x2 maps of two samples
    # (8 values per channel) at its bottom: fp32 rounding is amplified by orders of magnitude and which implementation draws the
    # larger error depends on the batch -- on the golden 'tiny' batch the reference's fp32 gradient is 2.5e-2 (max-norm) from fp64
    # and ours 7e-3, on this batch ours is 1.2e-2 (relative L2) and the reference's 2.5e-3.  So this is a
    # gross-error bound (a wrong 1/world, a missed bucket or shard is an error of order 1); the arithmetic of the world > 1 path
    # is pinned to 1e-5 against the single-rank runs by _check_against_single_rank_runs above.
    o64, r32 = ours.double(), mean32.double()
    rel_ours = float((o64 - mean64).norm() / mean64.norm())
    rel_ref = float((r32 - mean64).norm() / mean64.norm())
    print('hourglass two shards: gradient rel-L2 vs fp64 %.2e (reference fp32: %.2e), max |g| %.1f' % (rel_ours, rel_ref, float(mean64.abs().max())))
    assert rel_ours <= max(5e-2, 4.0 * rel_ref), (rel_ours, rel_ref)
    # Adam on both replicas: identical parameters, one update of at most lr per element away from the oracle's
    p_before = reps[0].model.device_state().A.tensor('param').clone()
    for rep in reps:
        rep.step.flush()
    torch.cuda.synchronize()
    pa, pb = (rep.model.device_state().A.tensor('param') for rep in reps)
    assert torch.equal(pa, pb)
    d = (pa - p_before).abs()
    assert float(d.max()) <= lr * 1.001 + 1e-7 and float(d.max()) > 0.5 * lr      # first Adam step: |update| <= lr
    s_sd, _ = _cases.state_dicts(name)
    fpd_ref.adam_update(s_sd, {k: mean32[o:o + s_sd[k].numel()].view_as(s_sd[k]) for k, o in
                               zip(keys, np.cumsum([0] + [s_sd[k].numel() for k in keys[:-1]]))}, {}, lr)
    got = reps[0].model.state_dict()
    dev = torch.cat([(got[k].cpu() - s_sd[k]).abs().reshape(-1) for k in keys])
    # lr * sign(g) with noise-level gradients flipping sign: a few elements differ by 2 lr, the bulk agrees
    assert float(dev.max()) <= 2 * lr + 1e-6 and float(dev.mean()) < 0.05 * lr, (float(dev.max()), float(dev.mean()))


def test_hrnet_two_shards_bucketed_reduction_equals_single_reduction():
    from oracle import hrnet_ref
    from tests._cases_hrnet import CONFIG, extra_cfg
    from tests.test_hrnet_gpu import build_pair
    c = CONFIG
    W, H = c['image']

    def make():
        s, t, _, _ = build_pair()
        return s, t
    GB = 4                                                  # two samples per shard (see the hourglass test)
    reps = _two_replicas(make, GB, H, W, c['alpha'], 1e-3)
    assert len([b for b in reps[0].step.student.state.table.buckets if b[1] > b[0]]) >= 3      # stage buckets
    x, tg, tw = fpd_ref.synth_batch(100, GB, c['joints'], c['image'], c['heat'])
    _run_shards(reps, x, tg, tw)
    total = _check_buckets_and_reduce(reps)
    _check_against_single_rank_runs(make, reps, total, x, tg, tw, GB, H, W, c['alpha'], 1e-3)
    ours = _flat_model_grads(reps[0].model).double()
    ex_s, ex_t = extra_cfg(c['s']), extra_cfg(c['t'])
    s_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(ex_s, c['joints']), 1)
    t_sd = fpd_ref.synth_state_dict(hrnet_ref.hrnet_keys(ex_t, c['joints']), 2)
    names = [k for k in s_sd if s_sd[k].is_floating_point() and 'running' not in k]
    fwd = dict(student_forward=lambda sd, xx, train: [hrnet_ref.hrnet_forward(sd, ex_s, xx, train=train)],
               teacher_forward=lambda sd, xx, train: [hrnet_ref.hrnet_forward(sd, ex_t, xx, train=train)])
    mean64, mean32 = 0, 0
    for r in range(2):
        s64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in copy.deepcopy(s_sd).items()}
        t64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in t_sd.items()}
        g = fpd_ref.fpd_step(s64, t64, 1, 1, x[2 * r:2 * r + 2].double(), tg[2 * r:2 * r + 2].double(), tw[2 * r:2 * r + 2].double(), c['alpha'], **fwd)
        mean64 = mean64 + torch.cat([g['grads'][k].reshape(-1) for k in names]) / 2
        g = fpd_ref.fpd_step(copy.deepcopy(s_sd), copy.deepcopy(t_sd), 1, 1, x[2 * r:2 * r + 2], tg[2 * r:2 * r + 2], tw[2 * r:2 * r + 2], c['alpha'], **fwd)
        mean32 = mean32 + torch.cat([g['grads'][k].reshape(-1) for k in names]).double() / 2
    rel = float((ours - mean64).norm() / mean64.norm())
    rel32 = float((mean32 - mean64).norm() / mean64.norm())
    # Against the oracle the figure is dominated by ReLU-kink flips (a pre-activation within rounding distance of zero takes the
    # other branch: ~1e-3 of the gradient each, tests/test_hrnet_graph_cpu.py; with ONE sample per shard it was 1.0e-2 where
    # torch's own fp32 happened to flip none).  A gross-error bound (a wrong 1/world or a missed bucket is an error of order
    # 1); the arithmetic of the world > 1 path itself is pinned to 1e-5 by _check_against_single_rank_runs above.
    print('hrnet two shards: gradient rel-L2 vs fp64 %.2e (reference fp32: %.2e)' % (rel, rel32))
    assert rel <= 1.5e-2, (rel, rel32)
    for rep in reps:
        rep.step.flush()
    torch.cuda.synchronize()
    pa, pb = (rep.model.device_state().A.tensor('param') for rep in reps)
    assert torch.equal(pa, pb)
